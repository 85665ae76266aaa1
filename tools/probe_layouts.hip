// Probe: MFMA fragment layouts + ds_read_b64_tr_b16 semantics on gfx950 (dev tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }

// C[16][16] = A[16][32] * B[32][16]; A row-major (k contiguous), Bt[n][k] (k contiguous)
__global__ void k_mfma32(const unsigned short* A, const unsigned short* Bt, float* C) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + (l >> 4) * 8 + j]; b[j] = Bt[(l & 15) * 32 + (l >> 4) * 8 + j]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// C[16][16] = A[16][16]*B[16][16]; 16x16x16: lane holds A[l&15][(l>>4)*4+j], B[(l>>4)*4+j][l&15]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* Bt, float* C) {
    int l = threadIdx.x;
    bf16x4 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = A[(l & 15) * 16 + (l >> 4) * 4 + j]; b[j] = Bt[(l & 15) * 16 + (l >> 4) * 4 + j]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// tr read: LDS holds M[row][col] (u16 = row*256+col), pitch P elements. lane i of 16-lane group g gives
// address of row (g*4 + i/4), col (i%4)*4.  Expect out[l][j] = M[g*4+j][l&15].
__global__ void k_tr(unsigned short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) lds[i] = (unsigned short)(((i / pitch) << 8) | (i % pitch));
    __syncthreads();
    int g = l >> 4, i = l & 15;
    unsigned addr = (unsigned)(uintptr_t)(&lds[(g * 4 + i / 4) * pitch + (i % 4) * 4]);
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    std::vector<unsigned short> A(16 * 32), Bt(16 * 32);
    std::vector<float> Af(16 * 32), Bf(16 * 32);
    srand(1);
    for (int i = 0; i < 16 * 32; ++i) { Af[i] = (rand() % 17 - 8) / 4.f; Bf[i] = (rand() % 13 - 6) / 2.f; A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    unsigned short *dA, *dB, *dO; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dO, 512);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    float C[256];
    k_mfma32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[j * 32 + k]; e = fmax(e, fabs(s - C[i * 16 + j])); }
    printf("mfma_16x16x32_bf16 layout max err %g  (%s)\n", e, e < 1e-3 ? "OK" : "MISMATCH");
    // 16x16x16 uses first 16 k of each row: build compact arrays
    std::vector<unsigned short> A2(256), B2(256); for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) { A2[i * 16 + k] = A[i * 32 + k]; B2[i * 16 + k] = Bt[i * 32 + k]; }
    hipMemcpy(dA, A2.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 512, hipMemcpyHostToDevice);
    k_mfma16<<<1, 64>>>(dA, dB, dC); hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost);
    e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 32 + k] * Bf[j * 32 + k]; e = fmax(e, fabs(s - C[i * 16 + j])); }
    printf("mfma_16x16x16bf16_1k layout max err %g  (%s)\n", e, e < 1e-3 ? "OK" : "MISMATCH");
    for (int pitch : {16, 64}) {
        unsigned short O[256];
        k_tr<<<1, 64>>>(dO, pitch); hipMemcpy(O, dO, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int g = l >> 4; unsigned short exp = (unsigned short)(((g * 4 + j) << 8) | (l & 15)); if (O[l * 4 + j] != exp) ++bad; }
        printf("ds_read_b64_tr_b16 pitch=%d: %d mismatches vs out[l][j]=M[g*4+j][l&15]\n", pitch, bad);
        if (bad) { for (int l = 0; l < 64; l += 5) printf("  lane %2d: %04x %04x %04x %04x\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]); }
    }
    hipError_t err = hipDeviceSynchronize();
    printf("final: %s\n", hipGetErrorString(err));
    return 0;
}
