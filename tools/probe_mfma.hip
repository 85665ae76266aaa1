// dev probe: issue rate of v_mfma_f32_16x16x32_bf16 -- NA independent accumulators per wave in a loop, W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/probe_mfma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NA, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f80 + i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NA, int NT>
void run(int blocks, const char* tag) {
    const int threads = NT;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NA, NT>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, NT>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NA;
    const double waves = threads / 64.0 * blocks;
    printf("%-28s ticks/MFMA/wave %.2f   kernel %.1f us   TFLOP/s %.0f\n", tag, h / mf, ms * 1e3,
           mf * waves * 16384.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<36, 256>(256, "36 acc, 1 wave/SIMD");
    run<36, 512>(256, "36 acc, 2 waves/SIMD");
    run<16, 256>(256, "16 acc, 1 wave/SIMD");
    run<8, 256>(256, "8 acc, 1 wave/SIMD");
    run<4, 256>(256, "4 acc, 1 wave/SIMD");
    run<16, 1024>(256, "16 acc, 4 waves/SIMD");
    return 0;
}
