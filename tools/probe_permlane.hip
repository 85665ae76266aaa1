// dev probe: the cross-row reductions through v_permlane16_swap / v_permlane32_swap (common.h) against the __shfl_xor form, bit for bit,
// full waves and waves whose upper half / odd lanes have left the loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../tulip_amd/csrc/common.h"
template <int W> __device__ float old_sum(float v) {
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    if (W >= 32) v += __shfl_xor(v, 16, 64);
    if (W >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__global__ void k(const float* in, float* out, int mode) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float v = in[i];
    float r[6] = {0, 0, 0, 0, 0, 0};
    const bool active = mode == 0 || (mode == 1 && threadIdx.x < 32) || (mode == 2 && threadIdx.x < 48);
    if (active) {
        r[0] = old_sum<32>(v); r[1] = group_sum<32>(v);
        r[2] = old_sum<64>(v); r[3] = group_sum<64>(v);
        float s = v; s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        r[4] = s; r[5] = rows_sum(v);
    }
    for (int j = 0; j < 6; ++j) out[j * gridDim.x * 64 + i] = r[j];
}
int main() {
    const int N = 64 * 256;
    float *h = new float[N], *r = new float[6 * N];
    for (int i = 0; i < N; ++i) h[i] = (float)((i * 2654435761u) % 100003) / 977.0f - 50.0f;
    float *din, *dout;
    hipMalloc(&din, N * 4); hipMalloc(&dout, 6 * N * 4);
    hipMemcpy(din, h, N * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, din, dout, mode);
        hipMemcpy(r, dout, 6 * N * 4, hipMemcpyDeviceToHost);
        int bad[3] = {0, 0, 0};
        for (int i = 0; i < N; ++i) {
            const int lane = i % 64;
            const bool act = mode == 0 || (mode == 1 && lane < 32) || (mode == 2 && lane < 48);
            if (!act) continue;
            for (int p = 0; p < 3; ++p) {
                if (p == 0 && mode == 2 && lane >= 32) continue;   // 32-lane group with half of it gone: undefined either way
                if (p >= 1 && mode != 0) continue;                 // 64-lane reductions need the whole wave
                bad[p] += memcmp(&r[(2 * p) * N + i], &r[(2 * p + 1) * N + i], 4) != 0;
            }
        }
        printf("mode %d: group_sum<32> mismatches %d, group_sum<64> %d, rows_sum %d\n", mode, bad[0], bad[1], bad[2]);
        rc |= bad[0] | bad[1] | bad[2];
    }
    return rc != 0;
}
