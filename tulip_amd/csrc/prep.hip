// Range-image input transforms as one pass (util/datasets.py:68-70,96-151,175-193,244-340):
// raw sensor image (metres; float32 or float16, any element strides so the interleaved (H,W,2) .npy payload
// and the transposed+flipped .rimg payload are read in place) -> high-res target (B,1,H,W) and low-res input
// (B,1,H/f,W/fw): x*scale -> range gate -> row/column subsample -> log1p -> roll along W.
// HBM-bound byte work: every raw element is read once, every output written once, tiles go through LDS so
// that both the read (along the source's unit-stride axis) and the write (along W) are coalesced.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "common.h"
#include "tulip_hip.h"

namespace {

constexpr int TI = 32, TJ = 64;

struct PrepArgs {
    int64_t sb, si, sj, base;      // element strides of the raw image and offset of pixel (0,0)
    float* hi;
    float* lo;
    int H, W, f, fw, rp, cp;
    float scale, gmin, gmax;
    int gate, logt, shift_hi, shift_lo, along_i;
};

template <typename T>
__device__ __forceinline__ float raw_to_float(T v);
template <>
__device__ __forceinline__ float raw_to_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ float raw_to_float<__half>(__half v) { return __half2float(v); }

template <typename T>
__global__ __launch_bounds__(256) void range_prep_kernel(const T* __restrict__ raw, PrepArgs a) {
    __shared__ float tile[TI][TJ + 1];
    const int b = blockIdx.z, i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ, t = threadIdx.x;
    const T* src = raw + (int64_t)b * a.sb + a.base;
#pragma unroll
    for (int k = 0; k < TI * TJ / 256; ++k) {
        // fast thread index along the source's unit-stride axis
        const int ii = a.along_i ? (t & (TI - 1)) : (t / TJ + k * (256 / TJ));
        const int jj = a.along_i ? (t / TI + k * (256 / TI)) : (t & (TJ - 1));
        const int i = i0 + ii, j = j0 + jj;
        float v = 0.f;
        const bool need = a.hi || ((i - a.rp) % a.f == 0);
        if (i < a.H && j < a.W && need) {
            v = raw_to_float<T>(src[(int64_t)i * a.si + (int64_t)j * a.sj]) * a.scale;   // ScaleTensor
            if (a.gate) v = (v >= a.gmin && v <= a.gmax) ? v : 0.f;                        // FilterInvalidPixels
            if (a.logt) v = log1pf(v);                                                      // LogTransform
        }
        tile[ii][jj] = v;
    }
    __syncthreads();
    const int Hl = a.H / a.f, Wl = a.W / a.fw;
#pragma unroll
    for (int k = 0; k < TI * TJ / 256; ++k) {
        const int ii = t / TJ + k * (256 / TJ), jj = t & (TJ - 1);
        const int i = i0 + ii, j = j0 + jj;
        if (i >= a.H || j >= a.W) continue;
        const float v = tile[ii][jj];
        if (a.hi) {
            int jo = j + a.shift_hi;                                                        // torch.roll along W
            if (jo >= a.W) jo -= a.W;
            a.hi[((int64_t)b * a.H + i) * a.W + jo] = v;
        }
        if (a.lo && i >= a.rp && j >= a.cp && (i - a.rp) % a.f == 0 && (j - a.cp) % a.fw == 0) {
            const int il = (i - a.rp) / a.f;
            int jl = (j - a.cp) / a.fw + a.shift_lo;
            if (jl >= Wl) jl -= Wl;
            if (il < Hl) a.lo[((int64_t)b * Hl + il) * Wl + jl] = v;
        }
    }
}

inline int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

}  // namespace

extern "C" int tulip_range_prep(const void* raw, int raw_dtype, int64_t batch_stride, int64_t row_stride,
                                int64_t col_stride, int64_t base_offset, float* hi, float* lo, int B, int H, int W,
                                int row_factor, int col_factor, int row_phase, int col_phase, float scale, int gate,
                                float min_range, float max_range, int log_transform, int roll_shift,
                                hipStream_t stream) {
    if (!raw || (!hi && !lo) || B <= 0 || H <= 0 || W <= 0 || row_factor < 1 || col_factor < 1) return TULIP_ERR_ARG;
    if (H % row_factor || W % col_factor || row_phase < 0 || row_phase >= row_factor || col_phase < 0 ||
        col_phase >= col_factor || (raw_dtype != 0 && raw_dtype != 1))
        return TULIP_ERR_ARG;
    PrepArgs a;
    a.sb = batch_stride; a.si = row_stride; a.sj = col_stride; a.base = base_offset;
    a.hi = hi; a.lo = lo;
    a.H = H; a.W = W; a.f = row_factor; a.fw = col_factor; a.rp = row_phase; a.cp = col_phase;
    a.scale = scale; a.gmin = min_range; a.gmax = max_range; a.gate = gate; a.logt = log_transform;
    const int Wl = W / col_factor;
    a.shift_hi = ((roll_shift % W) + W) % W;
    a.shift_lo = ((roll_shift % Wl) + Wl) % Wl;
    a.along_i = iabs64(row_stride) < iabs64(col_stride);
    const dim3 grid((W + TJ - 1) / TJ, (H + TI - 1) / TI, B);
    if (raw_dtype == 0)
        hipLaunchKernelGGL(range_prep_kernel<float>, grid, dim3(256), 0, stream, (const float*)raw, a);
    else
        hipLaunchKernelGGL(range_prep_kernel<__half>, grid, dim3(256), 0, stream, (const __half*)raw, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
