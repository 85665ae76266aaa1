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

// Row-major float32 sources (plain (B,H,W) or the interleaved (B,H,W,2) .npy payload): no transpose needed, so
// each thread takes 4 consecutive pixels of one row: 16-byte loads, 16-byte stores when the roll keeps alignment.
template <int SJ>
__global__ __launch_bounds__(256) void range_prep_rows_kernel(const float* __restrict__ raw, PrepArgs a) {
    const int W4 = a.W >> 2;
    const int64_t n4 = (int64_t)a.H * W4;
    const int b = blockIdx.y;
    const int Hl = a.H / a.f, Wl = a.W / a.fw;
    const float* src = raw + (int64_t)b * a.sb;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (int64_t)gridDim.x * 256) {
        const int i = (int)(k / W4), j = (int)(k - (int64_t)i * W4) * 4;
        const bool lo_row = a.lo && i >= a.rp && (i - a.rp) % a.f == 0;
        if (!a.hi && !lo_row) continue;
        float v[4];
        const float* p = src + (int64_t)i * a.si + (int64_t)j * SJ;
        if (SJ == 1) {
            const float4 q = *(const float4*)p;
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const float4 q0 = *(const float4*)p, q1 = *(const float4*)(p + 4);
            v[0] = q0.x; v[1] = q0.z; v[2] = q1.x; v[3] = q1.z;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x = v[c] * a.scale;
            if (a.gate) x = (x >= a.gmin && x <= a.gmax) ? x : 0.f;
            if (a.logt) x = log1pf(x);
            v[c] = x;
        }
        if (a.hi) {
            float* dst = a.hi + ((int64_t)b * a.H + i) * a.W;
            int jo = j + a.shift_hi;
            if (jo >= a.W) jo -= a.W;
            if ((a.shift_hi & 3) == 0) {
                *(float4*)(dst + jo) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { dst[jo] = v[c]; if (++jo == a.W) jo = 0; }
            }
        }
        if (lo_row) {
            float* dst = a.lo + ((int64_t)b * Hl + (i - a.rp) / a.f) * Wl;
            if (a.fw == 1 && (a.shift_lo & 3) == 0) {
                int jl = j + a.shift_lo;
                if (jl >= Wl) jl -= Wl;
                *(float4*)(dst + jl) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int jj = j + c;
                    if (jj >= a.cp && (jj - a.cp) % a.fw == 0) {
                        int jl = (jj - a.cp) / a.fw + a.shift_lo;
                        if (jl >= Wl) jl -= Wl;
                        dst[jl] = v[c];
                    }
                }
            }
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
    const bool rows16 = raw_dtype == 0 && (W & 3) == 0 && base_offset == 0 && (col_stride == 1 || col_stride == 2) &&
                        row_stride == (int64_t)W * col_stride && (batch_stride & 3) == 0 &&
                        ((uintptr_t)raw & 15) == 0 && (!hi || ((uintptr_t)hi & 15) == 0) && (!lo || ((uintptr_t)lo & 15) == 0) &&
                        (Wl & 3) == 0;
    if (rows16) {
        int64_t nb = ((int64_t)H * (W / 4) + 255) / 256;
        if (nb > 2048) nb = 2048;
        if (col_stride == 1)
            hipLaunchKernelGGL(range_prep_rows_kernel<1>, dim3((int)nb, B), dim3(256), 0, stream, (const float*)raw, a);
        else
            hipLaunchKernelGGL(range_prep_rows_kernel<2>, dim3((int)nb, B), dim3(256), 0, stream, (const float*)raw, a);
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    }
    const dim3 grid((W + TJ - 1) / TJ, (H + TI - 1) / TI, B);
    if (raw_dtype == 0)
        hipLaunchKernelGGL(range_prep_kernel<float>, grid, dim3(256), 0, stream, (const float*)raw, a);
    else
        hipLaunchKernelGGL(range_prep_kernel<__half>, grid, dim3(256), 0, stream, (const __half*)raw, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
