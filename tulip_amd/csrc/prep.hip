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

// ---- KITTI point cloud -> range image (kitti_utils/sample_kitti_dataset.py:24-66), float32 in numpy's op order
struct KittiProj {
    int rows, cols;
    float ang_start_y, ang_res_y, ang_res_x, max_range, min_range, half_cols;
};
__device__ __forceinline__ bool kitti_pixel(const float* __restrict__ p, const KittiProj& k, int& row, int& col) {
#pragma clang fp contract(off)
    const float x = p[0], y = p[1], z = p[2];
    const float deg = 180.0f, pi = 3.14159274101257324f;                       // float32(180.0), float32(np.pi)
    const float vert = atan2f(z, sqrtf(x * x + y * y)) * deg / pi;              // :32
    const float r = rintf((vert + k.ang_start_y) / k.ang_res_y);                // :33-34 (round half to even)
    const float hor = atan2f(x, y) * deg / pi;                                  // :36
    const float c = truncf((hor - 90.0f) / k.ang_res_x);                        // :38 np.int_ truncates
    if (!(fabsf(r) < 1e9f) || !(fabsf(c) < 1e9f)) return false;
    long long ci = -(long long)c + (long long)k.half_cols;
    if (ci >= k.cols) ci -= k.cols;                                             // :40-41
    const long long ri = (long long)r;
    row = (int)ri; col = (int)ci;
    return ri >= 0 && ri < k.rows && ci >= 0 && ci < k.cols;                    // :49
}
__global__ __launch_bounds__(256) void kitti_mark_kernel(const float* __restrict__ pts, int64_t n, KittiProj k,
                                                         int* __restrict__ winner) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int row, col;
        if (kitti_pixel(pts + i * 4, k, row, col)) atomicMax(&winner[row * k.cols + col], (int)i);   // last point wins
    }
}
__global__ __launch_bounds__(256) void kitti_resolve_kernel(const float* __restrict__ pts, KittiProj k,
                                                            int* __restrict__ winner, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int npix = k.rows * k.cols;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const int w = winner[i];
        float rng = 0.f, inten = 0.f;
        if (w >= 0) {
            const float* p = pts + (int64_t)w * 4;
            rng = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);               // :43
            if (rng > k.max_range || rng < k.min_range) rng = 0.f;               // :44-45
            inten = p[3];                                                        // :47-48 never fire (range already 0)
        }
        out[(int64_t)i * 2] = rng;
        out[(int64_t)i * 2 + 1] = inten;
        winner[i] = -1;                                                          // scratch handed back reset
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

extern "C" int tulip_kitti_range_map(const float* points, int64_t n, int rows, int cols, float ang_start_y,
                                     float ang_res_y, float ang_res_x, float max_range, float min_range,
                                     int32_t* winner, float* out, hipStream_t stream) {
    if (!points || !winner || !out || n < 0 || n > 0x7fffffff || rows <= 0 || cols <= 0 || !(ang_res_y > 0.f) ||
        !(ang_res_x > 0.f))
        return TULIP_ERR_ARG;
    KittiProj k{rows, cols, ang_start_y, ang_res_y, ang_res_x, max_range, min_range, (float)cols / 2.0f};
    if (n > 0) {
        int64_t nb = (n + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(kitti_mark_kernel, dim3((int)nb), dim3(256), 0, stream, points, n, k, winner);
    }
    hipLaunchKernelGGL(kitti_resolve_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, stream, points, k, winner,
                       out);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
