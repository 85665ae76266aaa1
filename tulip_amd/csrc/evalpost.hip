// Evaluation post-processing and 3-D metrics on the GPU:
//   engine_upsampling.py:411-426 (MC-dropout aggregate), :176-251 (expm1, gate, MAE, low-res rows restored),
//   util/evaluation.py:21-116 (range image -> point cloud), :125-135 (Chamfer), :148-175 (voxel IoU/precision/recall).
// Byte/elementwise work, a brute-force nearest-neighbour sweep on the vector ALUs (direct (a-b)^2 form: the
// |a|^2+|b|^2-2ab GEMM form loses the ~1e-4 m^2 distances to cancellation at |a| ~ 80 m), and a bitmap voxel set
// whose cost is O(points), not O(grid).  Everything is stream-ordered; nothing here synchronises with the host.
#include <hip/hip_runtime.h>

#include "common.h"
#include "tulip_hip.h"

namespace {

constexpr int RB = 1024;   // max blocks of the two-stage reductions

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
inline int blocks_for(int64_t work, int cap = RB) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ MC-dropout aggregate (engine:421-426)
__global__ __launch_bounds__(256) void mc_aggregate_kernel(const float* __restrict__ preds, int T, int64_t n,
                                                           float thr, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int t = 0; t < T; ++t) s += (double)preds[(int64_t)t * n + i];
        const double mean = s / T;
        double q = 0.0;
        for (int t = 0; t < T; ++t) { const double d = (double)preds[(int64_t)t * n + i] - mean; q += d * d; }
        const float mf = (float)mean;
        const float sd = (float)sqrt(q / (double)(T - 1));          // torch.std: unbiased
        out[i] = (sd > thr * mf) ? 0.f : mf;                          // pred_img[std > threshold*pred_img] = 0
    }
}

// ------------------------------------------------------------------ post-processing (engine:176-251)
struct PostArgs {
    const float* pred; const float* hi; const float* lo;
    float* pred_img; float* hi_img;
    int H, W, h, f, restore_rows, logt;
    float gmin, gmax, keep_close;
};
__global__ __launch_bounds__(256) void post_kernel(PostArgs a, double* __restrict__ partials) {
    __shared__ double red[4];
    const int64_t n = (int64_t)a.H * a.W;
    double e_all = 0.0, e_low = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int i = (int)(k / a.W), j = (int)(k - (int64_t)i * a.W);
        float p = a.pred[k], t = a.hi[k];
        if (a.logt) { p = expm1f(p); t = expm1f(t); }                               // :178-181
        p = (p >= a.gmin && p <= a.gmax) ? p : 0.f;                                 // :183-190
        e_all += (double)fabsf(p - t);                                              // :192-193 (before the rows are restored)
        if (a.restore_rows && i % a.f == 0) {
            float l = a.lo[(int64_t)(i / a.f) * a.W + j];
            if (a.logt) l = expm1f(l);
            e_low += (double)fabsf(p - l);                                          // :226-228
            p = l;                                                                  // :230
        }
        if (a.keep_close > 0.f) {                                                   // :247-249, :487-489
            if (p > a.keep_close) p = 0.f;
            if (t > a.keep_close) t = 0.f;
        }
        a.pred_img[k] = p;
        a.hi_img[k] = t;
    }
    e_all = block_sum_d(e_all, red);
    e_low = block_sum_d(e_low, red);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = e_all; partials[blockIdx.x * 2 + 1] = e_low; }
}
__global__ __launch_bounds__(256) void post_final_kernel(const double* __restrict__ partials, int nb, double inv_all,
                                                         double inv_low, float* __restrict__ out) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) { s0 += partials[i * 2]; s1 += partials[i * 2 + 1]; }
    s0 = block_sum_d(s0, red);
    s1 = block_sum_d(s1, red);
    if (threadIdx.x == 0) { out[0] = (float)(s0 * inv_all); out[1] = (float)(s1 * inv_low); }
}

// ------------------------------------------------------------------ range image -> xyz
// KITTI / CARLA (evaluation.py:52-116): products only, in the reference's order, so float32 results are bit exact
__global__ __launch_bounds__(256) void xyz_spherical_kernel(const float* __restrict__ img, const float* __restrict__ sh,
                                                            const float* __restrict__ ch, const float* __restrict__ sv,
                                                            const float* __restrict__ cv, float max_range, int H, int W,
                                                            float* __restrict__ xyz) {
    const int64_t n = (int64_t)H * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int i = (int)(k / W), j = (int)(k - (int64_t)i * W);
        const float r = img[k] * max_range;
        xyz[k * 3 + 0] = (sh[j] * cv[i]) * r;
        xyz[k * 3 + 1] = (ch[j] * cv[i]) * r;
        xyz[k * 3 + 2] = sv[i] * r;
    }
}
// DurLAR / Ouster OS1-128 (evaluation.py:21-50): float32 (r - origin_offset), then float64 multiply-then-add
// exactly as numpy evaluates it (no fused multiply-add), scattered to the staggered-beam pixel index.
__global__ __launch_bounds__(256) void xyz_durlar_kernel(const float* __restrict__ img, const double* __restrict__ colt,
                                                         const double* __restrict__ rowt,
                                                         const int32_t* __restrict__ row_offset, float max_range,
                                                         float origin_offset, double z_offset, int H, int W,
                                                         double* __restrict__ xyz) {
#pragma clang fp contract(off)
    const int64_t n = (int64_t)H * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int v = (int)(k / W), u = (int)(k - (int64_t)v * W);
        const float tf = img[k] * max_range - origin_offset;
        const double t = (double)tf;
        const double x = (t * colt[u]) * rowt[v] + colt[2 * W + u];            // cos(enc+az)*cos(el) + o*cos(enc)
        const double y = (t * colt[W + u]) * rowt[v] + colt[3 * W + u];        // sin(enc+az)*cos(el) + o*sin(enc)
        const double z = t * rowt[H + v];
        const int64_t dst = (int64_t)v * W + (u + W - row_offset[v]) % W;
        xyz[dst * 3 + 0] = -x;
        xyz[dst * 3 + 1] = -y;
        xyz[dst * 3 + 2] = z + z_offset;
    }
}

// ------------------------------------------------------------------ voxel metrics (engine:254-271, evaluation.py:148-175)
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const T* __restrict__ a, int64_t na, const T* __restrict__ b,
                                                             int64_t nb, double* __restrict__ partials) {
    __shared__ double red[6][4];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * 256) {
        const T* p = i < na ? a + i * 3 : b + (i - na) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double v = (double)p[c]; mn[c] = fmin(mn[c], v); mx[c] = fmax(mx[c], v); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fmin(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmax(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[c][threadIdx.x >> 6] = mn[c]; red[3 + c][threadIdx.x >> 6] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double* r = red[threadIdx.x];
        partials[blockIdx.x * 6 + threadIdx.x] = threadIdx.x < 3 ? fmin(fmin(r[0], r[1]), fmin(r[2], r[3]))
                                                                 : fmax(fmax(r[0], r[1]), fmax(r[2], r[3]));
    }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const double* __restrict__ partials, int nb,
                                                           double* __restrict__ out) {
    __shared__ double red[6][4];
    double v[6] = {1e300, 1e300, 1e300, -1e300, -1e300, -1e300};
    for (int i = threadIdx.x; i < nb; i += 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = fmin(v[c], partials[i * 6 + c]);
            v[3 + c] = fmax(v[3 + c], partials[i * 6 + 3 + c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(v[c], o, 64);
            v[c] = c < 3 ? fmin(v[c], w) : fmax(v[c], w);
        }
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v[c];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double* r = red[threadIdx.x];
        out[threadIdx.x] = threadIdx.x < 3 ? fmin(fmin(r[0], r[1]), fmin(r[2], r[3])) : fmax(fmax(r[0], r[1]), fmax(r[2], r[3]));
    }
}

// voxel index of a point in the clouds' own arithmetic: ((p - min)/grid).astype(int); dims = ((max-min)/grid).astype(int)+1
template <typename T>
__device__ __forceinline__ int64_t voxel_key(const T* __restrict__ p, const double* __restrict__ mm, T grid,
                                             int64_t& cells) {
    int64_t idx[3], dims[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T mn = (T)mm[c], mx = (T)mm[3 + c];
        dims[c] = (int64_t)((mx - mn) / grid) + 1;
        idx[c] = (int64_t)((p[c] - mn) / grid);
    }
    cells = dims[0] * dims[1] * dims[2];
    return (idx[0] * dims[1] + idx[1]) * dims[2] + idx[2];
}

template <typename T, bool IS_GT>
__global__ __launch_bounds__(256) void voxel_mark_kernel(const T* __restrict__ pts, int64_t n,
                                                         const double* __restrict__ mm, T grid,
                                                         uint32_t* __restrict__ mine, const uint32_t* __restrict__ other,
                                                         int64_t capacity_bits, unsigned long long* __restrict__ counters) {
    __shared__ double red[4];
    unsigned cnt = 0, inter = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int64_t cells;
        const int64_t key = voxel_key<T>(pts + i * 3, mm, grid, cells);
        if (cells > capacity_bits || key < 0 || key >= cells) { bad = 1; continue; }
        const uint32_t bit = 1u << (key & 31);
        const uint32_t old = atomicOr(&mine[key >> 5], bit);
        if (!(old & bit)) {                      // first point in this voxel
            ++cnt;
            if (IS_GT && (other[key >> 5] & bit)) ++inter;
        }
    }
    const double c = block_sum_d((double)cnt, red), i2 = block_sum_d((double)inter, red), b = block_sum_d((double)bad, red);
    if (threadIdx.x == 0) {
        if (c > 0) atomicAdd(&counters[IS_GT ? 1 : 0], (unsigned long long)c);
        if (IS_GT && i2 > 0) atomicAdd(&counters[2], (unsigned long long)i2);
        if (b > 0) atomicAdd(&counters[3], 1ull);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void voxel_clear_kernel(const T* __restrict__ a, int64_t na, const T* __restrict__ b,
                                                          int64_t nb, const double* __restrict__ mm, T grid,
                                                          uint32_t* __restrict__ bm_a, uint32_t* __restrict__ bm_b,
                                                          int64_t capacity_bits) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * 256) {
        int64_t cells;
        const bool first = i < na;
        const int64_t key = voxel_key<T>(first ? a + i * 3 : b + (i - na) * 3, mm, grid, cells);
        if (cells > capacity_bits || key < 0 || key >= cells) continue;
        (first ? bm_a : bm_b)[key >> 5] = 0u;
    }
}
template <typename T>
__global__ void voxel_final_kernel(unsigned long long* __restrict__ counters, const double* __restrict__ mm, T grid,
                                   double* __restrict__ out) {
    const double P = (double)counters[0], G = (double)counters[1], I = (double)counters[2];
    const bool bad = counters[3] != 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double iou = I / (P + G - I);                 // sum(intersection)/sum(union)
    const double precision = I / (I + (P - I));         // tp/(tp+fp)
    const double recall = I / (I + (G - I));            // tp/(tp+fn)
    out[0] = bad ? nan : iou;
    out[1] = bad ? nan : precision;
    out[2] = bad ? nan : recall;
    out[3] = bad ? nan : 2.0 * (precision * recall) / (precision + recall);
    for (int c = 0; c < 3; ++c) out[4 + c] = (double)((int64_t)(((T)mm[3 + c] - (T)mm[c]) / grid) + 1);
    out[7] = bad ? 1.0 : 0.0;
    counters[0] = counters[1] = counters[2] = counters[3] = 0ull;
}

// ------------------------------------------------------------------ Chamfer (evaluation.py:125-135)
constexpr int CQ = 4;        // query points per thread
constexpr int CT = 1024;     // target points per LDS tile
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, int64_t n, uint32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void nn_sq_kernel(const TA* __restrict__ a, int64_t na, const TB* __restrict__ b,
                                                    int64_t nb, int64_t chunk, uint32_t* __restrict__ dmin) {
    __shared__ float4 tile[CT];
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * 256 * CQ;
    float qx[CQ], qy[CQ], qz[CQ], m[CQ];
#pragma unroll
    for (int k = 0; k < CQ; ++k) {
        const int64_t q = q0 + tid + k * 256;
        const bool ok = q < na;
        qx[k] = ok ? (float)a[q * 3] : 0.f;
        qy[k] = ok ? (float)a[q * 3 + 1] : 0.f;
        qz[k] = ok ? (float)a[q * 3 + 2] : 0.f;
        m[k] = __uint_as_float(0x7f800000u);
    }
    const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = t0 + chunk < nb ? t0 + chunk : nb;
    for (int64_t base = t0; base < t1; base += CT) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CT / 256; ++k) {
            const int64_t j = base + tid + k * 256;
            // points past the end sit far away: they never win the min and need no branch below
            tile[tid + k * 256] = j < t1 ? make_float4((float)b[j * 3], (float)b[j * 3 + 1], (float)b[j * 3 + 2], 0.f)
                                         : make_float4(1e18f, 1e18f, 1e18f, 0.f);
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < CT; ++j) {
            const float4 p = tile[j];                       // same address in every lane: LDS broadcast
#pragma unroll
            for (int k = 0; k < CQ; ++k) {
                const float dx = qx[k] - p.x, dy = qy[k] - p.y, dz = qz[k] - p.z;
                m[k] = fminf(m[k], dx * dx + dy * dy + dz * dz);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CQ; ++k) {
        const int64_t q = q0 + tid + k * 256;
        if (q < na) atomicMin(&dmin[q], __float_as_uint(m[k]));   // non-negative floats order like their bit patterns
    }
}
__global__ __launch_bounds__(256) void chamfer_partial_kernel(const float* __restrict__ da, int64_t na,
                                                              const float* __restrict__ db, int64_t nb,
                                                              double* __restrict__ partials) {
    __shared__ double red[4];
    double sa = 0.0, sb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na; i += (int64_t)gridDim.x * 256) sa += (double)da[i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) sb += (double)db[i];
    sa = block_sum_d(sa, red);
    sb = block_sum_d(sb, red);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = sa; partials[blockIdx.x * 2 + 1] = sb; }
}
__global__ __launch_bounds__(256) void chamfer_final_kernel(const double* __restrict__ partials, int nblk, double inv_a,
                                                            double inv_b, double* __restrict__ out) {
    __shared__ double red[4];
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { sa += partials[i * 2]; sb += partials[i * 2 + 1]; }
    sa = block_sum_d(sa, red);
    sb = block_sum_d(sb, red);
    if (threadIdx.x == 0) out[0] = sa * inv_a + sb * inv_b;     // mean(dist1) + mean(dist2)
}

template <typename TA, typename TB>
void launch_nn(const void* a, int64_t na, const void* b, int64_t nb, uint32_t* dmin, hipStream_t stream) {
    const int qblocks = (int)((na + 256 * CQ - 1) / (256 * CQ));
    // enough (query block, target slice) pairs to fill 256 CUs a few times over; slices are whole LDS tiles
    int splits = (2048 + qblocks - 1) / qblocks;
    const int64_t tiles = (nb + CT - 1) / CT;
    if (splits > tiles) splits = (int)tiles;
    if (splits < 1) splits = 1;
    const int64_t chunk = ((tiles + splits - 1) / splits) * CT;
    splits = (int)((nb + chunk - 1) / chunk);
    hipLaunchKernelGGL((nn_sq_kernel<TA, TB>), dim3(qblocks, splits), dim3(256), 0, stream, (const TA*)a, na, (const TB*)b,
                       nb, chunk, dmin);
}

}  // namespace

extern "C" int tulip_mc_aggregate(const float* preds, int passes, int64_t n, float noise_threshold, float* out,
                                  hipStream_t stream) {
    if (!preds || !out || passes < 2 || n <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(mc_aggregate_kernel, dim3(blocks_for(n, 4096)), dim3(256), 0, stream, preds, passes, n,
                       noise_threshold, out);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_eval_postprocess(const float* pred, const float* hi, const float* lo, float* pred_img,
                                      float* hi_img, double* partials, float* mae_out, int H, int W, int h, int w,
                                      int log_transform, float gate_min, float gate_max, float keep_close,
                                      hipStream_t stream) {
    if (!pred || !hi || !lo || !pred_img || !hi_img || !partials || !mae_out || H <= 0 || W <= 0 || h <= 0 || H % h)
        return TULIP_ERR_ARG;
    PostArgs a;
    a.pred = pred; a.hi = hi; a.lo = lo; a.pred_img = pred_img; a.hi_img = hi_img;
    a.H = H; a.W = W; a.h = h; a.f = H / h; a.restore_rows = (w == W); a.logt = log_transform;
    a.gmin = gate_min; a.gmax = gate_max; a.keep_close = keep_close;
    const int nb = blocks_for((int64_t)H * W);
    hipLaunchKernelGGL(post_kernel, dim3(nb), dim3(256), 0, stream, a, partials);
    hipLaunchKernelGGL(post_final_kernel, dim3(1), dim3(256), 0, stream, partials, nb, 1.0 / ((double)H * W),
                       a.restore_rows ? 1.0 / ((double)h * W) : 0.0, mae_out);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_range_to_xyz(const float* img, const float* sin_h, const float* cos_h, const float* sin_v,
                                  const float* cos_v, float max_range, int H, int W, float* xyz, hipStream_t stream) {
    if (!img || !sin_h || !cos_h || !sin_v || !cos_v || !xyz || H <= 0 || W <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(xyz_spherical_kernel, dim3(blocks_for((int64_t)H * W)), dim3(256), 0, stream, img, sin_h, cos_h,
                       sin_v, cos_v, max_range, H, W, xyz);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_range_to_xyz_durlar(const float* img, const double* col_tables, const double* row_tables,
                                         const int32_t* row_offset, float max_range, float origin_offset,
                                         double z_offset, int H, int W, double* xyz, hipStream_t stream) {
    if (!img || !col_tables || !row_tables || !row_offset || !xyz || H <= 0 || W <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(xyz_durlar_kernel, dim3(blocks_for((int64_t)H * W)), dim3(256), 0, stream, img, col_tables,
                       row_tables, row_offset, max_range, origin_offset, z_offset, H, W, xyz);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

namespace {
template <typename T>
int voxel_metrics_t(const void* pp, int64_t np_, const void* pg, int64_t ng, double grid_size, uint32_t* bm_p,
                    uint32_t* bm_g, int64_t words, double* scratch, double* out, hipStream_t stream) {
    const T* a = (const T*)pp;
    const T* b = (const T*)pg;
    double* partials = scratch;                 // RB*6
    double* mm = scratch + RB * 6;              // 6
    unsigned long long* counters = (unsigned long long*)(scratch + RB * 6 + 8);   // 4
    const T grid = (T)grid_size;
    const int64_t cap = words * 32;
    const int nb = blocks_for(np_ + ng);
    hipLaunchKernelGGL(minmax_partial_kernel<T>, dim3(nb), dim3(256), 0, stream, a, np_, b, ng, partials);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, stream, partials, nb, mm);
    hipLaunchKernelGGL((voxel_mark_kernel<T, false>), dim3(blocks_for(np_)), dim3(256), 0, stream, a, np_, mm, grid, bm_p,
                       (const uint32_t*)bm_g, cap, counters);
    hipLaunchKernelGGL((voxel_mark_kernel<T, true>), dim3(blocks_for(ng)), dim3(256), 0, stream, b, ng, mm, grid, bm_g,
                       (const uint32_t*)bm_p, cap, counters);
    hipLaunchKernelGGL(voxel_final_kernel<T>, dim3(1), dim3(1), 0, stream, counters, mm, grid, out);
    hipLaunchKernelGGL(voxel_clear_kernel<T>, dim3(nb), dim3(256), 0, stream, a, np_, b, ng, mm, grid, bm_p, bm_g, cap);
    return TULIP_OK;
}
}  // namespace

extern "C" int tulip_voxel_metrics(const void* pcd_pred, int64_t n_pred, const void* pcd_gt, int64_t n_gt, int is_f64,
                                   double grid_size, uint32_t* bitmap_pred, uint32_t* bitmap_gt, int64_t bitmap_words,
                                   double* scratch, double* out, hipStream_t stream) {
    if (!pcd_pred || !pcd_gt || n_pred <= 0 || n_gt <= 0 || !bitmap_pred || !bitmap_gt || bitmap_words <= 0 ||
        !scratch || !out || !(grid_size > 0.0))
        return TULIP_ERR_ARG;
    if (is_f64)
        voxel_metrics_t<double>(pcd_pred, n_pred, pcd_gt, n_gt, grid_size, bitmap_pred, bitmap_gt, bitmap_words, scratch,
                                out, stream);
    else
        voxel_metrics_t<float>(pcd_pred, n_pred, pcd_gt, n_gt, grid_size, bitmap_pred, bitmap_gt, bitmap_words, scratch,
                               out, stream);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_chamfer_sq(const void* a, int64_t na, const void* b, int64_t nb, int is_f64, float* dist_a,
                                float* dist_b, double* scratch, double* out, hipStream_t stream) {
    if (!a || !b || na <= 0 || nb <= 0 || !dist_a || !dist_b || !scratch || !out) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks_for(na)), dim3(256), 0, stream, (uint32_t*)dist_a, na, 0x7f800000u);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks_for(nb)), dim3(256), 0, stream, (uint32_t*)dist_b, nb, 0x7f800000u);
    if (is_f64) {
        launch_nn<double, double>(a, na, b, nb, (uint32_t*)dist_a, stream);
        launch_nn<double, double>(b, nb, a, na, (uint32_t*)dist_b, stream);
    } else {
        launch_nn<float, float>(a, na, b, nb, (uint32_t*)dist_a, stream);
        launch_nn<float, float>(b, nb, a, na, (uint32_t*)dist_b, stream);
    }
    const int nblk = blocks_for(na > nb ? na : nb);
    hipLaunchKernelGGL(chamfer_partial_kernel, dim3(nblk), dim3(256), 0, stream, dist_a, na, dist_b, nb, scratch);
    hipLaunchKernelGGL(chamfer_final_kernel, dim3(1), dim3(256), 0, stream, scratch, nblk, 1.0 / (double)na,
                       1.0 / (double)nb, out);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
