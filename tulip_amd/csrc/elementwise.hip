// HBM-bound helpers: fp32->bf16 casts (with DropPath row scale / concat / inverse pixel shuffle),
// column sums (bias gradients), L1 loss forward/backward, fused AdamW.   gfx950 only.
#include <cstdlib>
#include "common.h"
#include "tulip_hip.h"

namespace {

// y[r][c] = bf16(x[r][c] * rowscale[r / rows_per_sample]); 8 elements / thread
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n8,
                                                        int cols8, const float* __restrict__ rowscale,
                                                        int rows_per_sample) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float s = 1.0f;
        if (rowscale) s = rowscale[(int)(i / cols8) / rows_per_sample];
        const float4 a = *(const float4*)(x + i * 8), b = *(const float4*)(x + i * 8 + 4);
        *(uint4*)(y + i * 8) = make_uint4(pack_bf16x2(a.x * s, a.y * s), pack_bf16x2(a.z * s, a.w * s),
                                          pack_bf16x2(b.x * s, b.y * s), pack_bf16x2(b.z * s, b.w * s));
    }
}

__global__ __launch_bounds__(256) void cast_flat_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = *(const float4*)(x + i * 4);
        *(uint2*)(y + i * 4) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = f2bf(x[n4 * 4 + threadIdx.x]);
}

__global__ __launch_bounds__(256) void cast_up_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint2 a = *(const uint2*)(x + i * 4);
        *(float4*)(y + i * 4) = make_float4(bf2f((bf16_t)(a.x & 0xffff)), bf2f((bf16_t)(a.x >> 16)),
                                            bf2f((bf16_t)(a.y & 0xffff)), bf2f((bf16_t)(a.y >> 16)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = bf2f(x[n4 * 4 + threadIdx.x]);
}


// Deterministic fold of per-workgroup partial rows (no atomics anywhere on this path: on gfx950 a chain
// of same-address device-scope fp32 atomics costs ~0.1-0.5 us per link, which dominated the first
// version of every reduction here):   out_r[i] += sum_{s<S} part_r[s*stride_r + i]   for two regions r.
// Block = (256/RL) float4 column-threads x RL row lanes, 4 independent loads in flight per thread.
struct ReduceRegion {
    const float* part;
    float* out;
    int64_t stride;  // floats between partial rows
    int64_t n4;      // float4 columns
    int nblocks;
    int overwrite;   // out = sum instead of out += sum
};
template <int RL>
__global__ __launch_bounds__(256) void reduce_rows_kernel(ReduceRegion r0, ReduceRegion r1, int S) {
    constexpr int CT = 256 / RL;
    __shared__ float4 red[RL > 1 ? 256 : 1];
    const bool first = (int)blockIdx.x < r0.nblocks;
    const ReduceRegion& r = first ? r0 : r1;
    const int blk = first ? blockIdx.x : blockIdx.x - r0.nblocks;
    const int ct = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int64_t col = (int64_t)blk * CT + ct;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < r.n4) {
        const float* base = r.part + col * 4;
        for (int s = rl; s < S; s += RL * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ss = s + u * RL;
                v[u] = ss < S ? *(const float4*)(base + (int64_t)ss * r.stride) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (RL > 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0 && col < r.n4) {
            for (int k = 1; k < RL; ++k) {
                const float4 v = red[k * CT + ct];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (rl == 0 && col < r.n4) {
        float4 o = r.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(r.out + col * 4);
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *(float4*)(r.out + col * 4) = o;
    }
}

// up to TULIP_REDUCE_REGIONS_MAX independent row reductions in one launch; a region may scatter its sums through
// the relative-position index (bias-table gradient) instead of adding them in place
struct MultiRegion {
    const float* part; float* out; const int* scatter;
    int64_t stride, n4;
    int first_block, rows, overwrite, rl, nh, LL, adam;
};
// ad: the optimizer buffers for the regions with adam = 1 -- their sums are complete gradients (overwrite regions: the one
// producer of these tensors in this backward), so the AdamW step (adamw_step4, common.h: the arithmetic of adamw_kernel) is
// taken right here and the gradient is never stored; the end-of-step AdamW skips the tensors (mask bit 1)
struct MultiRegions { MultiRegion r[TULIP_REDUCE_REGIONS_MAX]; int n; AdamRef ad; };
__global__ __launch_bounds__(256) void reduce_rows_multi_kernel(const MultiRegions R) {
    __shared__ float4 red[256];
    int i = 0;
    while (i + 1 < R.n && (int)blockIdx.x >= R.r[i + 1].first_block) ++i;
    const MultiRegion r = R.r[i];
    const int RL = r.rl, CT = 256 / RL, S = r.rows;        // RL, CT: powers of two
    const int ct = threadIdx.x & (CT - 1), rl = threadIdx.x >> (31 - __builtin_clz(CT));
    const int64_t col = (int64_t)(blockIdx.x - r.first_block) * CT + ct;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // the optimizer state of a region whose sum IS the gradient: fetched beside the partial rows, not behind them (round 4: a fold
    // launch was five or six dependent memory round trips -- 4 rows at a time, then parameter + moments -- for 17 MB: ~20 us)
    const bool stepper = r.adam && rl == 0 && col < r.n4 && !r.scatter;
    float4 pp, mm, vv;
    size_t aidx = 0;
    if (stepper) {
        const AdamRef& ad = R.ad;
        aidx = (size_t)((r.out + col * 4) - ad.g0);
        pp = ld_state(ad.p0 + aidx); mm = ld_state(ad.m0 + aidx); vv = ld_state(ad.v0 + aidx);
    }
    if (col < r.n4) {
        const float* base = r.part + col * 4;
        // 16 rows in flight per trip (a split-K fold of up to 16 slabs, or 16 x RL partial rows, is ONE round trip); summed in row order
        for (int s = rl; s < S; s += RL * 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int ss = s + u * RL;
                v[u] = ld_partial(base + (int64_t)(ss < S ? ss : rl) * r.stride);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (s + u * RL < S) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (RL > 1) {                                   // block-uniform
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0 && col < r.n4) {
            for (int k = 1; k < RL; ++k) {
                const float4 v = red[k * CT + ct];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (r.scatter && r.LL == 256) {                   // block-uniform
        // Scatter-add through the relative-position index (several (i,j) pairs share a table entry), deterministic:
        // a workgroup covers exactly one head here (RL = 4: 64 float4 columns = 256 dense values, set by the
        // launcher); the dense sums and the index go to LDS and one thread per table entry adds its pairs in index
        // order (LDS broadcast reads: ~1 us).
        __shared__ int sidx[256];
        __syncthreads();
        if (rl == 0) red[ct] = acc;
        sidx[threadIdx.x] = r.scatter[threadIdx.x];
        __syncthreads();
        const int h = blockIdx.x - r.first_block, e = threadIdx.x;
        const float* dense = (const float*)red;
        float s = 0.f;
        bool any = false;
#pragma unroll 8
        for (int ij = 0; ij < 256; ++ij) {
            const bool mine = sidx[ij] == e;
            s += mine ? dense[ij] : 0.f;
            any |= mine;
        }
        if (any) {                                     // (every entry of a relative-position table has pairs)
            if (r.overwrite) r.out[e * r.nh + h] = s;
            else r.out[e * r.nh + h] += s;
        }
        return;
    }
    if (rl == 0 && col < r.n4) {
        if (r.scatter) {
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = (int)col * 4 + e, h = fast_div(idx, r.LL), ij = idx - h * r.LL;
                atomicAdd(r.out + r.scatter[ij] * r.nh + h, v[e]);
            }
        } else {
            float4 o = r.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(r.out + col * 4);
            o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
            if (r.adam) {                               // block-uniform
                const AdamRef& ad = R.ad;
                const size_t idx = aidx;
                const bool decay = ad.mask64 ? (ad.mask64[idx >> 6] & 1u) != 0 : true;
                adamw_step4(pp, mm, vv, o, adamw_coef(ad.hyper, decay));
                st_state(ad.p0 + idx, pp); st_state(ad.m0 + idx, mm); st_state(ad.v0 + idx, vv);
                st_state_bf16x4(ad.pb0 + idx, make_uint2(pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w)));
            } else {
                *(float4*)(r.out + col * 4) = o;
            }
        }
    }
}

// ---------------------------------------------------------------- loss (tulip.py:690-700)
constexpr int LOSS_BLOCKS = 1024;
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                         float* __restrict__ partials, int64_t n, int log_transform) {
    __shared__ float red[2][4];
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = pred[i], t = tgt[i];
        s0 += fabsf(p - t);
        if (log_transform) s1 += fabsf(expm1f(p) - expm1f(t));
    }
    s0 = group_sum<64>(s0);
    s1 = group_sum<64>(s1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partials, float* __restrict__ losses,
                                                       int nblocks, double inv_n, int log_transform) {
    __shared__ double red[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { s0 += partials[i * 2]; s1 += partials[i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        losses[0] = (float)(a * inv_n);
        losses[1] = log_transform ? (float)(b * inv_n) : (float)(a * inv_n);
    }
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                     const float* __restrict__ gscale_dev, float gscale,
                                                     float* __restrict__ dpred, int64_t n) {
    const float g = (gscale_dev ? gscale_dev[0] : gscale) / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = pred[i] - tgt[i];
        dpred[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
}

// ---------------------------------------------------------------- AdamW
// hyper = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, int64_t n, const float* __restrict__ hyper,
                                                    const uint8_t* __restrict__ mask64, int zero_grad) {
    const AdamwCoef cd = adamw_coef(hyper, true), cn = adamw_coef(hyper, false);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const unsigned mk = mask64 ? mask64[i >> 4] : 1u;
        if (mk & 2u) continue;                       // this block's step was taken in a weight-gradient write-out
        float4 pp = *(float4*)(p + i * 4);
        const float4 gg = *(const float4*)(g + i * 4);
        float4 mm = *(float4*)(m + i * 4), vv = *(float4*)(v + i * 4);
        adamw_step4(pp, mm, vv, gg, (mk & 1u) ? cd : cn);
        *(float4*)(p + i * 4) = pp; *(float4*)(m + i * 4) = mm; *(float4*)(v + i * 4) = vv;
        if (zero_grad) *(float4*)(g + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pb) *(uint2*)(pb + i * 4) = make_uint2(pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w));
    }
}

// the same step over an explicit list of 64-element blocks (the few tensors left for the end of the step once everything else was
// stepped where its gradient was completed): 16 lanes per block, nothing scanned
__global__ __launch_bounds__(256) void adamw_kernel_blocks(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           bf16_t* __restrict__ pb, const int32_t* __restrict__ blocks,
                                                           int nblocks, const float* __restrict__ hyper,
                                                           const uint8_t* __restrict__ mask64, int zero_grad) {
    const AdamwCoef cd = adamw_coef(hyper, true), cn = adamw_coef(hyper, false);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < (int64_t)nblocks * 16; q += (int64_t)gridDim.x * 256) {
        const int blk = blocks[q >> 4];
        const int64_t i = (int64_t)blk * 16 + (q & 15);
        const unsigned mk = mask64 ? mask64[blk] : 1u;
        float4 pp = *(float4*)(p + i * 4);
        const float4 gg = *(const float4*)(g + i * 4);
        float4 mm = *(float4*)(m + i * 4), vv = *(float4*)(v + i * 4);
        adamw_step4(pp, mm, vv, gg, (mk & 1u) ? cd : cn);
        *(float4*)(p + i * 4) = pp; *(float4*)(m + i * 4) = mm; *(float4*)(v + i * 4) = vv;
        if (zero_grad) *(float4*)(g + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pb) *(uint2*)(pb + i * 4) = make_uint2(pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w));
    }
}

// ---------------------------------------------------------------- DropPath draws (tulip.py:25-29, timm drop_path)
// scale[slot][b] = floor(keep[slot] + u) / keep[slot], u ~ U[0,1) from a counter-based generator keyed by
// (seed, step counter, index): one launch per step, graph-replayable (the counter lives in device memory).
__global__ __launch_bounds__(256) void drop_scales_kernel(const DropDraw d) { drop_draw_block(d); }

// ---------------------------------------------------------------- gradient L2 norm (misc.py:317-329)
// fixed block -> partial mapping and a fixed fold order: the read-out is run-to-run deterministic
constexpr int NORM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, double* __restrict__ partials,
                                                            int64_t n) {
    __shared__ float red[4];
    const int64_t n4 = n >> 2;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = *(const float4*)(g + i * 4);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = g[n4 * 4 + threadIdx.x]; s += t * t; }
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partials, int nblocks,
                                                          const float* __restrict__ scale_dev, float scale,
                                                          float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partials[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = (float)(sqrt(t) * (double)(scale_dev ? scale_dev[0] * scale : scale));
    }
}

inline int grid_for(int64_t work, int cap = 256 * 8) {
    int64_t b = (work + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int tulip_cast_f32_bf16(const float* x, uint16_t* y, int rows, int cols, const float* rowscale,
                                   int rows_per_sample, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return TULIP_OK;
    if (cols & 7) return TULIP_ERR_ARG;
    if (rowscale && rows_per_sample <= 0) return TULIP_ERR_ARG;
    const int64_t n8 = (int64_t)rows * cols / 8;
    hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, x, y, n8, cols / 8, rowscale,
                       rows_per_sample);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_cast_flat(const float* x, uint16_t* y, int64_t n, hipStream_t stream) {
    if (n <= 0) return TULIP_OK;
    hipLaunchKernelGGL(cast_flat_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, x, y, n);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, hipStream_t stream) {
    if (n <= 0) return TULIP_OK;
    hipLaunchKernelGGL(cast_up_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, x, y, n);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

static int reduce_rows_impl(const float* part0, int64_t stride0, float* out0, int64_t n0, const float* part1,
                            int64_t stride1, float* out1, int64_t n1, int nrows, int overwrite, hipStream_t stream) {
    if (nrows <= 0 || (n0 <= 0 && n1 <= 0)) return TULIP_OK;
    if ((n0 & 3) || (n1 & 3) || (stride0 & 3) || (stride1 & 3)) return TULIP_ERR_ARG;
    if (n0 < 0) n0 = 0;
    if (n1 < 0) n1 = 0;
    // few columns -> spread the partial rows over 16 row lanes; many columns -> one thread per column
    const bool wide = (n0 + n1) / 4 >= 8192;
    const int CT = wide ? 256 : 16;
    ReduceRegion r0{part0, out0, stride0, n0 / 4, (int)((n0 / 4 + CT - 1) / CT), overwrite};
    ReduceRegion r1{part1, out1, stride1, n1 / 4, (int)((n1 / 4 + CT - 1) / CT), overwrite};
    const dim3 grid(r0.nblocks + r1.nblocks);
    if (wide) hipLaunchKernelGGL(reduce_rows_kernel<1>, grid, dim3(256), 0, stream, r0, r1, nrows);
    else hipLaunchKernelGGL(reduce_rows_kernel<16>, grid, dim3(256), 0, stream, r0, r1, nrows);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_reduce_rows2(const float* part0, int64_t stride0, float* out0, int64_t n0, const float* part1,
                                  int64_t stride1, float* out1, int64_t n1, int nrows, hipStream_t stream) {
    return reduce_rows_impl(part0, stride0, out0, n0, part1, stride1, out1, n1, nrows, 0, stream);
}

extern "C" int tulip_reduce_rows_multi(const tulip_reduce_region* regions, int n, hipStream_t stream) {
    return tulip_reduce_rows_multi_adamw(regions, n, nullptr, stream);
}

extern "C" int tulip_reduce_rows_multi_adamw(const tulip_reduce_region* regions, int n, const tulip_adamw_ref* adam,
                                             hipStream_t stream) {
    if (n < 0 || n > TULIP_REDUCE_REGIONS_MAX || (n && !regions)) return TULIP_ERR_ARG;
    if (adam && (!adam->hyper || !adam->grad || !adam->param || !adam->exp_avg || !adam->exp_avg_sq || !adam->param_bf16))
        return TULIP_ERR_ARG;
    MultiRegions R;
    R.ad = adam ? AdamRef{adam->hyper, adam->grad, adam->param, adam->exp_avg, adam->exp_avg_sq, (bf16_t*)adam->param_bf16,
                          adam->decay_mask64}
                : AdamRef{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    R.n = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const tulip_reduce_region& g = regions[i];
        if (g.n <= 0 || g.rows <= 0) continue;
        if ((g.n & 3) || (g.stride & 3) || !g.partials || !g.out) return TULIP_ERR_ARG;
        if (g.scatter_index && (g.scatter_nh <= 0 || g.scatter_len <= 0 || g.n != (int64_t)g.scatter_nh * g.scatter_len))
            return TULIP_ERR_ARG;
        MultiRegion& r = R.r[R.n++];
        r.part = g.partials; r.out = g.out; r.scatter = g.scatter_index; r.stride = g.stride; r.n4 = g.n / 4;
        r.rows = g.rows; r.overwrite = g.overwrite; r.nh = g.scatter_nh; r.LL = g.scatter_len;
        // the step is taken only where the sum IS the gradient: an overwrite region without a scatter, optimizer buffers given
        if (g.adamw && (!adam || !g.overwrite || g.scatter_index)) return TULIP_ERR_ARG;
        r.adam = g.adamw ? 1 : 0;
        // one thread per float4 column when there are few rows; otherwise spread the rows over 2..16 row lanes until
        // the region has enough workgroups to hide the strided loads (32-row slabs of a 37k-column weight gradient
        // took 50-80 us with one thread per column)
        r.rl = 1;
        while (r.rl < 16 && r.rl * 8 <= g.rows && (r.n4 * r.rl + 255) / 256 < 512) r.rl *= 2;
        if (g.scatter_index && g.scatter_len == 256) r.rl = 4;     // one workgroup per head: deterministic scatter
        r.first_block = blocks;
        const int ct = 256 / r.rl;
        blocks += (int)((r.n4 + ct - 1) / ct);
    }
    if (blocks == 0) return TULIP_OK;
    hipLaunchKernelGGL(reduce_rows_multi_kernel, dim3(blocks), dim3(256), 0, stream, R);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_reduce_splits(const float* slabs, float* out, int64_t n, int splits, hipStream_t stream) {
    return tulip_reduce_rows2(slabs, n, out, n, nullptr, 0, nullptr, 0, splits, stream);
}

extern "C" int tulip_l1_loss_fwd(const float* pred, const float* target, float* partials, float* losses, int64_t n,
                                 int log_transform, hipStream_t stream) {
    if (n <= 0) return TULIP_ERR_ARG;
    const int nb = grid_for(n, LOSS_BLOCKS);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, stream, pred, target, partials, n, log_transform);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, stream, partials, losses, nb, 1.0 / (double)n,
                       log_transform);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

// second stage alone: the partial sums came out of another kernel (tulip_tail_fwd_ln: one [loss | pixel loss] pair per workgroup)
extern "C" int tulip_l1_loss_final(const float* partials, float* losses, int nblocks, int64_t n, int log_transform,
                                   hipStream_t stream) {
    if (!partials || !losses || nblocks <= 0 || n <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, stream, partials, losses, nblocks, 1.0 / (double)n, log_transform);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_l1_loss_bwd(const float* pred, const float* target, const float* gscale_dev, float gscale,
                                 float* dpred, int64_t n, hipStream_t stream) {
    if (n <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, pred, target, gscale_dev, gscale, dpred,
                       n);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_adamw(float* p, float* g, float* m, float* v, uint16_t* p_bf16, int64_t n,
                           const float* hyper, const uint8_t* decay_mask64, int zero_grad, hipStream_t stream) {
    if (n <= 0) return TULIP_OK;
    if (n & 3) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, m, v, p_bf16, n, hyper,
                       decay_mask64, zero_grad);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_adamw_blocks(float* p, float* g, float* m, float* v, uint16_t* p_bf16, const int32_t* blocks, int nblocks,
                                  const float* hyper, const uint8_t* decay_mask64, int zero_grad, hipStream_t stream) {
    if (nblocks <= 0) return TULIP_OK;
    if (!p || !g || !m || !v || !blocks || !hyper) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel_blocks, dim3(grid_for((int64_t)nblocks * 16)), dim3(256), 0, stream, p, g, m, v,
                       (bf16_t*)p_bf16, blocks, nblocks, hyper, decay_mask64, zero_grad);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_drop_path_scales(const float* keep, float* scale, float* u_out, int nslots, int B,
                                      uint64_t seed, uint64_t* counter, hipStream_t stream) {
    if (!keep || !scale || !counter || nslots <= 0 || B <= 0) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(drop_scales_kernel, dim3(1), dim3(256), 0, stream,
                       DropDraw{keep, scale, u_out, nslots, B, (unsigned long long)seed, (unsigned long long*)counter});
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_grad_norm(const float* g, int64_t n, double* partials, const float* scale_dev, float scale,
                               float* out, hipStream_t stream) {
    if (n <= 0 || !partials || !out) return TULIP_ERR_ARG;
    const int nb = grid_for((n + 3) / 4, NORM_BLOCKS);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, g, partials, n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, nb, scale_dev, scale, out);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

// diagnostics: the constant-rate 100 MHz device clock (one base for every XCD, unlike s_memtime) at this point of the
// stream -- a graph node like any other, so tools/step_stamps.py can time-line a captured step without a tracer
__global__ void stamp_kernel(unsigned long long* dst) { *dst = __builtin_amdgcn_s_memrealtime(); }
extern "C" int tulip_stamp_realtime(uint64_t* dst, hipStream_t stream) {
    if (!dst) return TULIP_ERR_ARG;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long*)dst);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_abi_version(void) { return TULIP_ABI_VERSION; }
extern "C" const char* tulip_build_arch(void) { return "gfx950"; }
extern "C" int tulip_dev_variants(void) { return TULIP_DEV_VARIANTS; }
