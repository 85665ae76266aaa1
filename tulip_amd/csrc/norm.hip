// LayerNorm forward/backward over the fp32 residual stream (optionally through the PatchMerging
// 2x2 gather) and the PatchEmbedding conv+LN.  All HBM-bound: one pass over x, 16-byte accesses,
// wave shuffles for the row statistics.   gfx950 only.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "tulip_hip.h"

namespace {

struct RowGeom {
    int C;      // normalised width (4*Cin when merge)
    int merge;  // PatchMerging gather (tulip.py:92-99)
    int B, H, W;  // geometry of the *source* tensor (B,H,W,C/4) when merge
};

// pointer to 4 consecutive floats (element e..e+3 of logical row `row`)
__device__ __forceinline__ size_t src_off(const RowGeom& g, int row, int e) {
    if (!g.merge) return (size_t)row * g.C + e;
    const int cin = g.C >> 2;
    const int q = fast_div(e, cin), ci = e - q * cin;  // q: 0 (0,0) 1 (1,0) 2 (0,1) 3 (1,1)
    const int w2 = g.W >> 1, h2 = g.H >> 1;
    const int t = fast_div(row, w2), wq = row - t * w2;
    const int b = fast_div(t, h2), hq = t - b * h2;
    const int h = 2 * hq + (q & 1), w = 2 * wq + (q >> 1);
    return (((size_t)b * g.H + h) * g.W + w) * cin + ci;
}

// ------------------------------------------------------------------ forward
// LPR lanes per row, NCH float4 chunks per lane held in registers (C <= 4*LPR*NCH)
// EXACT: C == 4 LPR NCH (the launcher checked): no lane ever holds a chunk beyond the row -- no exec-masked branch around the loads
template <int LPR, int NCH, bool EXACT = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     RowGeom g, float eps) {
    const int lane = threadIdx.x % LPR;
    const int rpb = 256 / LPR;
    const int nch = g.C >> 2;
    const float invC = 1.0f / (float)g.C;
    for (int row = blockIdx.x * rpb + threadIdx.x / LPR; row < rows; row += gridDim.x * rpb) {
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                v[i] = *(const float4*)(x + src_off(g, row, c * 4));
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            } else {
                v[i] = make_float4(0, 0, 0, 0);
            }
        }
        const float mu = group_sum<LPR>(s) * invC;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                ss += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(group_sum<LPR>(ss) * invC + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                const float4 ga = *(const float4*)(gamma + c * 4);
                const float4 be = *(const float4*)(beta + c * 4);
                const float o0 = (v[i].x - mu) * rs * ga.x + be.x, o1 = (v[i].y - mu) * rs * ga.y + be.y;
                const float o2 = (v[i].z - mu) * rs * ga.z + be.z, o3 = (v[i].w - mu) * rs * ga.w + be.w;
                *(uint2*)(y + (size_t)row * g.C + c * 4) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            }
        }
    }
}

// ------------------------------------------------------------------ backward (dx)
template <int LPR, int NCH, bool EXACT = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dres, float* dx,
                                                     int rows, RowGeom g, float* __restrict__ ppart,
                                                     bf16_t* __restrict__ ycast, const float* __restrict__ cscale,
                                                     int crps, const float* __restrict__ slabs, int nslab) {
    // slabs (optional, instead of dy): the incoming gradient is still the `nslab` raw split-K partial slabs
    // [nslab][rows][C] of the data-gradient GEMM in front (tulip_gemm_bf16 with TULIP_EPI_SPLIT_F32): they are folded
    // here, in slab order, and rounded to bf16 exactly as the GEMM's own fold launch would have stored them.
    // ycast (optional): the NEXT consumer of dx on the backward chain is always a GEMM that wants
    // bf16(dx * DropPath scale of the branch it enters) -- emitted here, from registers, instead of a cast pass.
    // PARTS: also accumulate d(gamma)=sum dy*xhat and d(beta)=sum dy over this block's rows and emit one
    // partial row [2C] per block (folded later by tulip_reduce_rows2) -- no second pass over dy / x.
    constexpr bool PARTS = NCH <= 8;
    extern __shared__ __attribute__((aligned(16))) float lds_part[];  // [256/LPR][2C]
    const int lane = threadIdx.x % LPR;
    const int rpb = 256 / LPR;
    const int nch = g.C >> 2;
    const float invC = 1.0f / (float)g.C;
    float4 pg[PARTS ? NCH : 1], pb[PARTS ? NCH : 1];
#pragma unroll
    for (int i = 0; i < (PARTS ? NCH : 1); ++i) { pg[i] = make_float4(0, 0, 0, 0); pb[i] = make_float4(0, 0, 0, 0); }
    for (int row = blockIdx.x * rpb + threadIdx.x / LPR; row < rows; row += gridDim.x * rpb) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                const float4 xv = *(const float4*)(x + src_off(g, row, c * 4));
                uint2 d;
                if (slabs) {
                    const float4 a = fold_slabs4(slabs + (size_t)row * g.C + c * 4, (size_t)rows * g.C, nslab);
                    d = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
                } else {
                    d = *(const uint2*)(dy + (size_t)row * g.C + c * 4);
                }
                const float4 ga = *(const float4*)(gamma + c * 4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                const float d0 = bf2f((bf16_t)(d.x & 0xffff)), d1 = bf2f((bf16_t)(d.x >> 16));
                const float d2 = bf2f((bf16_t)(d.y & 0xffff)), d3 = bf2f((bf16_t)(d.y >> 16));
                if (PARTS) {
                    pg[i].x += d0 * xh[i].x; pg[i].y += d1 * xh[i].y; pg[i].z += d2 * xh[i].z; pg[i].w += d3 * xh[i].w;
                    pb[i].x += d0; pb[i].y += d1; pb[i].z += d2; pb[i].w += d3;
                }
                gy[i] = make_float4(d0 * ga.x, d1 * ga.y, d2 * ga.z, d3 * ga.w);
                s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
                s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
            }
        }
        const float m1 = group_sum<LPR>(s1) * invC;
        const float m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                const size_t off = src_off(g, row, c * 4);
                float4 o = make_float4(rs * (gy[i].x - m1 - xh[i].x * m2), rs * (gy[i].y - m1 - xh[i].y * m2),
                                       rs * (gy[i].z - m1 - xh[i].z * m2), rs * (gy[i].w - m1 - xh[i].w * m2));
                if (dres) {
                    const float4 r = *(const float4*)(dres + off);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *(float4*)(dx + off) = o;
                if (ycast) {
                    const int cdx = g.merge ? (g.C >> 2) : g.C;          // channels of the dx tensor
                    const float sc = cscale ? cscale[fast_div(fast_div((int)off, cdx), crps)] : 1.0f;   // off < 2^31 (checked)
                    *(uint2*)(ycast + off) = make_uint2(pack_bf16x2(o.x * sc, o.y * sc), pack_bf16x2(o.z * sc, o.w * sc));
                }
            }
        }
    }
    if (PARTS && ppart) {
        const int C2 = 2 * g.C;
        float* mine = lds_part + (threadIdx.x / LPR) * C2;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (EXACT || c < nch) {
                *(float4*)(mine + c * 4) = pg[i];
                *(float4*)(mine + g.C + c * 4) = pb[i];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < C2; idx += 256) {
            float s = 0.f;
#pragma unroll 4
            for (int k = 0; k < rpb; ++k) s += lds_part[k * C2 + idx];
            ppart[(size_t)blockIdx.x * C2 + idx] = s;
        }
    }
}

// ------------------------------------------------------------------ backward (dgamma, dbeta)
// thread (tx, ty): column chunk tx (+ blockIdx.x * TPR), rows ty, ty+RL, ... of this block's row range
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* dgamma, float* dbeta,
                                                            int rows, RowGeom g, int tpr_log2, int rows_per_block) {
    __shared__ float red[8][256];
    const int TPR = 1 << tpr_log2;
    const int tx = threadIdx.x & (TPR - 1), ty = threadIdx.x >> tpr_log2;
    const int RL = 256 >> tpr_log2;
    const int c = blockIdx.x * TPR + tx;
    const int nch = g.C >> 2;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < nch) {
        const int r0 = blockIdx.y * rows_per_block;
        const int r1 = min(rows, r0 + rows_per_block);
        constexpr int UNR = 4;
        for (int rb = r0 + ty; rb < r1; rb += RL * UNR) {
            float4 xv[UNR]; uint2 d[UNR]; float mu[UNR], rs[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = rb + u * RL;
                xv[u] = make_float4(0, 0, 0, 0); d[u] = make_uint2(0, 0); mu[u] = 0.f; rs[u] = 0.f;
                if (row < r1) {
                    mu[u] = mean[row]; rs[u] = rstd[row];
                    xv[u] = *(const float4*)(x + src_off(g, row, c * 4));
                    d[u] = *(const uint2*)(dy + (size_t)row * g.C + c * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float d0 = bf2f((bf16_t)(d[u].x & 0xffff)), d1 = bf2f((bf16_t)(d[u].x >> 16));
                const float d2 = bf2f((bf16_t)(d[u].y & 0xffff)), d3 = bf2f((bf16_t)(d[u].y >> 16));
                a[0] += d0 * (xv[u].x - mu[u]) * rs[u]; a[1] += d1 * (xv[u].y - mu[u]) * rs[u];
                a[2] += d2 * (xv[u].z - mu[u]) * rs[u]; a[3] += d3 * (xv[u].w - mu[u]) * rs[u];
                a[4] += d0; a[5] += d1; a[6] += d2; a[7] += d3;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[i][threadIdx.x] = a[i];
    __syncthreads();
    if (ty == 0 && c < nch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = 0.f;
            for (int k = 0; k < RL; ++k) s += red[i][(k << tpr_log2) + tx];
            // one block per column group (gridDim.y == 1, the launcher's choice up to 2048 rows: the engine's only use, the deepest
            // PatchMerging norm of tulip_large): a plain ordered add -- replicas and repeated runs stay bit-identical
            float* o = i < 4 ? dgamma + c * 4 + i : dbeta + c * 4 + (i - 4);
            if (gridDim.y == 1) *o += s;
            else atomicAdd(o, s);
        }
    }
}

template <typename F>
int dispatch_ln(int C, F&& f) {
    const int nch = C >> 2;
    auto go = [&](auto lpr, auto n) {
        if (nch == decltype(lpr)::value * decltype(n)::value) return f(lpr, n, std::true_type{});
        return f(lpr, n, std::false_type{});
    };
    // widths whose row is a whole number of chunks per lane get exactly that many (768 -> 3, 1536 -> 6, 3072 -> 12: the 768-wide
    // LayerNorms of the deep stage ran with 4 slots per lane, one of them empty, each behind a bounds branch)
    if (nch == 64 * 3) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 3>{}, std::true_type{});
    if (nch == 64 * 6) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 6>{}, std::true_type{});
    if (nch == 64 * 12) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 12>{}, std::true_type{});
    if (nch <= 16 * 2) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
    if (nch <= 16 * 4) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
    if (nch <= 64 * 2) return go(std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
    if (nch <= 64 * 4) return go(std::integral_constant<int, 64>{}, std::integral_constant<int, 4>{});
    if (nch <= 64 * 8) return go(std::integral_constant<int, 64>{}, std::integral_constant<int, 8>{});
    if (nch <= 64 * 24) return go(std::integral_constant<int, 64>{}, std::integral_constant<int, 24>{});
    return TULIP_ERR_ARG;
}

bool geom_ok(int rows, int C, int merge, int B, int H, int W) {
    if (C <= 0 || (C & 3)) return false;
    if (merge) {
        if ((C & 15) || (H & 1) || (W & 1)) return false;
        if (rows != B * (H / 2) * (W / 2)) return false;
    }
    return true;
}

// ------------------------------------------------------------------ patch embedding
struct EmbedGeom {
    int B, Cin, Hin, Win, E, p0, p1, kw, circular, Ho, Wo, taps;
};

__device__ __forceinline__ float embed_tap(const float* __restrict__ img, const EmbedGeom& g, int b, int h, int w,
                                           int t) {
    const int tt = fast_div(t, g.kw), k = t - tt * g.kw;
    const int ic = fast_div(tt, g.p0), i = tt - ic * g.p0;
    int col = g.p1 * w + k;
    if (g.circular) { col -= 2; if (col < 0) col += g.Win; if (col >= g.Win) col -= g.Win; }
    return img[(((size_t)b * g.Cin + ic) * g.Hin + (g.p0 * h + i)) * g.Win + col];
}

constexpr int EMB_MAXT = 16;  // taps held in registers: (1,4) patches (4 or 8 taps) and (4,4) patches (16)

// all taps of one token into registers with independent loads (taps <= EMB_MAXT)
__device__ __forceinline__ void embed_taps(const float* __restrict__ img, const EmbedGeom& g, int b, int h, int w,
                                           float (&xt)[EMB_MAXT]) {
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) xt[t] = (t < g.taps) ? embed_tap(img, g, b, h, w, t) : 0.f;
}

// one wave per token (grid-stride), lane owns channels lane and lane+64 (E <= 128); conv weights of the
// lane's two channels live in registers for the whole kernel.
template <bool INREG>
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ out,
                                                              bf16_t* __restrict__ out16, int ld16, EmbedGeom g,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int ntok = g.B * g.Ho * g.Wo;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < g.E, v1 = c1 < g.E;
    const float invE = 1.0f / (float)g.E;
    float w0[EMB_MAXT], w1[EMB_MAXT];
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) {
        w0[t] = (INREG && v0 && t < g.taps) ? w[c0 * g.taps + t] : 0.f;
        w1[t] = (INREG && v1 && t < g.taps) ? w[c1 * g.taps + t] : 0.f;
    }
    const float b0 = v0 ? bias[c0] : 0.f, b1 = v1 ? bias[c1] : 0.f;
    const float ga0 = v0 ? gamma[c0] : 0.f, ga1 = v1 ? gamma[c1] : 0.f;
    const float be0 = v0 ? beta[c0] : 0.f, be1 = v1 ? beta[c1] : 0.f;
    for (int tok = wave; tok < ntok; tok += nwaves) {
        const int t2 = fast_div(tok, g.Wo), wq = tok - t2 * g.Wo;
        const int b = fast_div(t2, g.Ho), h = t2 - b * g.Ho;
        float a0 = b0, a1 = b1;
        if (INREG) {
            float xt[EMB_MAXT];
            embed_taps(img, g, b, h, wq, xt);
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) { a0 += w0[t] * xt[t]; a1 += w1[t] * xt[t]; }
        } else {
            for (int t = 0; t < g.taps; ++t) {
                const float xv = embed_tap(img, g, b, h, wq, t);
                if (v0) a0 += w[c0 * g.taps + t] * xv;
                if (v1) a1 += w[c1 * g.taps + t] * xv;
            }
        }
        const float mu = group_sum<64>((v0 ? a0 : 0.f) + (v1 ? a1 : 0.f)) * invE;
        const float d0 = v0 ? a0 - mu : 0.f, d1 = v1 ? a1 - mu : 0.f;
        const float rs = rsqrtf(group_sum<64>(d0 * d0 + d1 * d1) * invE + eps);
        const float y0 = d0 * rs * ga0 + be0, y1 = d1 * rs * ga1 + be1;
        if (v0) out[(size_t)tok * g.E + c0] = y0;
        if (v1) out[(size_t)tok * g.E + c1] = y1;
        if (out16) {                                  // bf16 copy (x_save half of the first skip concat)
            if (v0) out16[(size_t)tok * ld16 + c0] = f2bf(y0);
            if (v1) out16[(size_t)tok * ld16 + c1] = f2bf(y1);
        }
    }
}

// Single-channel images with one-row patches (every range-image configuration of the reference: in_chans 1, patch
// (1,4), KW = 8 taps with circular padding or 4 without): a tap is just a column offset, and a wave works on 4
// consecutive tokens at once so that their loads / two LayerNorm reductions / stores are 4 independent chains (the
// generic kernel spends most of its 30 us in per-tap index arithmetic and one dependent chain per token).
template <int KW>
__global__ __launch_bounds__(256) void patch_embed_fwd_row_kernel(const float* __restrict__ img,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ out,
                                                                  bf16_t* __restrict__ out16, int ld16, EmbedGeom g,
                                                                  float eps, const DropDraw dd) {
    constexpr int UNR = 4;
    if (dd.keep && blockIdx.x == 0) drop_draw_block(dd);         // (block-uniform) the step's DropPath draws: nobody in this launch reads them
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int ntok = g.B * g.Ho * g.Wo;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < g.E, v1 = c1 < g.E;
    const float invE = 1.0f / (float)g.E;
    float w0[KW], w1[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        w0[t] = v0 ? w[c0 * KW + t] : 0.f;
        w1[t] = v1 ? w[c1 * KW + t] : 0.f;
    }
    const float b0 = v0 ? bias[c0] : 0.f, b1 = v1 ? bias[c1] : 0.f;
    const float ga0 = v0 ? gamma[c0] : 0.f, ga1 = v1 ? gamma[c1] : 0.f;
    const float be0 = v0 ? beta[c0] : 0.f, be1 = v1 ? beta[c1] : 0.f;
    for (int tok0 = wave * UNR; tok0 < ntok; tok0 += nwaves * UNR) {
        float xt[UNR][KW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tok = min(tok0 + u, ntok - 1);
            const int trow = fast_div(tok, g.Wo), wq = tok - trow * g.Wo;
            const float* rowp = img + (size_t)trow * g.Win;                  // Cin == 1, p0 == 1: image row = b*Ho + h
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                int col = g.p1 * wq + k;
                if (g.circular) { col -= 2; if (col < 0) col += g.Win; if (col >= g.Win) col -= g.Win; }
                xt[u][k] = rowp[col];
            }
        }
        float a0[UNR], a1[UNR], mu[UNR], rs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            a0[u] = b0; a1[u] = b1;
#pragma unroll
            for (int k = 0; k < KW; ++k) { a0[u] += w0[k] * xt[u][k]; a1[u] += w1[k] * xt[u][k]; }
            mu[u] = (v0 ? a0[u] : 0.f) + (v1 ? a1[u] : 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) mu[u] = group_sum<64>(mu[u]) * invE;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            a0[u] = v0 ? a0[u] - mu[u] : 0.f; a1[u] = v1 ? a1[u] - mu[u] : 0.f;
            rs[u] = a0[u] * a0[u] + a1[u] * a1[u];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) rs[u] = rsqrtf(group_sum<64>(rs[u]) * invE + eps);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tok = tok0 + u;
            if (tok >= ntok) break;
            const float y0 = a0[u] * rs[u] * ga0 + be0, y1 = a1[u] * rs[u] * ga1 + be1;
            if (v0) out[(size_t)tok * g.E + c0] = y0;
            if (v1) out[(size_t)tok * g.E + c1] = y1;
            if (out16) {
                if (v0) out16[(size_t)tok * ld16 + c0] = f2bf(y0);
                if (v1) out16[(size_t)tok * ld16 + c1] = f2bf(y1);
            }
        }
    }
}

template <bool INREG>
__global__ __launch_bounds__(256) void patch_embed_bwd_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ dout, float* dw, float* db,
                                                              float* dgamma, float* dbeta, EmbedGeom g, float eps,
                                                              int pstride) {
    // pstride > 0: dw/db/dgamma/dbeta address row 0 of a [gridDim.x][pstride] partial buffer (plain stores,
    // folded by tulip_reduce_rows2); pstride == 0: accumulate atomically into the gradients themselves.
    __shared__ float red[4][64][2 * (EMB_MAXT + 3)];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
    const int ntok = g.B * g.Ho * g.Wo;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < g.E, v1 = c1 < g.E;
    const float invE = 1.0f / (float)g.E;
    float w0[EMB_MAXT], w1[EMB_MAXT], aw0[EMB_MAXT], aw1[EMB_MAXT];
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) {
        w0[t] = (INREG && v0 && t < g.taps) ? w[c0 * g.taps + t] : 0.f;
        w1[t] = (INREG && v1 && t < g.taps) ? w[c1 * g.taps + t] : 0.f;
        aw0[t] = 0.f; aw1[t] = 0.f;
    }
    float ab0 = 0.f, ab1 = 0.f, ag0 = 0.f, ag1 = 0.f, abe0 = 0.f, abe1 = 0.f;
    const float b0 = v0 ? bias[c0] : 0.f, b1 = v1 ? bias[c1] : 0.f;
    const float ga0 = v0 ? gamma[c0] : 0.f, ga1 = v1 ? gamma[c1] : 0.f;
    for (int tok = wave; tok < ntok; tok += nwaves) {
        const int t2 = fast_div(tok, g.Wo), wq = tok - t2 * g.Wo;
        const int b = fast_div(t2, g.Ho), h = t2 - b * g.Ho;
        const float dy0 = v0 ? dout[(size_t)tok * g.E + c0] : 0.f, dy1 = v1 ? dout[(size_t)tok * g.E + c1] : 0.f;
        float xt[EMB_MAXT];
        float a0 = b0, a1 = b1;
        if (INREG) {
            embed_taps(img, g, b, h, wq, xt);
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) { a0 += w0[t] * xt[t]; a1 += w1[t] * xt[t]; }
        } else {
            for (int t = 0; t < g.taps; ++t) {
                const float xv = embed_tap(img, g, b, h, wq, t);
                if (v0) a0 += w[c0 * g.taps + t] * xv;
                if (v1) a1 += w[c1 * g.taps + t] * xv;
            }
        }
        const float mu = group_sum<64>((v0 ? a0 : 0.f) + (v1 ? a1 : 0.f)) * invE;
        const float d0 = v0 ? a0 - mu : 0.f, d1 = v1 ? a1 - mu : 0.f;
        const float rs = rsqrtf(group_sum<64>(d0 * d0 + d1 * d1) * invE + eps);
        const float xh0 = d0 * rs, xh1 = d1 * rs;
        ag0 += dy0 * xh0; ag1 += dy1 * xh1; abe0 += dy0; abe1 += dy1;
        const float gy0 = dy0 * ga0, gy1 = dy1 * ga1;
        const float m1 = group_sum<64>(gy0 + gy1) * invE;
        const float m2 = group_sum<64>(gy0 * xh0 + gy1 * xh1) * invE;
        const float dc0 = v0 ? rs * (gy0 - m1 - xh0 * m2) : 0.f, dc1 = v1 ? rs * (gy1 - m1 - xh1 * m2) : 0.f;
        ab0 += dc0; ab1 += dc1;
        if (INREG) {
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) { aw0[t] += dc0 * xt[t]; aw1[t] += dc1 * xt[t]; }
        } else {
            for (int t = 0; t < g.taps; ++t) {
                const float xv = embed_tap(img, g, b, h, wq, t);
                if (v0) atomicAdd(dw + c0 * g.taps + t, dc0 * xv);
                if (v1) atomicAdd(dw + c1 * g.taps + t, dc1 * xv);
            }
        }
    }
    // block-level reduce over the 4 waves, then one atomic per value per block
    float* r = red[wid][lane];
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) { r[t] = aw0[t]; r[EMB_MAXT + 3 + t] = aw1[t]; }
    r[EMB_MAXT] = ab0; r[EMB_MAXT + 1] = ag0; r[EMB_MAXT + 2] = abe0;
    r[2 * EMB_MAXT + 3] = ab1; r[2 * EMB_MAXT + 4] = ag1; r[2 * EMB_MAXT + 5] = abe1;
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = half ? c1 : c0;
            if (c >= g.E) continue;
            const int o = half * (EMB_MAXT + 3);
            float acc[EMB_MAXT + 3];
#pragma unroll
            for (int k = 0; k < EMB_MAXT + 3; ++k)
                acc[k] = (red[0][lane][o + k] + red[1][lane][o + k]) + (red[2][lane][o + k] + red[3][lane][o + k]);
            if (pstride > 0) {
                const size_t ro = (size_t)blockIdx.x * pstride;
                if (INREG) {
#pragma unroll
                    for (int t = 0; t < EMB_MAXT; ++t) if (t < g.taps) dw[ro + c * g.taps + t] = acc[t];
                }
                db[ro + c] = acc[EMB_MAXT]; dgamma[ro + c] = acc[EMB_MAXT + 1]; dbeta[ro + c] = acc[EMB_MAXT + 2];
            } else {
                if (INREG) {
#pragma unroll
                    for (int t = 0; t < EMB_MAXT; ++t) if (t < g.taps) atomicAdd(dw + c * g.taps + t, acc[t]);
                }
                atomicAdd(db + c, acc[EMB_MAXT]); atomicAdd(dgamma + c, acc[EMB_MAXT + 1]);
                atomicAdd(dbeta + c, acc[EMB_MAXT + 2]);
            }
        }
    }
}

// ---- 16 lanes per token, 4 tokens per wave (E % 16 == 0, E <= 128, taps <= TAPS): lane `sub` owns the
// CPL = E/16 consecutive channels sub*CPL .. +CPL-1, so a token's row is written as 16 contiguous pieces and the
// LayerNorm reductions are 4-step group sums; four tokens advance through the dependent shuffle chains together.
template <int CPL, int TAPS>
__global__ __launch_bounds__(256) void patch_embed_bwd16_kernel(const float* __restrict__ img,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ dout, float* dw, float* db,
                                                                float* dgamma, float* dbeta, EmbedGeom g, float eps,
                                                                int pstride) {
    // partial-row mode only (pstride > 0): row blockIdx.x of a [gridDim.x][pstride] buffer, folded by the caller
    __shared__ float red[4][16 * CPL][TAPS + 3];           // [wave][channel][dw taps | db dgamma dbeta]
    const int lane = threadIdx.x & 63, sub = lane & 15, slot = lane >> 4, wid = threadIdx.x >> 6;
    const int group = (blockIdx.x * 4 + wid) * 4 + slot, ngroups = gridDim.x * 16;
    const int ntok = g.B * g.Ho * g.Wo;
    const float invE = 1.0f / (float)g.E;
    float wr[CPL][TAPS], br[CPL], gar[CPL];
    float aw[CPL][TAPS], ab[CPL], ag[CPL], abe[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = sub * CPL + k;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) { wr[k][t] = t < g.taps ? w[c * g.taps + t] : 0.f; aw[k][t] = 0.f; }
        br[k] = bias[c]; gar[k] = gamma[c]; ab[k] = 0.f; ag[k] = 0.f; abe[k] = 0.f;
    }
    for (int tok = group; tok < ntok; tok += ngroups) {
        const int t2 = fast_div(tok, g.Wo), wq = tok - t2 * g.Wo;
        const int b = fast_div(t2, g.Ho), h = t2 - b * g.Ho;
        float xt[TAPS];
        #pragma unroll
        for (int t = 0; t < TAPS; ++t) xt[t] = (t < g.taps) ? embed_tap(img, g, b, h, wq, t) : 0.f;
        const float* dyp = dout + (size_t)tok * g.E + sub * CPL;
        float a[CPL], dy[CPL], s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            dy[k] = dyp[k];
            a[k] = br[k];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) a[k] += wr[k][t] * xt[t];
            s += a[k];
        }
        const float mu = group_sum<16>(s) * invE;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { a[k] -= mu; q += a[k] * a[k]; }
        const float rs = rsqrtf(group_sum<16>(q) * invE + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            a[k] *= rs;                                   // xhat
            ag[k] += dy[k] * a[k]; abe[k] += dy[k];
            dy[k] *= gar[k];                              // gy
            s1 += dy[k]; s2 += dy[k] * a[k];
        }
        const float m1 = group_sum<16>(s1) * invE, m2 = group_sum<16>(s2) * invE;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const float dc = rs * (dy[k] - m1 - a[k] * m2);
            ab[k] += dc;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) aw[k][t] += dc * xt[t];
        }
    }
    // fold the wave's 4 token slots (lanes sub, sub+16, sub+32, sub+48), then the 4 waves through LDS
    auto fold4 = [](float v) { return rows_sum(v); };
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) aw[k][t] = fold4(aw[k][t]);
        ab[k] = fold4(ab[k]); ag[k] = fold4(ag[k]); abe[k] = fold4(abe[k]);
    }
    if (slot == 0) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            float* r = red[wid][sub * CPL + k];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) r[t] = aw[k][t];
            r[TAPS] = ab[k]; r[TAPS + 1] = ag[k]; r[TAPS + 2] = abe[k];
        }
    }
    __syncthreads();
    const size_t ro = (size_t)blockIdx.x * pstride;
    for (int i = threadIdx.x; i < g.E * (TAPS + 3); i += 256) {
        const int c = i / (TAPS + 3), v = i - c * (TAPS + 3);
        const float acc = (red[0][c][v] + red[1][c][v]) + (red[2][c][v] + red[3][c][v]);
        if (v < TAPS) { if (v < g.taps) dw[ro + c * g.taps + v] = acc; }
        else if (v == TAPS) db[ro + c] = acc;
        else if (v == TAPS + 1) dgamma[ro + c] = acc;
        else dbeta[ro + c] = acc;
    }
}

bool embed_geom(EmbedGeom& g, int B, int Cin, int Hin, int Win, int E, int p0, int p1, int kw, int circular) {
    if (E <= 0 || E > 128 || p0 <= 0 || p1 <= 0 || Hin % p0 || Win % p1) return false;
    if (circular && kw != p1 + 4) return false;   // kernel (p0, 8) over a (2,2)-padded row with stride 4 (tulip.py:41,60)
    if (!circular && kw != p1) return false;
    g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.E = E; g.p0 = p0; g.p1 = p1; g.kw = kw; g.circular = circular;
    g.Ho = Hin / p0; g.Wo = Win / p1; g.taps = Cin * p0 * kw;
    return true;
}

}  // namespace

extern "C" int tulip_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean,
                                   float* rstd, int rows, int C, float eps, int merge, int B, int H, int W,
                                   hipStream_t stream) {
    if (rows <= 0) return TULIP_OK;
    if (!geom_ok(rows, C, merge, B, H, W)) return TULIP_ERR_ARG;
    RowGeom g{C, merge, B, H, W};
    return dispatch_ln(C, [&](auto lpr, auto nch, auto exact) {
        constexpr int LPR = decltype(lpr)::value, NCH = decltype(nch)::value;
        const int rpb = 256 / LPR;
        const int grid = std::min((rows + rpb - 1) / rpb, 256 * 16);
        hipLaunchKernelGGL((ln_fwd_kernel<LPR, NCH, decltype(exact)::value>), dim3(grid), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd,
                           rows, g, eps);
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    });
}

// ------------------------------------------------------------------ split-K fold + residual epilogue + LayerNorm
// One wave per output row of a GEMM that left raw split-K slabs (TULIP_EPI_SPLIT_F32): fold the slabs in slab order, add the
// bias, form the residual output exactly as tulip_gemm_bf16's own fold launch would (out = aux + rowscale * v), and
// normalise the row it already holds -- the LayerNorm that follows the proj / fc2 Linear of an unfused Swin block
// (tulip.py:344-347) costs no launch of its own.
template <int NCH>
__global__ __launch_bounds__(256) void splitk_resid_ln_kernel(const float* __restrict__ slabs, int nslab, int M, int N,
                                                              const float* __restrict__ bias, const float* __restrict__ aux,
                                                              int ldaux, const float* __restrict__ rowscale, int rps,
                                                              float* __restrict__ out, int ldo, bf16_t* __restrict__ out_bf16,
                                                              int ldo2, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, bf16_t* __restrict__ ln_out,
                                                              float* __restrict__ mean, float* __restrict__ rstd, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float sc = rowscale ? rowscale[fast_div(row, rps)] : 1.0f;
    float4 v[NCH];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + i * 64) * 4;
        float4 a = fold_slabs4(slabs + (size_t)row * N + c, (size_t)M * N, nslab);
        if (bias) {
            const float4 b = *(const float4*)(bias + c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (aux) {
            const float4 q = *(const float4*)(aux + (size_t)row * ldaux + c);
            a = make_float4(q.x + sc * a.x, q.y + sc * a.y, q.z + sc * a.z, q.w + sc * a.w);
        }
        *(float4*)(out + (size_t)row * ldo + c) = a;
        if (out_bf16) *(uint2*)(out_bf16 + (size_t)row * ldo2 + c) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
        v[i] = a;
        s1 += (a.x + a.y) + (a.z + a.w);
    }
    const float mu = group_sum<64>(s1) / (float)N;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const float d0 = v[i].x - mu, d1 = v[i].y - mu, d2 = v[i].z - mu, d3 = v[i].w - mu;
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rs = rsqrtf(group_sum<64>(s2) / (float)N + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + i * 64) * 4;
        const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
        *(uint2*)(ln_out + (size_t)row * N + c) =
            make_uint2(pack_bf16x2((v[i].x - mu) * rs * ga.x + be.x, (v[i].y - mu) * rs * ga.y + be.y),
                       pack_bf16x2((v[i].z - mu) * rs * ga.z + be.z, (v[i].w - mu) * rs * ga.w + be.w));
    }
}

// exactly the widths the launcher below instantiates (N / 256 in {1, 2, 3, 4, 6, 8}): a caller that gets "no" falls back to the
// two-launch form instead of an argument error mid-forward
extern "C" int tulip_splitk_resid_ln_supported(int N) {
    if (N <= 0 || N % 256 != 0) return 0;
    const int q = N / 256;
    return q == 1 || q == 2 || q == 3 || q == 4 || q == 6 || q == 8;
}

extern "C" int tulip_splitk_resid_ln(const float* slabs, int nslab, int M, int N, const float* bias, const float* aux,
                                     int ldaux, const float* rowscale, int rows_per_sample, float* out, int ldo,
                                     uint16_t* out_bf16, int ldo2, const float* gamma, const float* beta, uint16_t* ln_out,
                                     float* mean, float* rstd, float eps, hipStream_t stream) {
    if (M <= 0) return TULIP_OK;
    if (!slabs || nslab < 1 || !tulip_splitk_resid_ln_supported(N) || !out || !gamma || !beta || !ln_out || !mean || !rstd ||
        (ldo & 3) || (aux && (ldaux & 3)) || (out_bf16 && (ldo2 & 3)))
        return TULIP_ERR_ARG;
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    const dim3 grid((M + 3) / 4);
#define TULIP_SKLN(NCH)                                                                                                     \
    hipLaunchKernelGGL((splitk_resid_ln_kernel<NCH>), grid, dim3(256), 0, stream, slabs, nslab, M, N, bias, aux, ldaux, rowscale, \
                       rps, out, ldo, (bf16_t*)out_bf16, ldo2, gamma, beta, (bf16_t*)ln_out, mean, rstd, eps)
    switch (N / 256) {
        case 1: TULIP_SKLN(1); break;
        case 2: TULIP_SKLN(2); break;
        case 3: TULIP_SKLN(3); break;
        case 4: TULIP_SKLN(4); break;
        case 6: TULIP_SKLN(6); break;
        case 8: TULIP_SKLN(8); break;
        default: return TULIP_ERR_ARG;
    }
#undef TULIP_SKLN
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

// partial rows ([2C] each) tulip_layernorm_bwd emits for (rows, C); 0 = fused parameter partials unsupported
static int ln_bwd_part_rows(int rows, int C) {
    const int nch = C >> 2;
    if (nch > 64 * 8) return 0;
    const int rpb = nch <= 64 ? 16 : 4;
    return std::max(1, std::min((rows + rpb - 1) / rpb, 512));
}

extern "C" int tulip_layernorm_bwd_partial_rows(int rows, int C) { return rows > 0 ? ln_bwd_part_rows(rows, C) : 0; }

static int layernorm_bwd_impl(const uint16_t* dy, const float* slabs, int nslab, const float* x, const float* mean,
                              const float* rstd, const float* gamma, const float* dres, float* dx, int rows, int C, int merge,
                              int B, int H, int W, float* param_partials, uint16_t* dx_bf16, const float* cast_rowscale,
                              int cast_rows_per_sample, hipStream_t stream);

extern "C" int tulip_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd,
                                   const float* gamma, const float* dres, float* dx, int rows, int C, int merge, int B,
                                   int H, int W, float* param_partials, uint16_t* dx_bf16, const float* cast_rowscale,
                                   int cast_rows_per_sample, hipStream_t stream) {
    return layernorm_bwd_impl(dy, nullptr, 0, x, mean, rstd, gamma, dres, dx, rows, C, merge, B, H, W, param_partials, dx_bf16,
                              cast_rowscale, cast_rows_per_sample, stream);
}

extern "C" int tulip_layernorm_bwd_splitk(const float* slabs, int nslab, const float* x, const float* mean, const float* rstd,
                                          const float* gamma, const float* dres, float* dx, int rows, int C, int merge,
                                          int B, int H, int W, float* param_partials, uint16_t* dx_bf16,
                                          const float* cast_rowscale, int cast_rows_per_sample, hipStream_t stream) {
    if (!slabs || nslab < 1) return TULIP_ERR_ARG;
    return layernorm_bwd_impl(nullptr, slabs, nslab, x, mean, rstd, gamma, dres, dx, rows, C, merge, B, H, W, param_partials,
                              dx_bf16, cast_rowscale, cast_rows_per_sample, stream);
}

static int layernorm_bwd_impl(const uint16_t* dy, const float* slabs, int nslab, const float* x, const float* mean,
                              const float* rstd, const float* gamma, const float* dres, float* dx, int rows, int C, int merge,
                              int B, int H, int W, float* param_partials, uint16_t* dx_bf16, const float* cast_rowscale,
                              int cast_rows_per_sample, hipStream_t stream) {
    if (rows <= 0) return TULIP_OK;
    if (!geom_ok(rows, C, merge, B, H, W)) return TULIP_ERR_ARG;
    if (param_partials && ln_bwd_part_rows(rows, C) == 0) return TULIP_ERR_ARG;
    if (dx_bf16 && cast_rowscale && (int64_t)rows * C >= (int64_t)1 << 31) return TULIP_ERR_ARG;   // 32-bit token index
    RowGeom g{C, merge, B, H, W};
    return dispatch_ln(C, [&](auto lpr, auto nch, auto exact) {
        constexpr int LPR = decltype(lpr)::value, NCH = decltype(nch)::value;
        const int rpb = 256 / LPR;
        const int grid = param_partials ? ln_bwd_part_rows(rows, C) : std::min((rows + rpb - 1) / rpb, 256 * 16);
        const size_t lds = param_partials ? (size_t)rpb * 2 * C * sizeof(float) : 0;
        hipLaunchKernelGGL((ln_bwd_kernel<LPR, NCH, decltype(exact)::value>), dim3(grid), dim3(256), lds, stream, dy, x, mean, rstd, gamma,
                           dres, dx, rows, g, param_partials, dx_bf16, cast_rowscale,
                           cast_rows_per_sample > 0 ? cast_rows_per_sample : 1, slabs, nslab);
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    });
}

extern "C" int tulip_layernorm_bwd_params(const uint16_t* dy, const float* x, const float* mean, const float* rstd,
                                          float* dgamma, float* dbeta, int rows, int C, int merge, int B, int H, int W,
                                          hipStream_t stream) {
    if (rows <= 0) return TULIP_OK;
    if (!geom_ok(rows, C, merge, B, H, W)) return TULIP_ERR_ARG;
    RowGeom g{C, merge, B, H, W};
    const int nch = C >> 2;
    int tpr_log2 = 4;
    while ((1 << tpr_log2) < nch && tpr_log2 < 6) ++tpr_log2;
    const int TPR = 1 << tpr_log2;
    const int gx = (nch + TPR - 1) / TPR;
    const int RLh = 256 >> tpr_log2;
    int gy = rows <= 2048 ? 1 : std::max(1, std::min((rows + RLh * 4 - 1) / (RLh * 4), 2048 / gx));
    const int rows_per_block = (rows + gy - 1) / gy;
    gy = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(gx, gy), dim3(256), 0, stream, dy, x, mean, rstd, dgamma, dbeta, rows,
                       g, tpr_log2, rows_per_block);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_patch_embed_fwd(const float* img, const float* w, const float* b, const float* gamma,
                                     const float* beta, float* out, int B, int Cin, int Hin, int Win, int E, int p0,
                                     int p1, int kw, int circular, float eps, uint16_t* out_bf16, int ld_bf16,
                                     hipStream_t stream) {
    return tulip_patch_embed_fwd_draw(img, w, b, gamma, beta, out, B, Cin, Hin, Win, E, p0, p1, kw, circular, eps, out_bf16, ld_bf16,
                                      nullptr, stream);
}

extern "C" int tulip_patch_embed_fwd_draw(const float* img, const float* w, const float* b, const float* gamma,
                                          const float* beta, float* out, int B, int Cin, int Hin, int Win, int E, int p0,
                                          int p1, int kw, int circular, float eps, uint16_t* out_bf16, int ld_bf16,
                                          const tulip_drop_draw* draw, hipStream_t stream) {
    EmbedGeom g;
    if (!embed_geom(g, B, Cin, Hin, Win, E, p0, p1, kw, circular)) return TULIP_ERR_ARG;
    if (draw && (!draw->keep || !draw->scale || !draw->counter || draw->nslots <= 0 || draw->B <= 0)) return TULIP_ERR_ARG;
    const int ntok = g.B * g.Ho * g.Wo;
    const bool row = g.Cin == 1 && g.p0 == 1 && g.taps == g.kw && (g.kw == 8 || g.kw == 4) && E <= 128 && ntok > 0;
    DropDraw dd{nullptr, nullptr, nullptr, 0, 0, 0ull, nullptr};
    if (draw && row) {
        dd = DropDraw{draw->keep, draw->scale, draw->u_out, draw->nslots, draw->B, (unsigned long long)draw->seed,
                      (unsigned long long*)draw->counter};
    } else if (draw) {           // the generic kernels: the draw keeps its own launch
        const int rc = tulip_drop_path_scales(draw->keep, draw->scale, draw->u_out, draw->nslots, draw->B, draw->seed, draw->counter,
                                              stream);
        if (rc != TULIP_OK) return rc;
    }
    if (ntok <= 0) return TULIP_OK;
    const int grid = std::min((ntok + 3) / 4, 256 * 8);
    if (row) {
        const int grid4 = std::min((ntok + 15) / 16, 256 * 2);   // 8 waves per CU, each amortises its weight loads
        if (g.kw == 8)
            hipLaunchKernelGGL(patch_embed_fwd_row_kernel<8>, dim3(grid4), dim3(256), 0, stream, img, w, b, gamma, beta,
                               out, (bf16_t*)out_bf16, ld_bf16, g, eps, dd);
        else
            hipLaunchKernelGGL(patch_embed_fwd_row_kernel<4>, dim3(grid4), dim3(256), 0, stream, img, w, b, gamma, beta,
                               out, (bf16_t*)out_bf16, ld_bf16, g, eps, dd);
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    }
    if (g.taps <= EMB_MAXT)
        hipLaunchKernelGGL(patch_embed_fwd_kernel<true>, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, beta, out,
                           (bf16_t*)out_bf16, ld_bf16, g, eps);
    else
        hipLaunchKernelGGL(patch_embed_fwd_kernel<false>, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, beta, out,
                           (bf16_t*)out_bf16, ld_bf16, g, eps);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_patch_embed_bwd_blocks(int ntok) { return std::max(1, std::min((ntok + 3) / 4, 512)); }

extern "C" int tulip_patch_embed_bwd(const float* img, const float* w, const float* b, const float* gamma,
                                     const float* dout, float* dw, float* db, float* dgamma, float* dbeta, int B,
                                     int Cin, int Hin, int Win, int E, int p0, int p1, int kw, int circular, float eps,
                                     int partial_stride, hipStream_t stream) {
    EmbedGeom g;
    if (!embed_geom(g, B, Cin, Hin, Win, E, p0, p1, kw, circular)) return TULIP_ERR_ARG;
    const int ntok = g.B * g.Ho * g.Wo;
    if (ntok <= 0) return TULIP_OK;
    if (partial_stride > 0 && g.taps > EMB_MAXT) return TULIP_ERR_ARG;
    const int grid = tulip_patch_embed_bwd_blocks(ntok);
    if (partial_stride > 0 && g.taps <= EMB_MAXT && (E == 96 || E == 48)) {
#define TULIP_PE_BWD(CPL, TAPS) hipLaunchKernelGGL((patch_embed_bwd16_kernel<CPL, TAPS>), dim3(grid), dim3(256), 0, stream, \
                                                  img, w, b, gamma, dout, dw, db, dgamma, dbeta, g, eps, partial_stride)
        if (g.taps <= 8) { if (E == 96) TULIP_PE_BWD(6, 8); else TULIP_PE_BWD(3, 8); }
        else { if (E == 96) TULIP_PE_BWD(6, 16); else TULIP_PE_BWD(3, 16); }
#undef TULIP_PE_BWD
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    }
    if (g.taps <= EMB_MAXT)
        hipLaunchKernelGGL(patch_embed_bwd_kernel<true>, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, dout, dw,
                           db, dgamma, dbeta, g, eps, partial_stride);
    else
        hipLaunchKernelGGL(patch_embed_bwd_kernel<false>, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, dout, dw,
                           db, dgamma, dbeta, g, eps, 0);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
