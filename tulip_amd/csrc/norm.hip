// LayerNorm forward/backward over the fp32 residual stream (optionally through the PatchMerging
// 2x2 gather) and the PatchEmbedding conv+LN.  All HBM-bound: one pass over x, 16-byte accesses,
// wave shuffles for the row statistics.   gfx950 only.
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
    const int q = e / cin, ci = e - q * cin;  // q: 0 (0,0) 1 (1,0) 2 (0,1) 3 (1,1)
    const int w2 = g.W >> 1, h2 = g.H >> 1;
    const int wq = row % w2, t = row / w2;
    const int hq = t % h2, b = t / h2;
    const int h = 2 * hq + (q & 1), w = 2 * wq + (q >> 1);
    return (((size_t)b * g.H + h) * g.W + w) * cin + ci;
}

// ------------------------------------------------------------------ forward
// LPR lanes per row, NCH float4 chunks per lane held in registers (C <= 4*LPR*NCH)
template <int LPR, int NCH>
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
            if (c < nch) {
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
            if (c < nch) {
                float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                ss += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(group_sum<LPR>(ss) * invC + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (c < nch) {
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
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dres, float* dx,
                                                     int rows, RowGeom g) {
    const int lane = threadIdx.x % LPR;
    const int rpb = 256 / LPR;
    const int nch = g.C >> 2;
    const float invC = 1.0f / (float)g.C;
    for (int row = blockIdx.x * rpb + threadIdx.x / LPR; row < rows; row += gridDim.x * rpb) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (c < nch) {
                const float4 xv = *(const float4*)(x + src_off(g, row, c * 4));
                const uint2 d = *(const uint2*)(dy + (size_t)row * g.C + c * 4);
                const float4 ga = *(const float4*)(gamma + c * 4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gy[i] = make_float4(bf2f((bf16_t)(d.x & 0xffff)) * ga.x, bf2f((bf16_t)(d.x >> 16)) * ga.y,
                                    bf2f((bf16_t)(d.y & 0xffff)) * ga.z, bf2f((bf16_t)(d.y >> 16)) * ga.w);
                s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
                s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
            }
        }
        const float m1 = group_sum<LPR>(s1) * invC;
        const float m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * LPR;
            if (c < nch) {
                const size_t off = src_off(g, row, c * 4);
                float4 o = make_float4(rs * (gy[i].x - m1 - xh[i].x * m2), rs * (gy[i].y - m1 - xh[i].y * m2),
                                       rs * (gy[i].z - m1 - xh[i].z * m2), rs * (gy[i].w - m1 - xh[i].w * m2));
                if (dres) {
                    const float4 r = *(const float4*)(dres + off);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *(float4*)(dx + off) = o;
            }
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
        for (int row = r0 + ty; row < r1; row += RL) {
            const float mu = mean[row], rs = rstd[row];
            const float4 xv = *(const float4*)(x + src_off(g, row, c * 4));
            const uint2 d = *(const uint2*)(dy + (size_t)row * g.C + c * 4);
            const float d0 = bf2f((bf16_t)(d.x & 0xffff)), d1 = bf2f((bf16_t)(d.x >> 16));
            const float d2 = bf2f((bf16_t)(d.y & 0xffff)), d3 = bf2f((bf16_t)(d.y >> 16));
            a[0] += d0 * (xv.x - mu) * rs; a[1] += d1 * (xv.y - mu) * rs;
            a[2] += d2 * (xv.z - mu) * rs; a[3] += d3 * (xv.w - mu) * rs;
            a[4] += d0; a[5] += d1; a[6] += d2; a[7] += d3;
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
            if (i < 4) atomicAdd(dgamma + c * 4 + i, s);
            else atomicAdd(dbeta + c * 4 + (i - 4), s);
        }
    }
}

template <typename F>
int dispatch_ln(int C, F&& f) {
    const int nch = C >> 2;
    if (nch <= 16 * 2) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
    if (nch <= 16 * 4) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
    if (nch <= 64 * 2) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
    if (nch <= 64 * 4) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 4>{});
    if (nch <= 64 * 8) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 8>{});
    if (nch <= 64 * 24) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 24>{});
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
    const int k = t % g.kw, tt = t / g.kw;
    const int i = tt % g.p0, ic = tt / g.p0;
    int col = g.p1 * w + k;
    if (g.circular) { col -= 2; if (col < 0) col += g.Win; if (col >= g.Win) col -= g.Win; }
    return img[(((size_t)b * g.Cin + ic) * g.Hin + (g.p0 * h + i)) * g.Win + col];
}

// one wave per token (grid-stride), lane owns channels lane and lane+64 (E <= 128)
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ out,
                                                              EmbedGeom g, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int ntok = g.B * g.Ho * g.Wo;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < g.E, v1 = c1 < g.E;
    const float invE = 1.0f / (float)g.E;
    for (int tok = wave; tok < ntok; tok += nwaves) {
        const int wq = tok % g.Wo, t2 = tok / g.Wo;
        const int h = t2 % g.Ho, b = t2 / g.Ho;
        float a0 = v0 ? bias[c0] : 0.f, a1 = v1 ? bias[c1] : 0.f;
        for (int t = 0; t < g.taps; ++t) {
            const float xv = embed_tap(img, g, b, h, wq, t);
            if (v0) a0 += w[c0 * g.taps + t] * xv;
            if (v1) a1 += w[c1 * g.taps + t] * xv;
        }
        const float mu = group_sum<64>((v0 ? a0 : 0.f) + (v1 ? a1 : 0.f)) * invE;
        const float d0 = v0 ? a0 - mu : 0.f, d1 = v1 ? a1 - mu : 0.f;
        const float rs = rsqrtf(group_sum<64>(d0 * d0 + d1 * d1) * invE + eps);
        if (v0) out[(size_t)tok * g.E + c0] = d0 * rs * gamma[c0] + beta[c0];
        if (v1) out[(size_t)tok * g.E + c1] = d1 * rs * gamma[c1] + beta[c1];
    }
}

constexpr int EMB_MAXT = 8;
__global__ __launch_bounds__(256) void patch_embed_bwd_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ dout, float* dw, float* db,
                                                              float* dgamma, float* dbeta, EmbedGeom g, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int ntok = g.B * g.Ho * g.Wo;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < g.E, v1 = c1 < g.E;
    const float invE = 1.0f / (float)g.E;
    const bool inreg = g.taps <= EMB_MAXT;
    float aw0[EMB_MAXT], aw1[EMB_MAXT];
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) { aw0[t] = 0.f; aw1[t] = 0.f; }
    float ab0 = 0.f, ab1 = 0.f, ag0 = 0.f, ag1 = 0.f, abe0 = 0.f, abe1 = 0.f;
    const float ga0 = v0 ? gamma[c0] : 0.f, ga1 = v1 ? gamma[c1] : 0.f;
    for (int tok = wave; tok < ntok; tok += nwaves) {
        const int wq = tok % g.Wo, t2 = tok / g.Wo;
        const int h = t2 % g.Ho, b = t2 / g.Ho;
        float a0 = v0 ? bias[c0] : 0.f, a1 = v1 ? bias[c1] : 0.f;
        for (int t = 0; t < g.taps; ++t) {
            const float xv = embed_tap(img, g, b, h, wq, t);
            if (v0) a0 += w[c0 * g.taps + t] * xv;
            if (v1) a1 += w[c1 * g.taps + t] * xv;
        }
        const float mu = group_sum<64>((v0 ? a0 : 0.f) + (v1 ? a1 : 0.f)) * invE;
        const float d0 = v0 ? a0 - mu : 0.f, d1 = v1 ? a1 - mu : 0.f;
        const float rs = rsqrtf(group_sum<64>(d0 * d0 + d1 * d1) * invE + eps);
        const float xh0 = d0 * rs, xh1 = d1 * rs;
        const float dy0 = v0 ? dout[(size_t)tok * g.E + c0] : 0.f, dy1 = v1 ? dout[(size_t)tok * g.E + c1] : 0.f;
        ag0 += dy0 * xh0; ag1 += dy1 * xh1; abe0 += dy0; abe1 += dy1;
        const float gy0 = dy0 * ga0, gy1 = dy1 * ga1;
        const float m1 = group_sum<64>(gy0 + gy1) * invE;
        const float m2 = group_sum<64>(gy0 * xh0 + gy1 * xh1) * invE;
        const float dc0 = v0 ? rs * (gy0 - m1 - xh0 * m2) : 0.f, dc1 = v1 ? rs * (gy1 - m1 - xh1 * m2) : 0.f;
        ab0 += dc0; ab1 += dc1;
        if (inreg) {
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) {
                if (t < g.taps) {
                    const float xv = embed_tap(img, g, b, h, wq, t);
                    aw0[t] += dc0 * xv; aw1[t] += dc1 * xv;
                }
            }
        } else {
            for (int t = 0; t < g.taps; ++t) {
                const float xv = embed_tap(img, g, b, h, wq, t);
                if (v0) atomicAdd(dw + c0 * g.taps + t, dc0 * xv);
                if (v1) atomicAdd(dw + c1 * g.taps + t, dc1 * xv);
            }
        }
    }
    if (v0) {
        atomicAdd(db + c0, ab0); atomicAdd(dgamma + c0, ag0); atomicAdd(dbeta + c0, abe0);
        if (inreg) {
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) if (t < g.taps) atomicAdd(dw + c0 * g.taps + t, aw0[t]);
        }
    }
    if (v1) {
        atomicAdd(db + c1, ab1); atomicAdd(dgamma + c1, ag1); atomicAdd(dbeta + c1, abe1);
        if (inreg) {
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) if (t < g.taps) atomicAdd(dw + c1 * g.taps + t, aw1[t]);
        }
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
    return dispatch_ln(C, [&](auto lpr, auto nch) {
        constexpr int LPR = decltype(lpr)::value, NCH = decltype(nch)::value;
        const int rpb = 256 / LPR;
        const int grid = min((rows + rpb - 1) / rpb, 256 * 16);
        hipLaunchKernelGGL((ln_fwd_kernel<LPR, NCH>), dim3(grid), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd,
                           rows, g, eps);
        TULIP_CHECK_LAUNCH();
        return TULIP_OK;
    });
}

extern "C" int tulip_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd,
                                   const float* gamma, const float* dres, float* dx, int rows, int C, int merge, int B,
                                   int H, int W, hipStream_t stream) {
    if (rows <= 0) return TULIP_OK;
    if (!geom_ok(rows, C, merge, B, H, W)) return TULIP_ERR_ARG;
    RowGeom g{C, merge, B, H, W};
    return dispatch_ln(C, [&](auto lpr, auto nch) {
        constexpr int LPR = decltype(lpr)::value, NCH = decltype(nch)::value;
        const int rpb = 256 / LPR;
        const int grid = min((rows + rpb - 1) / rpb, 256 * 16);
        hipLaunchKernelGGL((ln_bwd_kernel<LPR, NCH>), dim3(grid), dim3(256), 0, stream, dy, x, mean, rstd, gamma, dres,
                           dx, rows, g);
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
    int gy = max(1, min((rows + 63) / 64, 1024 / gx));
    const int rows_per_block = (rows + gy - 1) / gy;
    gy = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(gx, gy), dim3(256), 0, stream, dy, x, mean, rstd, dgamma, dbeta, rows,
                       g, tpr_log2, rows_per_block);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_patch_embed_fwd(const float* img, const float* w, const float* b, const float* gamma,
                                     const float* beta, float* out, int B, int Cin, int Hin, int Win, int E, int p0,
                                     int p1, int kw, int circular, float eps, hipStream_t stream) {
    EmbedGeom g;
    if (!embed_geom(g, B, Cin, Hin, Win, E, p0, p1, kw, circular)) return TULIP_ERR_ARG;
    const int ntok = g.B * g.Ho * g.Wo;
    if (ntok <= 0) return TULIP_OK;
    const int grid = min((ntok + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(patch_embed_fwd_kernel, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, beta, out, g, eps);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_patch_embed_bwd(const float* img, const float* w, const float* b, const float* gamma,
                                     const float* dout, float* dw, float* db, float* dgamma, float* dbeta, int B,
                                     int Cin, int Hin, int Win, int E, int p0, int p1, int kw, int circular, float eps,
                                     hipStream_t stream) {
    EmbedGeom g;
    if (!embed_geom(g, B, Cin, Hin, Win, E, p0, p1, kw, circular)) return TULIP_ERR_ARG;
    const int ntok = g.B * g.Ho * g.Wo;
    if (ntok <= 0) return TULIP_OK;
    const int grid = min((ntok + 3) / 4, 256);
    hipLaunchKernelGGL(patch_embed_bwd_kernel, dim3(grid), dim3(256), 0, stream, img, w, b, gamma, dout, dw, db, dgamma,
                       dbeta, g, eps);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
