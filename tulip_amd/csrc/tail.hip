// Fused output head (tulip.py:724-731): conv1x1 E->16E (+bias) -> LeakyReLU(0.01) -> PixelShuffle(4)
// -> conv1x1 E->1, written straight to pred (B,1,4H,4W).   gfx950 only.
//
// The expand conv is a [tokens x 16E x E] GEMM whose output column oc = c*16 + (i*4+j) is exactly one
// 16-wide MFMA fragment per channel c.  Issued as We.Xn^T, lane l holds for token (l&15) the four
// sub-pixels (i = l>>4, j = 0..3) of channel c, so LeakyReLU, the decoder_pred weight and the sum
// over c are a per-lane FMA chain: the (B,16E,H,W) intermediate (100 MB at B=8) never exists and
// pred rows are stored as 16-byte vectors.  The backward recomputes the pre-activation the same way.
#include "common.h"
#include "tulip_hip.h"

namespace {

struct TailGeom {
    int M, H, W, E;
};

template <int KS>
__device__ __forceinline__ void load_x(const bf16_t* __restrict__ xn, const TailGeom& g, int m0, int li, int gq,
                                       bf16x8 (&xb)[2][KS]) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tok = m0 + mf * 16 + li, k = ks * 32 + gq * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (tok < g.M && k < g.E) v = *(const bf16x8*)(xn + (size_t)tok * g.E + k);
            xb[mf][ks] = v;
        }
}

template <int KS>
__device__ __forceinline__ void expand_channel(const bf16_t* __restrict__ We, const TailGeom& g, int c, int li, int gq,
                                               const bf16x8 (&xb)[2][KS], f32x4 (&acc)[2]) {
    bf16x8 wa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + gq * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < g.E) v = *(const bf16x8*)(We + (size_t)(c * 16 + li) * g.E + k);
        wa[ks] = v;
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], xb[mf][ks], a, 0, 0, 0);
        acc[mf] = a;  // a[r] = Z[token li of frag mf][c*16 + 4*gq + r]
    }
}

__device__ __forceinline__ size_t pred_off(const TailGeom& g, int tok, int i) {
    const int t = fast_div(tok, g.W), w = tok - t * g.W;
    const int b = fast_div(t, g.H), h = t - b * g.H;
    return ((size_t)b * 4 * g.H + 4 * h + i) * (4 * g.W) + 4 * w;
}

// Work split: a workgroup owns 32 tokens; its 4 waves share them and each walks a quarter of the E channels
// (the first version gave each wave 32 tokens and all E channels: 1 wave per SIMD and a 96-deep chain of
// dependent L2 loads -- latency-bound at 85 / 111 us).  Forward: the four partial pixel sums meet in LDS.
template <int KS>
__global__ __launch_bounds__(256, 2) void tail_fwd_kernel(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ We,
                                                       const float* __restrict__ be, const float* __restrict__ wd,
                                                       float* __restrict__ pred, TailGeom g) {
    __shared__ __attribute__((aligned(16))) float red[4][32][16];
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 32;
    bf16x8 xb[2][KS];
    load_x<KS>(xn, g, m0, li, gq, xb);
    float pacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int cper = (g.E + 3) / 4, c0 = wid * cper, c1 = min(g.E, c0 + cper);
    for (int c = c0; c < c1; ++c) {
        f32x4 acc[2];
        expand_channel<KS>(We, g, c, li, gq, xb, acc);
        const float4 b4 = *(const float4*)(be + c * 16 + gq * 4);
        const float wc = wd[c];
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[mf][r] + bb[r];
                pacc[mf][r] += wc * (z > 0.f ? z : 0.01f * z);
            }
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
        *(float4*)&red[wid][mf * 16 + li][gq * 4] = make_float4(pacc[mf][0], pacc[mf][1], pacc[mf][2], pacc[mf][3]);
    __syncthreads();
    if (threadIdx.x < 128) {
        const int t = threadIdx.x >> 2, i = threadIdx.x & 3;   // token, sub-row i (4 pixels j = 0..3)
        const int tok = m0 + t;
        if (tok < g.M) {
            float4 o = *(const float4*)&red[0][t][i * 4];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = *(const float4*)&red[w][t][i * 4];
                o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            }
            *(float4*)(pred + pred_off(g, tok, i)) = o;
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void tail_bwd_kernel(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ We,
                                                       const float* __restrict__ be, const float* __restrict__ wd,
                                                       const float* __restrict__ dpred, bf16_t* __restrict__ dz,
                                                       float* dwd, TailGeom g, const float* __restrict__ target,
                                                       const float* __restrict__ gscale_dev, float gscale) {
    // target != NULL: `dpred` is the forward's pred and the L1 gradient sign(pred-target)*g/N
    // (tulip.py:692-693 backward) is formed here instead of by a separate pass.
    __shared__ float lds_dwd[128];
    __shared__ __attribute__((aligned(16))) bf16_t stile[4][32 * 72];   // 64 cols + 8 pad (bank spread)
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    if (threadIdx.x < 128) lds_dwd[threadIdx.x] = 0.f;
    __syncthreads();
    const int m0 = blockIdx.x * 32;
    bf16x8 xb[2][KS];
    load_x<KS>(xn, g, m0, li, gq, xb);
    float dp[2][4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int tok = m0 + mf * 16 + li;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok < g.M) {
            const size_t po = pred_off(g, tok, gq);
            d = *(const float4*)(dpred + po);
            if (target) {
                const float4 t = *(const float4*)(target + po);
                const float gs = (gscale_dev ? gscale_dev[0] : gscale) / (16.0f * (float)g.M);
                const float e[4] = {d.x - t.x, d.y - t.y, d.z - t.z, d.w - t.w};
                d = make_float4(e[0] > 0.f ? gs : (e[0] < 0.f ? -gs : 0.f), e[1] > 0.f ? gs : (e[1] < 0.f ? -gs : 0.f),
                                e[2] > 0.f ? gs : (e[2] < 0.f ? -gs : 0.f), e[3] > 0.f ? gs : (e[3] < 0.f ? -gs : 0.f));
            }
        }
        dp[mf][0] = d.x; dp[mf][1] = d.y; dp[mf][2] = d.z; dp[mf][3] = d.w;
    }
    const int N = 16 * g.E;
    const int cper = (g.E + 3) / 4, c0 = wid * cper, c1 = min(g.E, c0 + cper);
    // dz leaves through a per-wave LDS tile [32 tokens][4 channels x 16] so that every token row is written as
    // 128 contiguous bytes with 16-B stores (the MFMA layout alone gives 8-B pieces in 32-B segments, which made
    // this 100 MB write the whole cost of the kernel)
    bf16_t* tile = stile[wid];
    for (int cg = c0; cg < c1; cg += 4) {
        const int ng = min(4, c1 - cg);
        for (int cc = 0; cc < ng; ++cc) {
            const int c = cg + cc;
            f32x4 acc[2];
            expand_channel<KS>(We, g, c, li, gq, xb, acc);
            const float4 b4 = *(const float4*)(be + c * 16 + gq * 4);
            const float wc = wd[c];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            float part = 0.f;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[mf][r] + bb[r];
                    const bool pos = z > 0.f;
                    part += dp[mf][r] * (pos ? z : 0.01f * z);
                    o[r] = dp[mf][r] * wc * (pos ? 1.0f : 0.01f);
                }
                *(uint2*)(tile + (mf * 16 + li) * 72 + cc * 16 + gq * 4) =
                    make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            }
            part = group_sum<64>(part);
            if (lane == 0) lds_dwd[c] = part;          // channel c belongs to this wave alone
        }
        // wave-private tile: LDS ops of one wave execute in order, no barrier needed
        const int chunks = ng * 2;                      // 16-B chunks per token row
        for (int id = lane; id < 32 * chunks; id += 64) {
            const int t = fast_div(id, chunks), ch = id - t * chunks;
            const int tok = m0 + t;
            if (tok < g.M)
                *(uint4*)(dz + (size_t)tok * N + cg * 16 + ch * 8) = *(const uint4*)(tile + t * 72 + ch * 8);
        }
    }
    __syncthreads();
    // one plain partial row per workgroup: dwd[blockIdx.x][128] (folded by tulip_reduce_rows2)
    if (threadIdx.x < 128) dwd[(size_t)blockIdx.x * 128 + threadIdx.x] = threadIdx.x < g.E ? lds_dwd[threadIdx.x] : 0.f;
}

}  // namespace

extern "C" int tulip_tail_fwd(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, float* pred,
                              int B, int H, int W, int E, hipStream_t stream) {
    if (E <= 0 || (E & 7) || E > 128) return TULIP_ERR_ARG;
    TailGeom g{B * H * W, H, W, E};
    if (g.M <= 0) return TULIP_OK;
    const dim3 grid((g.M + 31) / 32), block(256);
    const int ks = (E + 31) / 32;
    if (ks <= 2) hipLaunchKernelGGL(tail_fwd_kernel<2>, grid, block, 0, stream, xn, We, be, wd, pred, g);
    else if (ks == 3) hipLaunchKernelGGL(tail_fwd_kernel<3>, grid, block, 0, stream, xn, We, be, wd, pred, g);
    else hipLaunchKernelGGL(tail_fwd_kernel<4>, grid, block, 0, stream, xn, We, be, wd, pred, g);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_tail_bwd(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd,
                              const float* dpred, uint16_t* dz, float* dwd, int B, int H, int W, int E,
                              const float* target, const float* gscale_dev, float gscale, hipStream_t stream) {
    if (E <= 0 || (E & 7) || E > 128) return TULIP_ERR_ARG;
    TailGeom g{B * H * W, H, W, E};
    if (g.M <= 0) return TULIP_OK;
    const dim3 grid((g.M + 31) / 32), block(256);
    const int ks = (E + 31) / 32;
    if (ks <= 2) hipLaunchKernelGGL(tail_bwd_kernel<2>, grid, block, 0, stream, xn, We, be, wd, dpred, dz, dwd, g, target, gscale_dev, gscale);
    else if (ks == 3) hipLaunchKernelGGL(tail_bwd_kernel<3>, grid, block, 0, stream, xn, We, be, wd, dpred, dz, dwd, g, target, gscale_dev, gscale);
    else hipLaunchKernelGGL(tail_bwd_kernel<4>, grid, block, 0, stream, xn, We, be, wd, dpred, dz, dwd, g, target, gscale_dev, gscale);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
