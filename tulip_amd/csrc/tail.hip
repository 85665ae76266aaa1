// Fused output head (tulip.py:724-731): conv1x1 E->16E (+bias) -> LeakyReLU(0.01) -> PixelShuffle(4)
// -> conv1x1 E->1, written straight to pred (B,1,4H,4W).   gfx950 only.
//
// The expand conv is a [tokens x 16E x E] GEMM whose output column oc = c*16 + (i*4+j) is exactly one
// 16-wide MFMA fragment per channel c.  Issued as We.Xn^T, lane l holds for token (l&15) the four
// sub-pixels (i = l>>4, j = 0..3) of channel c, so LeakyReLU, the decoder_pred weight and the sum
// over c are a per-lane FMA chain: the (B,16E,H,W) intermediate (100 MB at B=8) never exists and
// pred rows are stored as 16-byte vectors.  The backward recomputes the pre-activation the same way.
#include <stdlib.h>
#include <type_traits>
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

// FULL: E == 32 KS, every lane's 8 columns exist -- no exec-masked branch (and no zero fill) around each of the KS loads of a channel:
// with them the channel loop was 82 instructions for 6 MFMAs, and four waves per SIMD take turns on its issue slots
template <int KS, bool FULL = false>
__device__ __forceinline__ void expand_channel(const bf16_t* __restrict__ We, const TailGeom& g, int c, int li, int gq,
                                               const bf16x8 (&xb)[2][KS], f32x4 (&acc)[2]) {
    bf16x8 wa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + gq * 8;
        if constexpr (FULL) {
            wa[ks] = *(const bf16x8*)(We + (size_t)(c * 16 + li) * (KS * 32) + k);
        } else {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (k < g.E) v = *(const bf16x8*)(We + (size_t)(c * 16 + li) * g.E + k);
            wa[ks] = v;
        }
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], xb[mf][ks], a, 0, 0, 0);
        acc[mf] = a;  // a[r] = Z[token li of frag mf][c*16 + 4*gq + r]
    }
}

// LeakyReLU(0.01)(z) = max(z, 0.01 z): the same value as the compare / select form, two instructions instead of three
__device__ __forceinline__ float leaky01(float z) { return fmaxf(z, 0.01f * z); }

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
    auto channels = [&](auto F) {
        for (int c = c0; c < c1; ++c) {
            f32x4 acc[2];
            expand_channel<KS, decltype(F)::value>(We, g, c, li, gq, xb, acc);
            const float4 b4 = *(const float4*)(be + c * 16 + gq * 4);
            const float wc = wd[c];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc[mf][r] += wc * leaky01(acc[mf][r] + bb[r]);
        }
    };
    if (g.E == KS * 32) channels(std::true_type{}); else channels(std::false_type{});      // (uniform)
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

// The same forward with norm_up (tulip.py:720) in front and the L1 / pixel loss partial sums (tulip.py:690-700) behind it, one
// launch instead of four on the latency-bound chain: every wave normalises the workgroup's 32 token rows itself (lane
// (li, gq) holds channels 32 ks + 8 gq .. + 7 of token li: exactly its MFMA B fragments, so the LayerNorm output never
// leaves registers on the way to the expand conv); wave 0 also stores it (bf16) with the row statistics for the backward.
struct TailNorm { const float* x; const float* gamma; const float* beta; float eps; bf16_t* xn; float* mean; float* rstd; };
struct TailLoss { const float* target; float* partials; int log_transform; };
template <int KS>
__global__ __launch_bounds__(256, 2) void tail_fwd_ln_kernel(const TailNorm nrm, const bf16_t* __restrict__ We,
                                                          const float* __restrict__ be, const float* __restrict__ wd,
                                                          float* __restrict__ pred, TailGeom g, const TailLoss ls) {
    __shared__ __attribute__((aligned(16))) float red[4][32][16];
    __shared__ float lred[2][2];
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 32;
    bf16x8 xb[2][KS];
    const float invE = 1.0f / (float)g.E;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int tok = m0 + mf * 16 + li;
        const bool valid = tok < g.M;
        float v[KS][8];
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + gq * 8;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (valid && k < g.E) {
                a = *(const float4*)(nrm.x + (size_t)tok * g.E + k);
                b = *(const float4*)(nrm.x + (size_t)tok * g.E + k + 4);
            }
            v[ks][0] = a.x; v[ks][1] = a.y; v[ks][2] = a.z; v[ks][3] = a.w;
            v[ks][4] = b.x; v[ks][5] = b.y; v[ks][6] = b.z; v[ks][7] = b.w;
            sm += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
        }
        sm = rows_sum(sm);
        const float mu = sm * invE;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (ks * 32 + gq * 8 < g.E) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mu; q += d * d; }
            }
        q = rows_sum(q);
        const float rs = rsqrtf(q * invE + nrm.eps);
        if (wid == 0 && gq == 0 && valid) { nrm.mean[tok] = mu; nrm.rstd[tok] = rs; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + gq * 8;
            bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (valid && k < g.E) {
                const float4 g0 = *(const float4*)(nrm.gamma + k), g1 = *(const float4*)(nrm.gamma + k + 4);
                const float4 b0 = *(const float4*)(nrm.beta + k), b1 = *(const float4*)(nrm.beta + k + 4);
                const uint4 pk = make_uint4(
                    pack_bf16x2((v[ks][0] - mu) * rs * g0.x + b0.x, (v[ks][1] - mu) * rs * g0.y + b0.y),
                    pack_bf16x2((v[ks][2] - mu) * rs * g0.z + b0.z, (v[ks][3] - mu) * rs * g0.w + b0.w),
                    pack_bf16x2((v[ks][4] - mu) * rs * g1.x + b1.x, (v[ks][5] - mu) * rs * g1.y + b1.y),
                    pack_bf16x2((v[ks][6] - mu) * rs * g1.z + b1.z, (v[ks][7] - mu) * rs * g1.w + b1.w));
                o = __builtin_bit_cast(bf16x8, pk);
                if (wid == 0) *(uint4*)(nrm.xn + (size_t)tok * g.E + k) = pk;
            }
            xb[mf][ks] = o;
        }
    }
    float pacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int cper = (g.E + 3) / 4, c0 = wid * cper, c1 = min(g.E, c0 + cper);
    auto channels = [&](auto F) {
        for (int c = c0; c < c1; ++c) {
            f32x4 acc[2];
            expand_channel<KS, decltype(F)::value>(We, g, c, li, gq, xb, acc);
            const float4 b4 = *(const float4*)(be + c * 16 + gq * 4);
            const float wc = wd[c];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc[mf][r] += wc * leaky01(acc[mf][r] + bb[r]);
        }
    };
    if (g.E == KS * 32) channels(std::true_type{}); else channels(std::false_type{});      // (uniform)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
        *(float4*)&red[wid][mf * 16 + li][gq * 4] = make_float4(pacc[mf][0], pacc[mf][1], pacc[mf][2], pacc[mf][3]);
    __syncthreads();
    float l0 = 0.f, l1 = 0.f;
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
            const size_t po = pred_off(g, tok, i);
            *(float4*)(pred + po) = o;
            if (ls.partials) {
                const float4 q = *(const float4*)(ls.target + po);
                l0 = (fabsf(o.x - q.x) + fabsf(o.y - q.y)) + (fabsf(o.z - q.z) + fabsf(o.w - q.w));
                if (ls.log_transform)
                    l1 = (fabsf(expm1f(o.x) - expm1f(q.x)) + fabsf(expm1f(o.y) - expm1f(q.y))) +
                         (fabsf(expm1f(o.z) - expm1f(q.z)) + fabsf(expm1f(o.w) - expm1f(q.w)));
            }
        }
    }
    if (ls.partials) {                                          // uniform
        l0 = group_sum<64>(l0); l1 = group_sum<64>(l1);
        if (lane == 0 && wid < 2) { lred[0][wid] = l0; lred[1][wid] = l1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            ls.partials[blockIdx.x * 2] = lred[0][0] + lred[0][1];
            ls.partials[blockIdx.x * 2 + 1] = lred[1][0] + lred[1][1];
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


// ---------------------------------------------------------------------------------------------------------------------
// The backward without the (B,16E,H,W) gradient tensor.  tail_bwd_kernel above writes dz = d(loss)/d(expand
// pre-activation) -- 100 MB at batch 8 -- for two consumers: the data-gradient GEMM dxn = dz . We (read once) and the weight
// gradient dWe = dz^T . xn, dbe = colsum(dz) (read again on the side queue).  Both recompute it instead:
//   * tail_bwd_dgrad_kernel (the chain): token-sliced like the forward; the dz values of two channels, rounded to bf16 in
//     the accumulator layout, are a 32-deep B operand of the next MFMA up to a permutation of the contraction index, and
//     We^T is read with the same permutation from a wave-private LDS tile through ds_read_b64_tr_b16 (csrc/swin96.hip's
//     chained-operand trick); dxn leaves as bf16 [tokens][E], the operand of norm_up's backward;
//   * tail_wgrad_kernel (the side queue): channel-sliced; Z^T is computed by swapping the MFMA operands, so a lane holds
//     4 consecutive TOKENS of one output channel: two 16-token blocks are the 32-deep contraction fragment of dWe^T += xn^T . dz
//     with xn^T read transposed from the step's LDS tile.
// PITCH: rows of a bf16 [rows][E] tile read by transpose reads are padded to 32 B x odd (conflict-free 16 x 32-B blocks).
constexpr int tr_pitch(int E) { return (((E * 2 + 31) / 32) | 1) * 32; }
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_tl;
__device__ __forceinline__ bf16x4 trr_t(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_tl*)p);
}
__device__ __forceinline__ bf16x8 cat8_t(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x4 pack4_t(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    return __builtin_bit_cast(bf16x4, (u32x2_t){pack_bf16x2(a, b), pack_bf16x2(c, d)});
}
// LeakyReLU(0.01)' times an upstream value d without a compare / select pair (the kernels below are vector-ALU bound beside
// their MFMAs): d * (0.505 + 0.495 sign(z)) = d | 0.01 d, the sign taken from z's sign bit (v_and_or_b32) and applied by one
// fma on the pre-scaled pair (a = 0.505 d, b = 0.495 d).  z = +0 counts as positive (torch: 0.01 at exactly 0; measure zero).
__device__ __forceinline__ float leaky_grad_sel(float z, float a, float b) {
    const float sg = __uint_as_float((__float_as_uint(z) & 0x80000000u) | 0x3f800000u);
    return fmaf(sg, b, a);
}

// upstream gradient of pred at 4 consecutive pixels: either given, or the L1 gradient formed from (pred, target)
__device__ __forceinline__ float4 pred_grad4(const float* __restrict__ dpred, const float* __restrict__ target, size_t po,
                                             float gs) {
    float4 d = *(const float4*)(dpred + po);
    if (target) {
        const float4 t = *(const float4*)(target + po);
        const float e[4] = {d.x - t.x, d.y - t.y, d.z - t.z, d.w - t.w};
        d = make_float4(e[0] > 0.f ? gs : (e[0] < 0.f ? -gs : 0.f), e[1] > 0.f ? gs : (e[1] < 0.f ? -gs : 0.f),
                        e[2] > 0.f ? gs : (e[2] < 0.f ? -gs : 0.f), e[3] > 0.f ? gs : (e[3] < 0.f ? -gs : 0.f));
    }
    return d;
}

// norm_up's backward (autograd of tulip.py:720) in the epilogue of tail_bwd_dgrad_kernel: x = the LayerNorm's input rows,
// dx / dx_bf16 / param_partials as tulip_layernorm_bwd writes them (one partial row [dgamma | dbeta] per workgroup)
struct TailNormBwd {
    const float* x; const float* mean; const float* rstd; const float* gamma;
    float* dx; bf16_t* dx_bf16; const float* cast_rowscale; int cast_rows_per_sample; float* param_partials;
};
template <int KS, int NB>
__global__ __launch_bounds__(256, 2) void tail_bwd_dgrad_kernel(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ We,
                                                             const float* __restrict__ be, const float* __restrict__ wd,
                                                             const float* __restrict__ dpred, bf16_t* __restrict__ dxn,
                                                             float* dwd, TailGeom g, const float* __restrict__ target,
                                                             const float* __restrict__ gscale_dev, float gscale,
                                                             const TailNormBwd nb) {
    constexpr int E = NB * 16, PITCH = tr_pitch(E), RP = E + 4;        // RP: fp32 row pitch of the cross-wave reduction
    constexpr int TILE = 32 * PITCH, RED = 32 * RP * 4;
    constexpr int MAIN = 4 * (TILE > RED ? TILE : RED);
    __shared__ float lds_dwd[128];
    // phase 1: four wave-private [32 output channels][E] tiles of We; phase 2 (overlaid): four [32 tokens][E] fp32 partial dxn;
    // behind them the [32 tokens][2E] LayerNorm affine-gradient terms of the fused norm_up backward
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN + 32 * 2 * E * 4];
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    if (threadIdx.x < 128) lds_dwd[threadIdx.x] = 0.f;
    __syncthreads();
    const int m0 = blockIdx.x * 32;
    bf16x8 xb[2][KS];
    load_x<KS>(xn, g, m0, li, gq, xb);
    float dp[2][4], dq[2][4];
    const float gs = (gscale_dev ? gscale_dev[0] : gscale) / (16.0f * (float)g.M);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int tok = m0 + mf * 16 + li;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok < g.M) d = pred_grad4(dpred, target, pred_off(g, tok, gq), gs);
        dp[mf][0] = d.x; dp[mf][1] = d.y; dp[mf][2] = d.z; dp[mf][3] = d.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dq[mf][r] = 0.495f * dp[mf][r]; dp[mf][r] *= 0.505f; }     // leaky_grad_sel's pair
    }
    const int cper = g.E / 4, c0 = wid * cper, c1 = c0 + cper;          // E % 16 == 0: an even number of channels per wave
    unsigned char* tile = smem + wid * TILE;
    f32x4 dxa[NB][2];
#pragma unroll
    for (int n = 0; n < NB; ++n) { dxa[n][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dxa[n][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* trp = tile + (4 * gq + (li >> 2)) * PITCH + ((li & 3) * 4) * 2;
    for (int c = c0; c < c1; c += 2) {
        bf16x4 ob[2][2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            // rows c*16 .. +15 of We: A operand of the recomputed expand conv, and (through the tile) of the data gradient
            bf16x8 wa[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 32 + gq * 8;
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (NB == 2 * KS || k < E) {              // (E == 32 KS at compile time: no branch around the load, see expand_channel)
                    v = *(const bf16x8*)(We + (size_t)((c + cc) * 16 + li) * E + k);
                    *(bf16x8*)(tile + (cc * 16 + li) * PITCH + k * 2) = v;
                }
                wa[ks] = v;
            }
            const float4 b4 = *(const float4*)(be + (c + cc) * 16 + gq * 4);
            const float wc = wd[c + cc];
            float part = 0.f;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                f32x4 a = {b4.x, b4.y, b4.z, b4.w};                       // the MFMAs accumulate on top of the bias
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], xb[mf][ks], a, 0, 0, 0);
                // LeakyReLU(0.01): d(out)/dz = 1 | 0.01, applied to the upstream gradient by leaky_grad_sel
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sel = leaky_grad_sel(a[r], dp[mf][r], dq[mf][r]);
                    part = fmaf(sel, a[r], part);                         // d(decoder_pred.weight[c]) += dpred * leaky(z)
                    o[r] = sel * wc;
                }
                ob[cc][mf] = pack4_t(o[0], o[1], o[2], o[3]);            // dz[token li][(c+cc)*16 + 4gq + r], bf16
            }
            part = group_sum<64>(part);
            if (lane == 0) lds_dwd[c + cc] = part;                        // channel c belongs to this wave alone
        }
        // dxn^T[k][token] += We^T[k][oc] . dz[oc][token] over the pair's 32 output channels (k order: 4gq.., 16+4gq..)
        const bf16x8 dzf[2] = {mfma_operand_fence(cat8_t(ob[0][0], ob[1][0])), mfma_operand_fence(cat8_t(ob[0][1], ob[1][1]))};
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const bf16x8 wt = cat8_t(trr_t(trp + 32 * n), trr_t(trp + 32 * n + 16 * PITCH));
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) dxa[n][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt, dzf[mf], dxa[n][mf], 0, 0, 0);
        }
    }
    __syncthreads();                                                      // every wave is done with its We tile
    float* red = (float*)smem + wid * (32 * RP);
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) *(f32x4*)(red + (mf * 16 + li) * RP + 16 * n + 4 * gq) = dxa[n][mf];
    __syncthreads();
    if (nb.x == nullptr) {                                               // uniform: dxn itself is the output
        for (int id = threadIdx.x; id < 32 * (E / 4); id += 256) {
            const int t = id / (E / 4), c4 = id - t * (E / 4);
            const float* r0 = (const float*)smem + t * RP + c4 * 4;
            float4 o = *(const float4*)r0;
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = *(const float4*)(r0 + w * (32 * RP));
                o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            }
            if (m0 + t < g.M) *(uint2*)(dxn + (size_t)(m0 + t) * E + c4 * 4) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        }
    } else {
        // LayerNorm backward of the 32 rows: 8 lanes per row, lane j owns the 4-channel pieces j, j + 8, ...
        constexpr int NP = (E / 4 + 7) / 8;
        const int t = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int tok = m0 + t;
        const bool valid = tok < g.M;
        const float mu = valid ? nb.mean[tok] : 0.f, rs = valid ? nb.rstd[tok] : 0.f;
        float gy[NP][4], xh[NP][4];
        float s1 = 0.f, s2 = 0.f;
        float* pp = (float*)(smem + MAIN) + t * (2 * E);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c4 = j + 8 * p;
            if (c4 < E / 4) {
                const float* r0 = (const float*)smem + t * RP + c4 * 4;
                float4 o = *(const float4*)r0;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 v = *(const float4*)(r0 + w * (32 * RP));
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                }
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) xv = *(const float4*)(nb.x + (size_t)tok * E + c4 * 4);
                const float4 ga = *(const float4*)(nb.gamma + c4 * 4);
                const float dy[4] = {o.x, o.y, o.z, o.w}, xr[4] = {xv.x, xv.y, xv.z, xv.w}, gv[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[p][e] = (xr[e] - mu) * rs;
                    pp[c4 * 4 + e] = dy[e] * xh[p][e];                    // d(gamma) term
                    pp[E + c4 * 4 + e] = dy[e];                           // d(beta) term
                    gy[p][e] = dy[e] * gv[e];
                    s1 += gy[p][e];
                    s2 += gy[p][e] * xh[p][e];
                }
            }
        }
        s1 = group_sum<8>(s1) * (1.0f / E);
        s2 = group_sum<8>(s2) * (1.0f / E);
        const float sc = (nb.dx_bf16 && nb.cast_rowscale && valid) ? nb.cast_rowscale[fast_div(tok, nb.cast_rows_per_sample)] : 1.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c4 = j + 8 * p;
            if (c4 < E / 4 && valid) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = rs * (gy[p][e] - s1 - xh[p][e] * s2);
                *(float4*)(nb.dx + (size_t)tok * E + c4 * 4) = make_float4(d[0], d[1], d[2], d[3]);
                if (nb.dx_bf16)
                    *(uint2*)(nb.dx_bf16 + (size_t)tok * E + c4 * 4) = make_uint2(pack_bf16x2(d[0] * sc, d[1] * sc), pack_bf16x2(d[2] * sc, d[3] * sc));
            }
        }
        __syncthreads();
        if (nb.param_partials) {
            for (int c = threadIdx.x; c < 2 * E; c += 256) {
                const float* col = (const float*)(smem + MAIN) + c;
                float a = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) a += col[r * (2 * E)];
                nb.param_partials[(size_t)blockIdx.x * (2 * E) + c] = a;
            }
        }
    }
    if (threadIdx.x < 128) dwd[(size_t)blockIdx.x * 128 + threadIdx.x] = threadIdx.x < g.E ? lds_dwd[threadIdx.x] : 0.f;
}

template <int N, class F>
__device__ __forceinline__ void static_for_t(F&& f) {
    if constexpr (N > 0) { static_for_t<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// one workgroup: NWV waves x NC channels (a slice of 16 NWV NC output channels of the expand conv) over `steps` 32-token
// steps.  Every slice re-reads xn and pred / target (320 B per token: more than a 12th of the dz row it replaces), so the
// slices are made as wide as the accumulators allow: 8 waves x 3 channels = 4 slices at E = 96.
template <int KS, int NB, int NC, int NWV>
__global__ __launch_bounds__(64 * NWV, (NWV == 4 && NC <= 2 ? 2 : 1)) void tail_wgrad_kernel(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ We,
                                                      const float* __restrict__ be, const float* __restrict__ wd,
                                                      const float* __restrict__ dpred, const float* __restrict__ target,
                                                      const float* __restrict__ gscale_dev, float gscale,
                                                      float* __restrict__ slab_w, float* __restrict__ slab_b, TailGeom g,
                                                      int nslices, int steps_per_split, int wshift) {
    constexpr int NT = 64 * NWV;
    constexpr int E = NB * 16, PITCH = tr_pitch(E), CPR = E / 8, NCHUNK = 32 * CPR, PER = (NCHUNK + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char xt[2][32 * PITCH];
    __shared__ __attribute__((aligned(16))) float dpt[2][16][36];          // [sub-pixel][token], rows padded to 144 B
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    const int slice = blockIdx.x % nslices, split = blockIdx.x / nslices;
    const int cb = (slice * NWV + wid) * NC;                               // nslices = E / (NWV NC)
    const int total = (g.M + 31) >> 5;
    const int s0 = split * steps_per_split, s1 = min(total, s0 + steps_per_split);
    const float gs = (gscale_dev ? gscale_dev[0] : gscale) / (16.0f * (float)g.M);
    // this wave's rows of We as B operands (column = output channel li of channel cb + cc), its bias and decoder weights
    bf16x8 wb[NC][KS];
    float bev[NC], wdv[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + gq * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (k < E) v = *(const bf16x8*)(We + (size_t)((cb + cc) * 16 + li) * E + k);
            wb[cc][ks] = v;
        }
        bev[cc] = be[(cb + cc) * 16 + li];
        wdv[cc] = wd[cb + cc];
    }
    f32x4 acc[NC][NB];
    float bsum[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        bsum[cc] = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[cc][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // Staging: the xn tile (16-B chunks) and one pred / target row piece per thread, RING steps ahead in registers -- a
    // step's compute (~800 cycles) is shorter than the memory latency, so with one step of distance every step waited for
    // its own loads.  fetch only ISSUES loads, unconditionally (rows clamped; a branch around a load makes the compiler's
    // counted vmcnt waits collapse to vmcnt(0)); the L1 sign is formed in stash, RING - 1 steps later.
    constexpr int RING = NWV == 8 ? 2 : 3;                                   // (256 registers per wave with 8 waves)
    typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
    u32x4_s xr[RING][PER];
    float4 dr[RING], tr[RING];
    const float* tsrc = target ? target : dpred;
    const int pt = (threadIdx.x & 127) >> 2, pi = threadIdx.x & 3;          // (token, sub-row) of this thread's pred piece
    // Address arithmetic is the vector-ALU budget of this kernel (a 64-bit multiply-add chain per load and step was more
    // work than the MFMAs): per-thread byte offsets inside a 32-token step are computed ONCE, a step adds them to a scalar
    // base.  Only the ragged last step of the tensor (M % 32 != 0) clamps rows; the pred / target offset is a scalar
    // plus a constant when W is a power of two >= 32 (wshift >= 0: a step's tokens then share one (b, h) row).
    unsigned xoff[PER], xcol[PER];
    int xrow[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = min((int)threadIdx.x + i * NT, NCHUNK - 1);
        xrow[i] = c / CPR;
        xcol[i] = (unsigned)(c - xrow[i] * CPR) * 16u;
        xoff[i] = (unsigned)xrow[i] * (E * 2) + xcol[i];
    }
    const unsigned ppix = (unsigned)(4 * g.W * pi + 4 * pt);
    auto fetch = [&](auto R, int step) {
        constexpr int r = decltype(R)::value;
        const int m0 = min(step, s1 - 1) * 32;
        const bool full = m0 + 32 <= g.M;                                      // uniform
        const unsigned char* xb = (const unsigned char*)(xn + (size_t)m0 * E);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            unsigned off = xoff[i];
            if (!full) off = (unsigned)min(xrow[i], g.M - 1 - m0) * (E * 2) + xcol[i];
            xr[r][i] = *(const u32x4_s*)(xb + off);
        }
        unsigned po;
        if (wshift >= 0 && full) po = (unsigned)(16 * (m0 >> wshift) * g.W + 4 * (m0 & (g.W - 1))) + ppix;
        else po = (unsigned)pred_off(g, min(m0 + pt, g.M - 1), pi);
        dr[r] = *(const float4*)(dpred + po);
        tr[r] = *(const float4*)(tsrc + po);
    };
    auto stash = [&](auto R, int buf, int step) {
        constexpr int r = decltype(R)::value;
        const int m0 = step * 32;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = threadIdx.x + i * NT;
            const int t = c / CPR, k8 = c - t * CPR;
            if (NCHUNK % NT == 0 || c < NCHUNK) {
                u32x4_s v = xr[r][i];
                if (m0 + 32 > g.M && m0 + t >= g.M) v = (u32x4_s){0u, 0u, 0u, 0u};   // rows past the last token contribute nothing
                *(u32x4_s*)(xt[buf] + t * PITCH + k8 * 16) = v;
            }
        }
        if (threadIdx.x < 128) {
            float4 d = dr[r];
            if (target) {
                const float4 q = tr[r];
                const float e[4] = {d.x - q.x, d.y - q.y, d.z - q.z, d.w - q.w};
                d = make_float4(e[0] > 0.f ? gs : (e[0] < 0.f ? -gs : 0.f), e[1] > 0.f ? gs : (e[1] < 0.f ? -gs : 0.f),
                                e[2] > 0.f ? gs : (e[2] < 0.f ? -gs : 0.f), e[3] > 0.f ? gs : (e[3] < 0.f ? -gs : 0.f));
            }
            if (m0 + pt >= g.M) d = make_float4(0.f, 0.f, 0.f, 0.f);
            dpt[buf][pi * 4][pt] = d.x; dpt[buf][pi * 4 + 1][pt] = d.y; dpt[buf][pi * 4 + 2][pt] = d.z; dpt[buf][pi * 4 + 3][pt] = d.w;
        }
    };
    auto compute = [&](int buf) {
            const unsigned char* x = xt[buf];
            bf16x8 xa[2][KS], xT[NB];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int k = ks * 32 + gq * 8;
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (k < E) v = *(const bf16x8*)(x + (mf * 16 + li) * PITCH + k * 2);
                    xa[mf][ks] = v;                                              // A operand: row = token li of block mf
                }
            const unsigned char* trp = x + (4 * gq + (li >> 2)) * PITCH + ((li & 3) * 4) * 2;
#pragma unroll
            for (int n = 0; n < NB; ++n) xT[n] = cat8_t(trr_t(trp + 32 * n), trr_t(trp + 32 * n + 16 * PITCH));   // row = channel 16n + li
            // the upstream gradient of this lane's sub-pixel li at its 2 x 4 tokens, as the pair leaky_grad_sel takes
            float dpl[2][4], dql[2][4];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const float4 d = *(const float4*)&dpt[buf][li][mf * 16 + 4 * gq];
                dpl[mf][0] = d.x; dpl[mf][1] = d.y; dpl[mf][2] = d.z; dpl[mf][3] = d.w;
#pragma unroll
                for (int r = 0; r < 4; ++r) { dql[mf][r] = 0.495f * dpl[mf][r]; dpl[mf][r] *= 0.505f; }   // leaky_grad_sel's pair
            }
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                bf16x4 ob[2];
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    f32x4 z = {bev[cc], bev[cc], bev[cc], bev[cc]};              // z[r] = Z[token mf*16 + 4gq + r][channel li], on top of the bias
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[mf][ks], wb[cc][ks], z, 0, 0, 0);
                    // dz / decoder_pred.weight[c]: the per-channel factor multiplies the accumulators once, at the end (the
                    // kernel is vector-ALU bound: 3 instructions per element here)
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o[r] = leaky_grad_sel(z[r], dpl[mf][r], dql[mf][r]); bsum[cc] += o[r]; }
                    ob[mf] = pack4_t(o[0], o[1], o[2], o[3]);
                }
                const bf16x8 dzf = mfma_operand_fence(cat8_t(ob[0], ob[1]));    // contraction = tokens 4gq.., 16+4gq..
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[cc][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xT[n], dzf, acc[cc][n], 0, 0, 0);
            }
    };
    if (s0 >= s1) return;                                                  // (uniform; the launcher creates no empty split)
    static_for_t<RING>([&](auto R) { fetch(R, s0 + decltype(R)::value); });
    stash(std::integral_constant<int, 0>{}, 0, s0);
    __syncthreads();
    // step s (ring slot j = (s - s0) % RING): compute on buffer (s - s0) & 1; k-step s + 1 goes from slot j + 1 into the other
    // buffer; slot j is refilled with k-step s + RING
    auto step = [&](auto J, int s) {
        constexpr int j = decltype(J)::value;
        const int buf = (s - s0) & 1;
        compute(buf);
        fetch(J, s + RING);
        if (s + 1 < s1) stash(std::integral_constant<int, (j + 1) % RING>{}, buf ^ 1, s + 1);
        __syncthreads();
    };
    for (int s = s0; s < s1; s += RING)
        static_for_t<RING>([&](auto J) { if (s + decltype(J)::value < s1) step(J, s + decltype(J)::value); });
    // acc[cc][n][r] = dWe[(cb+cc)*16 + li][16n + 4gq + r]: one plain slab per token split, folded by tulip_reduce_rows_multi
    float* ow = slab_w + (size_t)split * (16 * E) * E;
    float* ob_ = slab_b + (size_t)split * (16 * E);
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const int oc = (cb + cc) * 16 + li;
        const float wc = wdv[cc];
#pragma unroll
        for (int n = 0; n < NB; ++n)
            *(float4*)(ow + (size_t)oc * E + 16 * n + 4 * gq) =
                make_float4(wc * acc[cc][n][0], wc * acc[cc][n][1], wc * acc[cc][n][2], wc * acc[cc][n][3]);
        float b = wc * bsum[cc];
        b = rows_sum(b);
        if (gq == 0) ob_[oc] = b;
    }
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

extern "C" int tulip_tail_fused_bwd_supported(int E) { return E > 0 && E % 16 == 0 && E <= 128; }

static int tail_bwd_dgrad_impl(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                               uint16_t* dxn, float* dwd, int B, int H, int W, int E, const float* target,
                               const float* gscale_dev, float gscale, const TailNormBwd& nb, hipStream_t stream) {
    if (!tulip_tail_fused_bwd_supported(E) || !xn || !We || !be || !wd || !dpred || !dwd) return TULIP_ERR_ARG;
    TailGeom g{B * H * W, H, W, E};
    if (g.M <= 0) return TULIP_OK;
    const dim3 grid((g.M + 31) / 32), block(256);
#define TULIP_TBD(KS, NB) \
    hipLaunchKernelGGL((tail_bwd_dgrad_kernel<KS, NB>), grid, block, 0, stream, (const bf16_t*)xn, (const bf16_t*)We, be, wd, dpred, \
                       (bf16_t*)dxn, dwd, g, target, gscale_dev, gscale, nb)
    switch (E / 16) {
        case 1: TULIP_TBD(1, 1); break;
        case 2: TULIP_TBD(1, 2); break;
        case 3: TULIP_TBD(2, 3); break;
        case 4: TULIP_TBD(2, 4); break;
        case 5: TULIP_TBD(3, 5); break;
        case 6: TULIP_TBD(3, 6); break;
        case 7: TULIP_TBD(4, 7); break;
        default: TULIP_TBD(4, 8); break;
    }
#undef TULIP_TBD
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_tail_bwd_dgrad(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd,
                                    const float* dpred, uint16_t* dxn, float* dwd, int B, int H, int W, int E,
                                    const float* target, const float* gscale_dev, float gscale, hipStream_t stream) {
    if (!dxn) return TULIP_ERR_ARG;
    const TailNormBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr};
    return tail_bwd_dgrad_impl(xn, We, be, wd, dpred, dxn, dwd, B, H, W, E, target, gscale_dev, gscale, none, stream);
}

extern "C" int tulip_tail_bwd_dgrad_ln(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd,
                                       const float* dpred, float* dwd, int B, int H, int W, int E, const float* target,
                                       const float* gscale_dev, float gscale, const float* x, const float* mean,
                                       const float* rstd, const float* gamma, float* dx, uint16_t* dx_bf16,
                                       const float* cast_rowscale, int cast_rows_per_sample, float* ln_partials,
                                       hipStream_t stream) {
    if (!x || !mean || !rstd || !gamma || !dx) return TULIP_ERR_ARG;
    const TailNormBwd nb{x, mean, rstd, gamma, dx, (bf16_t*)dx_bf16, cast_rowscale,
                         cast_rows_per_sample > 0 ? cast_rows_per_sample : 1, ln_partials};
    return tail_bwd_dgrad_impl(xn, We, be, wd, dpred, nullptr, dwd, B, H, W, E, target, gscale_dev, gscale, nb, stream);
}

extern "C" int tulip_tail_fwd_ln(const float* x, const float* gamma, const float* beta, float eps, uint16_t* xn, float* mean,
                                 float* rstd, const uint16_t* We, const float* be, const float* wd, float* pred,
                                 const float* target, float* loss_partials, int log_transform, int B, int H, int W, int E,
                                 hipStream_t stream) {
    if (E <= 0 || (E & 7) || E > 128 || !x || !gamma || !beta || !xn || !mean || !rstd || !We || !be || !wd || !pred ||
        (loss_partials && !target))
        return TULIP_ERR_ARG;
    TailGeom g{B * H * W, H, W, E};
    if (g.M <= 0) return TULIP_OK;
    const dim3 grid((g.M + 31) / 32), block(256);
    const TailNorm nrm{x, gamma, beta, eps, (bf16_t*)xn, mean, rstd};
    const TailLoss ls{target, loss_partials, log_transform};
    const int ks = (E + 31) / 32;
    if (ks <= 2) hipLaunchKernelGGL(tail_fwd_ln_kernel<2>, grid, block, 0, stream, nrm, (const bf16_t*)We, be, wd, pred, g, ls);
    else if (ks == 3) hipLaunchKernelGGL(tail_fwd_ln_kernel<3>, grid, block, 0, stream, nrm, (const bf16_t*)We, be, wd, pred, g, ls);
    else hipLaunchKernelGGL(tail_fwd_ln_kernel<4>, grid, block, 0, stream, nrm, (const bf16_t*)We, be, wd, pred, g, ls);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

// channels per workgroup of tulip_tail_wgrad: 8 waves x 3 (E % 24 == 0; 256 registers per wave) or 4 waves x 4; about one
// workgroup per CU
static void tail_wgrad_plan(int M, int E, int* nslices, int* splits, int* steps_per_split) {
    const int total = (M + 31) / 32;
    *nslices = E / ((E % 24 == 0) ? 24 : 16);
    int sp = 256 / *nslices;
    if (sp < 1) sp = 1;
    if (sp > total) sp = total;
    *steps_per_split = (total + sp - 1) / sp;
    *splits = (total + *steps_per_split - 1) / *steps_per_split;
}

extern "C" int tulip_tail_wgrad_splits(int B, int H, int W, int E) {
    if (!tulip_tail_fused_bwd_supported(E) || B * H * W <= 0) return 0;
    int ns, sp, st;
    tail_wgrad_plan(B * H * W, E, &ns, &sp, &st);
    return sp;
}

extern "C" int tulip_tail_wgrad(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                                float* slabs_w, float* slabs_b, int B, int H, int W, int E, const float* target,
                                const float* gscale_dev, float gscale, hipStream_t stream) {
    if (!tulip_tail_fused_bwd_supported(E) || !xn || !We || !be || !wd || !dpred || !slabs_w || !slabs_b) return TULIP_ERR_ARG;
    TailGeom g{B * H * W, H, W, E};
    if (g.M <= 0) return TULIP_OK;
    int ns, sp, st;
    tail_wgrad_plan(g.M, E, &ns, &sp, &st);
    const bool wide = E % 24 == 0;
    const dim3 grid(ns * sp), block(wide ? 512 : 256);
    const int wsh = (W >= 32 && (W & (W - 1)) == 0) ? 31 - __builtin_clz((unsigned)W) : -1;
#define TULIP_TWG(KS, NB) \
    do { if (wide) hipLaunchKernelGGL((tail_wgrad_kernel<KS, NB, 3, 8>), grid, block, 0, stream, (const bf16_t*)xn, (const bf16_t*)We, be, wd, dpred, \
                       target, gscale_dev, gscale, slabs_w, slabs_b, g, ns, st, wsh); \
         else hipLaunchKernelGGL((tail_wgrad_kernel<KS, NB, 4, 4>), grid, block, 0, stream, (const bf16_t*)xn, (const bf16_t*)We, be, wd, dpred, \
                       target, gscale_dev, gscale, slabs_w, slabs_b, g, ns, st, wsh); } while (0)
    switch (E / 16) {
        case 1: TULIP_TWG(1, 1); break;
        case 2: TULIP_TWG(1, 2); break;
        case 3: TULIP_TWG(2, 3); break;
        case 4: TULIP_TWG(2, 4); break;
        case 5: TULIP_TWG(3, 5); break;
        case 6: TULIP_TWG(3, 6); break;
        case 7: TULIP_TWG(4, 7); break;
        default: TULIP_TWG(4, 8); break;
    }
#undef TULIP_TWG
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
