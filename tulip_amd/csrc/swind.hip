// The Swin block of the DEEP stages -- C = 768 (stage 3 of every TULIP model) and C = 1536 (stage 4 of tulip_large); heads of 32,
// 16-token windows (2x8, or the 1x16 backup window of a one-row grid), MLP C -> 4C -> C; tulip.py:338-352 with :282-324 and
// :194-200 inside -- as a chain of SLICED launches: four per block forward, four (+ the two LayerNorm backward launches of
// csrc/norm.hip) backward, none of which holds a partial sum.
//
// At these widths a block's weights are 14 / 57 MB of bf16 and a batch of 8 KITTI images has 32 windows (512 tokens): one workgroup
// per window (csrc/swinw.hip) would stream 14 MB through each of 32 CUs, and the GEMM chain it replaces spent ~10 us per launch
// whatever the size (launch + first-stage latency of 64 x 96 tiles that cannot fill the chip).  Here EIGHT workgroups share a group
// of G windows, workgroup (group, slice) runs on XCD `slice` (blockIdx % 8) -- every XCD's L2 holds one eighth of each weight
// matrix, every CU streams 0.15-0.6 MB (1.2-2.4 MB at C = 1536) of fragment-major weights straight into MFMA operands -- and the
// slicing ALTERNATES so that no launch needs another workgroup's result:
//   F1  norm1 (redundant per slice) -> qkv of this slice's C/256 HEADS (two waves per head, each half of the contraction) ->
//       attention -> the heads' 16 x (C/8) attention output;
//   F2  proj by OUTPUT channels (C/8 per slice) over the whole attention output + bias, DropPath, residual -> x1;
//   F3  norm2 (redundant) -> fc1 + GELU of this slice's C/2 HIDDEN channels;
//   F4  fc2 by OUTPUT channels over all 4C hidden channels + bias, DropPath, residual -> block output.
// Backward: B1 fc2' + GELU' by hidden channels; B2 fc1' by output channels -> d(norm2 output) in fp32 -> tulip_layernorm_bwd_splitk
// (one "slab"): norm2', residual, bf16 operand; B3 proj' + attention' by heads; B4 qkv' by output channels -> norm1' likewise.
// The first form of this file kept proj / fc2 (fc1' / qkv') as k-slices inside the by-heads / by-hidden launches and let the eight
// fp32 partial tiles of a group meet in memory (write-through stores, one arrival ticket per group, the last arriver adds them and
// runs the LayerNorm): two launches per direction, nobody waits -- and 19 of a launch's 31 us were that tail (in-kernel stamps,
// profiles/r5_deep_ticket_form_phases.txt: publish + ticket 4.2 us, gathering 7 x 49 KB by one workgroup 7.0 us, the finish
// 7.6 us), 83 + 87 us per block against 63 + 81 for the sequences.  A launch boundary (~2 us) is the cheaper all-to-all.
// Everything the un-fused sequence saves / hands to the weight-gradient launches is written exactly as that sequence writes it
// (the fc1_pre buffer carries gelu'(h), TULIP_BLOCK_FC1_GRAD, as in the split form of csrc/swinw.hip).
#ifndef TULIP_STORE_LATE_D
#define TULIP_STORE_LATE_D 1
#endif
#define TULIP_STORE_LATE TULIP_STORE_LATE_D
#include "common.h"
#include "tulip_hip.h"
#include "swin_stream.h"

namespace {

constexpr int NS = 8;       // slices = workgroups per window group = XCDs

template <int C, int G>
struct GeoD {
    static constexpr int NH = C / 32, NWV = C / 128, NT = NWV * 64, T = 16 * G, KS = C / 32, HPW = NH / NS, HID = 4 * C;
    static constexpr int HS = HID / NS, KH = KS / 2, KSH = HS / 32;
    static constexpr int LPT = C / 48;                 // lanes per token in the row prologues: 48 channels per lane
    static_assert(NWV == 2 * HPW && HS == 64 * NWV && C / 16 == 8 * NWV && (LPT == 16 || LPT == 32), "wave layout");
    static constexpr int XN_BYTES = T * C * 2;
    // by-heads launches (F1 / B3): LayerNorm output or bf16 gradient rows | k-half exchange | per-head tiles | sink
    static constexpr int A_QX = XN_BYTES, A_QX_BYTES = HPW * 6 * G * 1024;
    static constexpr int A_HT = A_QX + A_QX_BYTES, A_HT_BYTES = HPW * 3072;            // V tile (F1) / Q | K | dO tiles (B3)
    static constexpr int A_AFF = A_HT + A_HT_BYTES;                                    // gamma | beta of norm1 (fp32)
    static constexpr int A_WARM = A_AFF + 8 * C;
    static constexpr int A_SMEM = A_WARM + 1024;
    // by-hidden launches (F3 / B1): LayerNorm output or bf16 gradient rows | sink
    static constexpr int B_AFF = XN_BYTES;
    static constexpr int B_WARM = B_AFF + 8 * C;
    static constexpr int B_SMEM = B_WARM + 1024;
    // by-output-channel launches (F2 / F4 / B2 / B4): two buffers of KS k tiles of the operand rows | sink
    static constexpr int N_WARM = 2 * XN_BYTES;
    static constexpr int N_SMEM = N_WARM + 1024;
    static_assert(A_SMEM <= 163840 && N_SMEM <= 163840, "LDS");
};

// natural-order token of tile slot tt = 16 g + t of a group of G neighbouring windows (cyclic shift + window partition are address
// arithmetic, tulip.py:289-297); window wh x ww = 2 x 8 or 1 x 16 (lw = log2 ww)
struct TokMapD {
    int b, wy, wx0, H, W, wh, ww, lw, sh, sw;
    __device__ __forceinline__ void coords(int tt, int& hs, int& ws) const {
        const int t = tt & 15;
        hs = wy * wh + (t >> lw);
        ws = ((wx0 + (tt >> 4)) << lw) + (t & (ww - 1));
    }
    __device__ __forceinline__ size_t row(int tt) const {
        int hs, ws;
        coords(tt, hs, ws);
        int hh = hs + sh; if (hh >= H) hh -= H;
        int w2 = ws + sw; if (w2 >= W) w2 -= W;
        return ((size_t)b * H + hh) * W + w2;
    }
    __device__ __forceinline__ int label(int tt) const {
        int hs, ws;
        coords(tt, hs, ws);
        return 3 * region(hs, H, wh, sh) + region(ws, W, ww, sw);
    }
};
template <int G>
__device__ __forceinline__ TokMapD make_map_d(int H, int W, int wh, int ww, int sh, int sw, int grp) {
    const int nWx = W / ww, nWy = H / wh, gpr = nWx / G;
    TokMapD m;
    m.b = grp / (nWy * gpr);
    grp -= m.b * nWy * gpr;
    m.wy = grp / gpr;
    m.wx0 = (grp - m.wy * gpr) * G;
    m.H = H; m.W = W; m.wh = wh; m.ww = ww; m.lw = ww == 16 ? 4 : 3; m.sh = sh; m.sw = sw;
    return m;
}

// acc[i][g] += W[tile i][k] . X[token tile g][k]^T over KSTEPS 32-deep steps of a fragment-major weight stream (swin_stream.h);
// step p reads the KiB block wt[i] + 512 * ((p / SEG) * SEGSTRIDE + p % SEG) -- SEG-long runs of consecutive k tiles (the
// k-slices of a matrix with several sections: qkv^T) -- and the LDS k tile p of `act`
template <int NTILE, int KSTEPS, int PF, int SEG = KSTEPS, int SEGSTRIDE = KSTEPS>
struct WS {
    bf16x8 ring[PF][NTILE];
    const bf16_t* wt[NTILE];
    static constexpr int koff(int p) { return 512 * ((p / SEG) * SEGSTRIDE + (p % SEG)); }
    __device__ __forceinline__ void start() {
        static_for<PF>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            if constexpr (p < KSTEPS) {
#pragma unroll
                for (int i = 0; i < NTILE; ++i) ring[p][i] = *(const bf16x8*)(wt[i] + koff(p));
            }
        });
    }
    template <int G, int T>
    __device__ __forceinline__ void run(f32x4 (&acc)[NTILE][G], const unsigned char* act, int t, int gq) {
        static_for<KSTEPS>([&](auto K_) {
            constexpr int ks = decltype(K_)::value;
            bf16x8 a[NTILE];
#pragma unroll
            for (int i = 0; i < NTILE; ++i) a[i] = ring[ks % PF][i];
            if constexpr (ks + PF < KSTEPS) {
#pragma unroll
                for (int i = 0; i < NTILE; ++i) ring[ks % PF][i] = *(const bf16x8*)(wt[i] + koff(ks + PF));
            }
            bf16x8 b[G];
#pragma unroll
            for (int g = 0; g < G; ++g) b[g] = frag<T>(act, ks, 16 * g + t, gq);
#pragma unroll
            for (int i = 0; i < NTILE; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[g], acc[i][g], 0, 0, 0);
        });
    }
};

// Cold caches (csrc/swinw.hip, WeightWarm): the 32 workgroups of an XCD walk the same weight slice in the same order, so before
// their streams start they split it between them -- every wave touches a few KiB chunks nobody else touches (LDS-destination
// loads into a sink nobody reads), the slice lands in the XCD's L2 within a few miss latencies.  A matrix's slice is `nruns` runs
// of `run` consecutive KiB blocks, `stride` KiB blocks apart, starting at `w`.
struct SliceWarm {
    unsigned char* sink;
    int nslot, slot, ws, loff;
    bool on;
    __device__ __forceinline__ void init(int grp, int ngrp, int wid, int lane, unsigned char* lds_sink) {
        nslot = ngrp < 32 ? ngrp : 32; slot = grp; on = grp < 32;
        ws = __builtin_amdgcn_readfirstlane(wid); loff = lane * 16; sink = lds_sink;
    }
    template <int NWV>
    __device__ __forceinline__ void touch(const bf16_t* w, int nruns, int run, int stride) {
        if (!on) return;
        const int total = nruns * run;
        for (int c = slot * NWV + ws; c < total; c += nslot * NWV) {
            const int r = c / run, o = c - r * run;
            warm_touch16((const unsigned char*)(w + ((size_t)r * stride + o) * 512) + loff, sink);
        }
    }
};


struct DeepArgs {
    const float* xin; float* x1; float* xout;
    bf16_t *xn1, *qkv, *o, *xn2, *h, *g;
    float *mean1, *rstd1, *mean2, *rstd2;
    const bf16_t *wqkv, *wproj, *w1, *w2;           // fragment-major copies (backward: of the transposes)
    const float *bqkv, *bproj, *b1, *b2, *g1, *be1, *g2, *be2;
    const float* bias_table; const int* rel_index;
    const float *ds0, *ds1;
    bf16_t* out_bf16;
    // backward
    const float* dx;                                // d(block output) (B1)
    bf16_t *dyb_m, *dh, *dyb_a, *dqkv;
    float* dxn;                                     // fp32 d(LayerNorm output) [M][C]: B2 / B4 -> tulip_layernorm_bwd_splitk
    float* biaspart;
    unsigned long long* prof;
    int B, H, W, wh, ww, sh, sw, masked, save, ngrp;
    float eps, scale;
};
#if TULIP_DEV_VARIANTS
#define DEEP_STAMP(k) do { if (a.prof && lane == 0) a.prof[((size_t)blockIdx.x * NWV + wid) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEEP_STAMP(k) ((void)0)
#endif

// The group's T rows of an fp32 [M][C] tensor, LPT lanes per token (48 channels per lane): loaded FIRST in a launch (vmcnt retires
// in order: behind the cold weight loads these rows -- the critical path of the prologue -- would wait for every one of them), then
// either LayerNorm'ed (norm1 / norm2, tulip.py:340,347) or scaled (the DropPath factor of an incoming gradient) into the bf16
// fragment layout in LDS.
template <int C, int T, int NWV>
struct RowPro {
    static constexpr int LPT = C / 48, TPW = 64 / LPT, NJ = 12, NP = (T + NWV * TPW - 1) / (NWV * TPW);
    float4 xv[NP][NJ];
    size_t row[NP];
    int tl;
    __device__ __forceinline__ int tok(int p, int wid, int lane) const { return wid * TPW + lane / LPT + p * NWV * TPW; }
    __device__ __forceinline__ void load(const float* __restrict__ x, const TokMapD& tm, int wid, int lane) {
        tl = lane % LPT;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = tok(p, wid, lane);
            row[p] = tm.row(tt < T ? tt : T - 1);
            if (tt < T) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) xv[p][j] = *(const float4*)(x + row[p] * C + 4 * tl + 4 * LPT * j);
            }
        }
    }
    // LayerNorm -> LDS (and, sv, the saved bf16 rows + statistics)
    // aff: gamma[C] | beta[C] in LDS (staged by the whole workgroup with the rows' loads in flight: fetched behind the statistics
    // they were a second exposed miss latency for the four waves that normalise)
    __device__ __forceinline__ void layernorm(const float* aff, float eps, unsigned char* XN, bf16_t* save, float* mean, float* rstd,
                                              bool sv, int wid, int lane) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = tok(p, wid, lane);
            if (tt < T) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) sm += (xv[p][j].x + xv[p][j].y) + (xv[p][j].z + xv[p][j].w);
                const float mu = group_sum<LPT>(sm) * (1.0f / C);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float d0 = xv[p][j].x - mu, d1 = xv[p][j].y - mu, d2 = xv[p][j].z - mu, d3 = xv[p][j].w - mu;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                const float rs = rsqrtf(group_sum<LPT>(q) * (1.0f / C) + eps);
                if (sv && tl == 0) { mean[row[p]] = mu; rstd[row[p]] = rs; }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int c = 4 * tl + 4 * LPT * j;
                    const float4 ga = *(const float4*)(aff + c), be = *(const float4*)(aff + C + c);
                    const bf16x4 pk = pack4((xv[p][j].x - mu) * rs * ga.x + be.x, (xv[p][j].y - mu) * rs * ga.y + be.y,
                                            (xv[p][j].z - mu) * rs * ga.z + be.z, (xv[p][j].w - mu) * rs * ga.w + be.w);
                    if (sv) store_late((bf16x4*)(save + row[p] * C + c), pk);
                    put4<T>(XN, tt, c, pk);
                }
            }
        }
    }
    __device__ __forceinline__ void scaled(float s, unsigned char* XN, bf16_t* save, bool sv, int wid, int lane) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = tok(p, wid, lane);
            if (tt < T) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int c = 4 * tl + 4 * LPT * j;
                    const bf16x4 pk = pack4(xv[p][j].x * s, xv[p][j].y * s, xv[p][j].z * s, xv[p][j].w * s);
                    if (sv) store_late((bf16x4*)(save + row[p] * C + c), pk);
                    put4<T>(XN, tt, c, pk);
                }
            }
        }
    }
};

// 16-byte chunks of KS k tiles (columns [col0, col0 + C)) of bf16 rows with pitch ld -> registers -> LDS fragment layout
template <int C, int T, int NT>
struct RowStage {
    static constexpr int CH = C / 8, N = T * CH, NV = N / NT;
    static_assert(N % NT == 0, "chunks per thread");
    typedef unsigned u32x4_r __attribute__((ext_vector_type(4)));
    u32x4_r v[NV];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, int ld, int col0, const TokMapD& tm, int tid) {
        static_for<NV>([&](auto K_) {               // (compile-time indices: a run-time-indexed v[] is demoted to LDS / scratch)
            constexpr int k = decltype(K_)::value;
            const int q = tid + k * NT, tok = q / CH, c8 = (q - tok * CH) * 8;
            v[k] = *(const u32x4_r*)(src + tm.row(tok) * ld + col0 + c8);
        });
    }
    __device__ __forceinline__ void store(unsigned char* dst, int tid) const {
        static_for<NV>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            const int q = tid + k * NT, tok = q / CH, c8 = (q - tok * CH) * 8;
            *(u32x4_r*)(dst + (c8 >> 5) * (T * 64) + tok * 64 + ((((c8 >> 3) & 3) ^ swz4(tok)) << 4)) = v[k];
        });
    }
};

// =====================================================================================================================
// F1: norm1 -> qkv of this slice's heads -> attention -> the heads' attention output (tulip.py:340, :298-317)
template <int C, int G>
__global__ __launch_bounds__((GeoD<C, G>::NT)) void deep_attn_fwd_kernel(const DeepArgs a) {
    using Z = GeoD<C, G>;
    constexpr int T = Z::T, KS = Z::KS, KH = Z::KH, NWV = Z::NWV, HPW = Z::HPW, NH = Z::NH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::A_SMEM];
    unsigned char* const XN = smem;
    f32x4* const QX = (f32x4*)(smem + Z::A_QX);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int slice = blockIdx.x & (NS - 1), grp = blockIdx.x >> 3;
    const int hl = wid % HPW, kh = wid / HPW, head = slice * HPW + hl;
    unsigned char* const ldsV = smem + Z::A_HT + hl * 3072;
    const bool sv = a.save && slice == 0;           // the tensors every workgroup of a group holds: saved by the first
    const TokMapD tm = make_map_d<G>(a.H, a.W, a.wh, a.ww, a.sh, a.sw, grp);

    DEEP_STAMP(0);
    RowPro<C, T, NWV> pro;
    pro.load(a.xin, tm, wid, lane);
    const float4 affv = tid < C / 4 ? *(const float4*)(a.g1 + 4 * tid) : *(const float4*)(a.be1 + 4 * tid - C);    // NT = C/2 threads
    // the qkv weight stream of this wave's head (q, k, v: two 16-row tiles each) over its half of the contraction
    WS<6, KH, 2> wq;
#pragma unroll
    for (int i = 0; i < 6; ++i) wq.wt[i] = wtile_ptr(a.wqkv, (i >> 1) * (C / 16) + 2 * head + (i & 1), C, lane) + 512 * (kh * KH);
    wq.start();
    SliceWarm warm;
    warm.init(grp, a.ngrp, wid, lane, smem + Z::A_WARM);
    warm.touch<NWV>(a.wqkv + (size_t)(2 * slice * HPW) * KS * 512, 3, 2 * HPW * KS, (C / 16) * KS);
    float* const AFF = (float*)(smem + Z::A_AFF);
    *(float4*)(AFF + 4 * tid) = affv;
    __syncthreads();
    pro.layernorm(AFF, a.eps, XN, a.xn1, a.mean1, a.rstd1, sv, wid, lane);
    size_t rows[G];
    int lab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { rows[g] = tm.row(16 * g + t); lab[g] = tm.label(16 * g + t); }
    DEEP_STAMP(1);
    __syncthreads();
    DEEP_STAMP(2);

    f32x4 acc[6][G];
    zero(acc);
    wq.template run<G, T>(acc, XN + kh * KH * (T * 64), t, gq);
    DEEP_STAMP(3);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) QX[((hl * 6 + i) * G + g) * 64 + lane] = acc[i][g];
    }
    __syncthreads();
    DEEP_STAMP(4);
    if (kh == 1) return;
    float rpb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rpb[r] = a.bias_table[a.rel_index[t * 16 + gq * 4 + r] * NH + head];
    bf16x4 qkvp[6][G];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = (i >> 1) * C + 32 * head + 16 * (i & 1) + 4 * gq;
        const f32x4 bqi = ld4(a.bqkv + n);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 s = acc[i][g] + QX[((hl * 6 + i) * G + g) * 64 + lane] + bqi;
            qkvp[i][g] = pack4(s[0], s[1], s[2], s[3]);
            if (a.save && (i & 1))
                store_bf16_tile_pair<true>(a.qkv + rows[g] * (3 * C) + n - 16 - 4 * gq, qkvp[i - 1][g], qkvp[i][g], gq);
        }
    }
    // ---- attention of this head, one window at a time (tulip.py:300-317); scores as K.Q^T: lane = query t, keys 4gq + r
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bf16x8 qf = cat8(qkvp[0][g], qkvp[1][g]);
        const bf16x8 kf = cat8(qkvp[2][g], qkvp[3][g]);
        *(bf16x4*)(ldsV + t * 64 + (4 * gq) * 2) = qkvp[4][g];
        *(bf16x4*)(ldsV + t * 64 + (16 + 4 * gq) * 2) = qkvp[5][g];
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        if (a.masked & TULIP_ATTN_FP8) sc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf16x8_to_fp8(kf), bf16x8_to_fp8(qf), sc, 0, 0, 0);
        else sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sc, 0, 0, 0);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = sc[r] * a.scale + rpb[r];
            if (a.masked & TULIP_ATTN_MASKED) {
                const int kl = __shfl(lab[g], gq * 4 + r, 64);
                if (kl != lab[g]) x += -100.0f;
            }
            sc[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = __expf(sc[r] - mx); sum += sc[r]; }
        sum = rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        const bf16x4 pb = pack4(sc[0] * inv, sc[1] * inv, sc[2] * inv, sc[3] * inv);
        bf16x4 opp[2];
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const bf16x4 vt = trr(ldsV + (gq * 4 + (t >> 2)) * 64 + dc * 32 + (t & 3) * 8);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb, o, 0, 0, 0);   // o[r] = O[t][16dc + 4gq + r]
            opp[dc] = pack4(o[0], o[1], o[2], o[3]);
        }
        store_bf16_tile_pair(a.o + rows[g] * C + 32 * head, opp[0], opp[1], gq);     // (read by F2: a plain store)
    }
    DEEP_STAMP(5);
}

// =====================================================================================================================
// F2 / F4 / B2 / B4: a Linear by OUTPUT channels.  Workgroup (group, slice): the slice's C/8 output channels (one 16-row tile per
// wave) of  out[T][C] = in[T][32 KT] . W[C][32 KT]^T  for the group's T tokens; the operand rows pass through LDS KS k tiles at a
// time (double-buffered: the next chunk's rows are fetched under the current chunk's MFMAs), the wave's weight row tile streams
// from L2 PF KiB blocks ahead.  EPI 0: out = aux + rowscale[sample] * (acc + bias), fp32 (+ bf16 copy); EPI 1: out = acc, fp32.
struct NsArgs {
    const bf16_t* in; const bf16_t* w;
    const float *bias, *aux, *rowscale;
    float* out; bf16_t* out_bf16;
    unsigned long long* prof;
    int B, H, W, wh, ww, sh, sw, ngrp;
};
template <int C, int G, int KT, int EPI>
__global__ __launch_bounds__((GeoD<C, G>::NT)) void deep_nslice_kernel(const NsArgs a) {
    using Z = GeoD<C, G>;
    constexpr int T = Z::T, KS = Z::KS, NWV = Z::NWV, NT = Z::NT, NCH = KT / KS, PF = 12;
    static_assert(KT % KS == 0 && KT >= PF, "k chunks");
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::N_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int slice = blockIdx.x & (NS - 1), grp = blockIdx.x >> 3;
    const TokMapD tm = make_map_d<G>(a.H, a.W, a.wh, a.ww, a.sh, a.sw, grp);
    DEEP_STAMP(0);
    RowStage<C, T, NT> st;
    st.load(a.in, 32 * KT, 0, tm, tid);
    const bf16_t* const wt = a.w + ((size_t)(slice * NWV + wid) * KT) * 512 + lane * 8;
    bf16x8 ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = *(const bf16x8*)(wt + 512 * p);
    SliceWarm warm;
    warm.init(grp, a.ngrp, wid, lane, smem + Z::N_WARM);
    warm.touch<NWV>(a.w + (size_t)(slice * NWV) * KT * 512, 1, NWV * KT, 0);
    st.store(smem, tid);
    DEEP_STAMP(1);
    __syncthreads();
    DEEP_STAMP(2);
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<NCH>([&](auto CH_) {
        constexpr int ch = decltype(CH_)::value;
        const unsigned char* const buf = smem + (ch & 1) * Z::XN_BYTES;
        if constexpr (ch + 1 < NCH) st.load(a.in, 32 * KT, (ch + 1) * C, tm, tid);
        static_for<KS>([&](auto K_) {
            constexpr int ks = decltype(K_)::value, step = ch * KS + ks;
            const bf16x8 wa = ring[step % PF];
            if constexpr (step + PF < KT) ring[step % PF] = *(const bf16x8*)(wt + 512 * (step + PF));
#pragma unroll
            for (int g = 0; g < G; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, frag<T>(buf, ks, 16 * g + t, gq), acc[g], 0, 0, 0);
        });
        if constexpr (ch + 1 < NCH) {
            st.store(smem + ((ch + 1) & 1) * Z::XN_BYTES, tid);
            __syncthreads();
        }
    });
    DEEP_STAMP(3);
    const int c0 = slice * (C / 8) + 16 * wid + 4 * gq;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const size_t row = tm.row(16 * g + t);
        f32x4 o = acc[g];
        if constexpr (EPI == 0) {
            const float sc = a.rowscale ? a.rowscale[tm.b] : 1.0f;
            o = ld4(a.aux + row * C + c0) + sc * (o + ld4(a.bias + c0));
        }
        *(float4*)(a.out + row * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (EPI == 0) {
            if (a.out_bf16) *(bf16x4*)(a.out_bf16 + row * C + c0) = pack4(o[0], o[1], o[2], o[3]);
        }
    }
    DEEP_STAMP(4);
}

// =====================================================================================================================
// F3: norm2 -> fc1 + exact-erf GELU of this slice's hidden channels (tulip.py:347, :195-196)
template <int C, int G>
__global__ __launch_bounds__((GeoD<C, G>::NT)) void deep_fc1_fwd_kernel(const DeepArgs a) {
    using Z = GeoD<C, G>;
    constexpr int T = Z::T, KS = Z::KS, NWV = Z::NWV, HID = Z::HID, HS = Z::HS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::B_SMEM];
    unsigned char* const XN = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int slice = blockIdx.x & (NS - 1), grp = blockIdx.x >> 3;
    const bool sv = a.save && slice == 0;
    const TokMapD tm = make_map_d<G>(a.H, a.W, a.wh, a.ww, a.sh, a.sw, grp);
    DEEP_STAMP(0);
    RowPro<C, T, NWV> pro;
    pro.load(a.x1, tm, wid, lane);
    const float4 affv = tid < C / 4 ? *(const float4*)(a.g2 + 4 * tid) : *(const float4*)(a.be2 + 4 * tid - C);
    // fc1: this wave's 64 of the slice's HS hidden channels (4 tiles)
    WS<4, KS, 3> w1s;
#pragma unroll
    for (int i = 0; i < 4; ++i) w1s.wt[i] = wtile_ptr(a.w1, slice * (HS / 16) + 4 * wid + i, C, lane);
    w1s.start();
    SliceWarm warm;
    warm.init(grp, a.ngrp, wid, lane, smem + Z::B_WARM);
    warm.touch<NWV>(a.w1 + (size_t)(slice * (HS / 16)) * KS * 512, 1, (HS / 16) * KS, 0);
    float* const AFF = (float*)(smem + Z::B_AFF);
    *(float4*)(AFF + 4 * tid) = affv;
    __syncthreads();
    pro.layernorm(AFF, a.eps, XN, a.xn2, a.mean2, a.rstd2, sv, wid, lane);
    size_t rows[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rows[g] = tm.row(16 * g + t);
    DEEP_STAMP(1);
    __syncthreads();
    DEEP_STAMP(2);
    f32x4 acc[4][G];
    zero(acc);
    w1s.template run<G, T>(acc, XN, t, gq);
    DEEP_STAMP(3);
    bf16x4 hprev[G], gprev[G];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = slice * HS + 64 * wid + 16 * i + 4 * gq;
        const f32x4 bb = ld4(a.b1 + n);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            bf16x4 hp = pack4(acc[i][g][0] + bb[0], acc[i][g][1] + bb[1], acc[i][g][2] + bb[2], acc[i][g][3] + bb[3]);
            const f32x2 h01 = {bf2f((bf16_t)hp[0]), bf2f((bf16_t)hp[1])}, h23 = {bf2f((bf16_t)hp[2]), bf2f((bf16_t)hp[3])};
            f32x2 g01, g23;
            if (a.save) {                       // gelu'(h) is all the backward wants from h (TULIP_BLOCK_FC1_GRAD)
                f32x2 d01, d23;
                gelu_exact_and_grad2(h01, g01, d01);
                gelu_exact_and_grad2(h23, g23, d23);
                hp = pack4(d01.x, d01.y, d23.x, d23.y);
            } else {
                g01 = gelu_exact2(h01); g23 = gelu_exact2(h23);
            }
            const bf16x4 gp = pack4(g01.x, g01.y, g23.x, g23.y);
            if (i & 1) {
                const size_t off = rows[g] * HID + slice * HS + 64 * wid + 16 * (i - 1);
                store_bf16_tile_pair(a.g + off, gprev[g], gp, gq);               // (read by F4: a plain store)
                if (a.save) store_bf16_tile_pair<true>(a.h + off, hprev[g], hp, gq);
            } else {
                hprev[g] = hp; gprev[g] = gp;
            }
        }
    }
    DEEP_STAMP(4);
}

// =====================================================================================================================
// B1: fc2' and GELU' of this slice's hidden channels (tulip.py:196-198 backwards): d(h) = (bf16(dy s_mlp) . W2)[hid] * gelu'(h)
template <int C, int G>
__global__ __launch_bounds__((GeoD<C, G>::NT)) void deep_fc2_bwd_kernel(const DeepArgs a) {
    using Z = GeoD<C, G>;
    constexpr int T = Z::T, KS = Z::KS, NWV = Z::NWV, HID = Z::HID, HS = Z::HS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::B_SMEM];
    unsigned char* const DY = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int slice = blockIdx.x & (NS - 1), grp = blockIdx.x >> 3;
    const TokMapD tm = make_map_d<G>(a.H, a.W, a.wh, a.ww, a.sh, a.sw, grp);
    const float s1v = a.ds1 ? a.ds1[tm.b] : 1.0f;
    DEEP_STAMP(0);
    RowPro<C, T, NWV> pro;
    pro.load(a.dx, tm, wid, lane);
    // fc2^T ([HID][C]): this wave's 64 of the slice's hidden channels
    WS<4, KS, 3> w2s;
#pragma unroll
    for (int i = 0; i < 4; ++i) w2s.wt[i] = wtile_ptr(a.w2, slice * (HS / 16) + 4 * wid + i, C, lane);
    w2s.start();
    SliceWarm warm;
    warm.init(grp, a.ngrp, wid, lane, smem + Z::B_WARM);
    warm.touch<NWV>(a.w2 + (size_t)(slice * (HS / 16)) * KS * 512, 1, (HS / 16) * KS, 0);
    // bf16(dy * s_mlp): operand of fc2's weight gradient (saved by the first slice) and of fc2'
    pro.scaled(s1v, DY, a.dyb_m, slice == 0, wid, lane);
    size_t rows[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rows[g] = tm.row(16 * g + t);
    DEEP_STAMP(1);
    __syncthreads();
    DEEP_STAMP(2);
    f32x4 acc[4][G];
    zero(acc);
    w2s.template run<G, T>(acc, DY, t, gq);
    DEEP_STAMP(3);
    bf16x4 hv[4][G];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g)
            hv[i][g] = ld_saved((const bf16x4*)(a.h + rows[g] * HID + slice * HS + 64 * wid + 16 * i + 4 * gq));
    bf16x4 dprev[G];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bf16x4 dp = pack4(acc[i][g][0] * bf2f((bf16_t)hv[i][g][0]), acc[i][g][1] * bf2f((bf16_t)hv[i][g][1]),
                                    acc[i][g][2] * bf2f((bf16_t)hv[i][g][2]), acc[i][g][3] * bf2f((bf16_t)hv[i][g][3]));
            if (i & 1) store_bf16_tile_pair(a.dh + rows[g] * HID + slice * HS + 64 * wid + 16 * (i - 1), dprev[g], dp, gq);
            else dprev[g] = dp;
        }
    }
    DEEP_STAMP(4);
}

// =====================================================================================================================
// B3: proj' of this slice's heads -> attention' (tulip.py:300-318 backwards) -> d(qkv) of the heads, dense bias-gradient partials
template <int C, int G>
__global__ __launch_bounds__((GeoD<C, G>::NT)) void deep_attn_bwd_kernel(const DeepArgs a) {
    using Z = GeoD<C, G>;
    constexpr int T = Z::T, KS = Z::KS, KH = Z::KH, NWV = Z::NWV, HPW = Z::HPW, NH = Z::NH, NT = Z::NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::A_SMEM];
    unsigned char* const DY = smem;
    f32x4* const PX = (f32x4*)(smem + Z::A_QX);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int slice = blockIdx.x & (NS - 1), grp = blockIdx.x >> 3;
    const int hl = wid % HPW, kh = wid / HPW, head = slice * HPW + hl;
    const TokMapD tm = make_map_d<G>(a.H, a.W, a.wh, a.ww, a.sh, a.sw, grp);
    DEEP_STAMP(0);
    RowStage<C, T, NT> st;
    st.load(a.dyb_a, C, 0, tm, tid);
    // proj^T ([C in][C out]): dO of this wave's head (two tiles) over its half of the contraction
    WS<2, KH, 6> wps;
#pragma unroll
    for (int i = 0; i < 2; ++i) wps.wt[i] = wtile_ptr(a.wproj, 2 * head + i, C, lane) + 512 * (kh * KH);
    wps.start();
    SliceWarm warm;
    warm.init(grp, a.ngrp, wid, lane, smem + Z::A_WARM);
    warm.touch<NWV>(a.wproj + (size_t)(2 * slice * HPW) * KS * 512, 1, 2 * HPW * KS, 0);
    st.store(DY, tid);
    size_t rows[G];
    int lab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { rows[g] = tm.row(16 * g + t); lab[g] = tm.label(16 * g + t); }
    DEEP_STAMP(1);
    __syncthreads();
    DEEP_STAMP(2);
    f32x4 acc[2][G];
    zero(acc);
    wps.template run<G, T>(acc, DY + kh * KH * (T * 64), t, gq);
    DEEP_STAMP(3);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) PX[((hl * 2 + i) * G + g) * 64 + lane] = acc[i][g];
    }
    __syncthreads();
    DEEP_STAMP(4);
    if (kh == 1) return;
    // relative-position bias seen from the query side (query t, key 4gq+r) and from the key side (query 4gq+r, key t)
    float bias_q[4], bias_k[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bias_q[r] = a.bias_table[a.rel_index[t * 16 + gq * 4 + r] * NH + head];
        bias_k[r] = a.bias_table[a.rel_index[(gq * 4 + r) * 16 + t] * NH + head];
    }
    bf16x4 dop[2][G];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 s = acc[i][g] + PX[((hl * 2 + i) * G + g) * 64 + lane];
            dop[i][g] = pack4(s[0], s[1], s[2], s[3]);
        }
    bf16x4 qkvr[6][G];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g)
            qkvr[i][g] = ld_saved((const bf16x4*)(a.qkv + rows[g] * (3 * C) + (i >> 1) * C + 32 * head + 16 * (i & 1) + 4 * gq));
    unsigned char* ldsQ = smem + Z::A_HT + hl * 3072;
    unsigned char* ldsK = ldsQ + 1024;
    unsigned char* ldsD = ldsK + 1024;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int troff = (gq * 4 + (t >> 2)) * 64 + (t & 3) * 8;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
        bf16x8 qf = cat8(qkvr[0][g], qkvr[1][g]), kf = cat8(qkvr[2][g], qkvr[3][g]);
        if (a.masked & TULIP_ATTN_FP8) { qf = round_through_fp8(qf); kf = round_through_fp8(kf); }   // what the forward's scores saw
        const bf16x8 vf = cat8(qkvr[4][g], qkvr[5][g]);
        const bf16x8 df = cat8(dop[0][g], dop[1][g]);
        const int o0 = t * 64 + (4 * gq) * 2, o1 = t * 64 + (16 + 4 * gq) * 2;
        *(bf16x4*)(ldsQ + o0) = __builtin_shufflevector(qf, qf, 0, 1, 2, 3);  *(bf16x4*)(ldsQ + o1) = __builtin_shufflevector(qf, qf, 4, 5, 6, 7);
        *(bf16x4*)(ldsK + o0) = __builtin_shufflevector(kf, kf, 0, 1, 2, 3);  *(bf16x4*)(ldsK + o1) = __builtin_shufflevector(kf, kf, 4, 5, 6, 7);
        *(bf16x4*)(ldsD + o0) = dop[0][g];   *(bf16x4*)(ldsD + o1) = dop[1][g];
        f32x4 sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);    // S[t][4gq+r]
        f32x4 sk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);    // S[4gq+r][t]
        f32x4 dpq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df, z, 0, 0, 0);   // dP[t][4gq+r]
        f32x4 dpk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, z, 0, 0, 0);   // dP[4gq+r][t]
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float xq = sq[r] * a.scale + bias_q[r];
            float xk = sk[r] * a.scale + bias_k[r];
            if (a.masked & TULIP_ATTN_MASKED) {
                const int ol = __shfl(lab[g], gq * 4 + r, 64);
                if (ol != lab[g]) { xq += -100.0f; xk += -100.0f; }
            }
            sq[r] = xq; sk[r] = xk;
            mx = fmaxf(mx, xq);
        }
        mx = rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += __expf(sq[r] - mx);
        sum = rows_sum(sum);
        const float lse = mx + __logf(sum);
        float pq[4], pk[4], delta = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pq[r] = __expf(sq[r] - lse);
            pk[r] = __expf(sk[r] - __shfl(lse, gq * 4 + r, 64));
            delta += pq[r] * dpq[r];
        }
        delta = rows_sum(delta);
        float dsq[4], dsk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dsq[r] = pq[r] * (dpq[r] - delta);
            dsk[r] = pk[r] * (dpk[r] - __shfl(delta, gq * 4 + r, 64));
            bsum[r] += dsq[r];
        }
        const bf16x4 dsq_b = pack4(dsq[0], dsq[1], dsq[2], dsq[3]);
        const bf16x4 dsk_b = pack4(dsk[0], dsk[1], dsk[2], dsk[3]);
        const bf16x4 pk_b = pack4(pk[0], pk[1], pk[2], pk[3]);
        bf16x4 oprev[3];
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const bf16x4 kt = trr(ldsK + troff + dc * 32);     // K[4gq+e][16dc+t]
            const bf16x4 qt = trr(ldsQ + troff + dc * 32);
            const bf16x4 dt = trr(ldsD + troff + dc * 32);
            const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, dsq_b, z, 0, 0, 0);   // dQ[t][16dc+4gq+r] / scale
            const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dsk_b, z, 0, 0, 0);
            const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dt, pk_b, z, 0, 0, 0);
            const bf16x4 o[3] = {pack4(dq[0] * a.scale, dq[1] * a.scale, dq[2] * a.scale, dq[3] * a.scale),
                                 pack4(dk[0] * a.scale, dk[1] * a.scale, dk[2] * a.scale, dk[3] * a.scale),
                                 pack4(dv[0], dv[1], dv[2], dv[3])};
#pragma unroll
            for (int sec = 0; sec < 3; ++sec) {
                if (dc) store_bf16_tile_pair(a.dqkv + rows[g] * (3 * C) + sec * C + 32 * head, oprev[sec], o[sec], gq);
                else oprev[sec] = o[sec];
            }
        }
    }
    // dense relative-position-bias gradient of this head, summed over the group's windows: [NH][16 q][16 k]
    *(float4*)(a.biaspart + (size_t)grp * (NH * 256) + head * 256 + t * 16 + gq * 4) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
    DEEP_STAMP(5);
}

// windows per group: two once that still gives every CU a workgroup (and the LDS holds them)
inline int deep_g(int C, int B, int H, int W, int wh, int ww) {
    const int windows = B * (H / wh) * (W / ww), nWx = W / ww;
    if (C == 768 && nWx % 2 == 0 && windows / 2 >= 32) return 2;
    return 1;
}
inline bool deep_ok(int C, int B, int H, int W, int wh, int ww) {
    return (C == 768 || C == 1536) && B > 0 && H > 0 && W > 0 && ((wh == 2 && ww == 8) || (wh == 1 && ww == 16)) && H % wh == 0 &&
           W % ww == 0;
}

template <int C, int G>
int launch_deep(const DeepArgs& a, int which, hipStream_t stream) {
    const dim3 grid(a.ngrp * NS), block(GeoD<C, G>::NT);
    switch (which) {
        case 0: hipLaunchKernelGGL((deep_attn_fwd_kernel<C, G>), grid, block, 0, stream, a); break;
        case 1: hipLaunchKernelGGL((deep_fc1_fwd_kernel<C, G>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((deep_fc2_bwd_kernel<C, G>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((deep_attn_bwd_kernel<C, G>), grid, block, 0, stream, a); break;
    }
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
int dispatch_deep(const DeepArgs& a, int C, int G, int which, hipStream_t stream) {
    if (C == 768) return G == 2 ? launch_deep<768, 2>(a, which, stream) : launch_deep<768, 1>(a, which, stream);
    return launch_deep<1536, 1>(a, which, stream);
}
template <int C, int G>
int launch_ns(const NsArgs& a, int ksec, int epi, hipStream_t stream) {
    const dim3 grid(a.ngrp * NS), block(GeoD<C, G>::NT);
    constexpr int KS = C / 32;
    if (epi == 0 && ksec == 1) hipLaunchKernelGGL((deep_nslice_kernel<C, G, KS, 0>), grid, block, 0, stream, a);
    else if (epi == 0 && ksec == 4) hipLaunchKernelGGL((deep_nslice_kernel<C, G, 4 * KS, 0>), grid, block, 0, stream, a);
    else if (epi == 1 && ksec == 4) hipLaunchKernelGGL((deep_nslice_kernel<C, G, 4 * KS, 1>), grid, block, 0, stream, a);
    else if (epi == 1 && ksec == 3) hipLaunchKernelGGL((deep_nslice_kernel<C, G, 3 * KS, 1>), grid, block, 0, stream, a);
    else return TULIP_ERR_ARG;
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
// ksec: width of the contraction in units of C (proj 1, fc2 / fc1' 4, qkv' 3)
int dispatch_ns(const NsArgs& a, int C, int G, int ksec, int epi, hipStream_t stream) {
    if (C == 768) return G == 2 ? launch_ns<768, 2>(a, ksec, epi, stream) : launch_ns<768, 1>(a, ksec, epi, stream);
    return launch_ns<1536, 1>(a, ksec, epi, stream);
}

}  // namespace

extern "C" int tulip_swind_supported(int C, int H, int W, int wh, int ww) { return deep_ok(C, 1, H, W, wh, ww) ? 1 : 0; }

extern "C" int tulip_swind_groups(int C, int B, int H, int W, int wh, int ww) {
    if (!deep_ok(C, B, H, W, wh, ww)) return 0;
    return B * (H / wh) * (W / ww) / deep_g(C, B, H, W, wh, ww);
}

extern "C" int tulip_swind_block_fwd(const tulip_swin96_desc* d, int C, int wh, int ww, void* out_bf16, int phases, uint64_t* stamps,
                                     hipStream_t stream) {
    if (!d || !deep_ok(C, d->B, d->H, d->W, wh, ww) || d->shift_h < 0 || d->shift_h >= d->H || d->shift_w < 0 || d->shift_w >= d->W ||
        !(phases & 15))
        return TULIP_ERR_ARG;
    // attn_out, x1 and fc1_act pass from one launch to the next: needed in the inference form too
    const bool any = d->xn1 || d->qkv || d->xn2 || d->fc1_pre || d->mean1 || d->rstd1 || d->mean2 || d->rstd2;
    const bool all = d->xn1 && d->qkv && d->xn2 && d->fc1_pre && d->mean1 && d->rstd1 && d->mean2 && d->rstd2;
    if ((any && !all) || !d->x1 || !d->attn_out || !d->fc1_act || !d->x_in || !d->x_out) return TULIP_ERR_ARG;
    if (all && !(d->masked & TULIP_BLOCK_FC1_GRAD)) return TULIP_ERR_ARG;      // the fc1_pre buffer receives gelu'(h)
    DeepArgs a = {};
    const int G = deep_g(C, d->B, d->H, d->W, wh, ww);
    a.ngrp = d->B * (d->H / wh) * (d->W / ww) / G;
    a.xin = d->x_in; a.x1 = d->x1; a.xout = d->x_out;
    a.xn1 = (bf16_t*)d->xn1; a.qkv = (bf16_t*)d->qkv; a.o = (bf16_t*)d->attn_out; a.xn2 = (bf16_t*)d->xn2;
    a.h = (bf16_t*)d->fc1_pre; a.g = (bf16_t*)d->fc1_act;
    a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.wqkv = (const bf16_t*)d->w_qkv; a.wproj = (const bf16_t*)d->w_proj; a.w1 = (const bf16_t*)d->w_fc1; a.w2 = (const bf16_t*)d->w_fc2;
    a.bqkv = d->b_qkv; a.bproj = d->b_proj; a.b1 = d->b_fc1; a.b2 = d->b_fc2;
    a.g1 = d->norm1_weight; a.be1 = d->norm1_bias; a.g2 = d->norm2_weight; a.be2 = d->norm2_bias;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.out_bf16 = (bf16_t*)out_bf16;
    a.prof = (unsigned long long*)stamps;
#if !TULIP_DEV_VARIANTS
    if (stamps) return TULIP_ERR_NOT_BUILT;
#endif
    a.B = d->B; a.H = d->H; a.W = d->W; a.wh = wh; a.ww = ww; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.save = all ? 1 : 0;
    a.eps = d->eps; a.scale = 0.17677669529663687f;        // head_dim^-0.5 = 32^-0.5 (tulip.py:220)
    NsArgs n = {};
    n.B = a.B; n.H = a.H; n.W = a.W; n.wh = wh; n.ww = ww; n.sh = a.sh; n.sw = a.sw; n.ngrp = a.ngrp;
    const size_t pstride = (size_t)a.ngrp * NS * (C / 128) * 16;
    int rc;
    if (phases & 1) { if ((rc = dispatch_deep(a, C, G, 0, stream))) return rc; }
    if (phases & 2) {       // proj + bias, DropPath, residual (tulip.py:318, :344)
        n.in = a.o; n.w = a.wproj; n.bias = a.bproj; n.aux = a.xin; n.rowscale = a.ds0; n.out = a.x1; n.out_bf16 = nullptr;
        n.prof = a.prof ? a.prof + pstride : nullptr;
        if ((rc = dispatch_ns(n, C, G, 1, 0, stream))) return rc;
    }
    if (phases & 4) {
        DeepArgs b = a;
        if (b.prof) b.prof += 2 * pstride;
        if ((rc = dispatch_deep(b, C, G, 1, stream))) return rc;
    }
    if (phases & 8) {       // fc2 + bias, DropPath, residual (tulip.py:198, :351)
        n.in = a.g; n.w = a.w2; n.bias = a.b2; n.aux = a.x1; n.rowscale = a.ds1; n.out = a.xout; n.out_bf16 = a.out_bf16;
        n.prof = a.prof ? a.prof + 3 * pstride : nullptr;
        if ((rc = dispatch_ns(n, C, G, 4, 0, stream))) return rc;
    }
    return TULIP_OK;
}

extern "C" int tulip_swind_block_bwd(const tulip_swin96_bwd_desc* d, int C, int wh, int ww, float* d_norm_out, int phases,
                                     uint64_t* stamps, hipStream_t stream) {
    if (!d || !deep_ok(C, d->B, d->H, d->W, wh, ww) || d->shift_h < 0 || d->shift_h >= d->H || d->shift_w < 0 || d->shift_w >= d->W ||
        !(phases & 15) || !(d->masked & TULIP_BLOCK_FC1_GRAD) || !d->qkv || !d->fc1_pre || !d_norm_out)
        return TULIP_ERR_ARG;
    DeepArgs a = {};
    const int G = deep_g(C, d->B, d->H, d->W, wh, ww);
    a.ngrp = d->B * (d->H / wh) * (d->W / ww) / G;
    a.dx = d->dx;
    a.qkv = (bf16_t*)d->qkv; a.h = (bf16_t*)d->fc1_pre;
    a.wqkv = (const bf16_t*)d->w_qkv; a.wproj = (const bf16_t*)d->w_proj; a.w1 = (const bf16_t*)d->w_fc1; a.w2 = (const bf16_t*)d->w_fc2;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.dyb_m = (bf16_t*)d->d_out_mlp; a.dh = (bf16_t*)d->d_fc1_pre; a.dyb_a = (bf16_t*)d->d_out_attn; a.dqkv = (bf16_t*)d->d_qkv;
    a.dxn = d_norm_out; a.biaspart = d->bias_partials;
    a.prof = (unsigned long long*)stamps;
#if !TULIP_DEV_VARIANTS
    if (stamps) return TULIP_ERR_NOT_BUILT;
#endif
    a.B = d->B; a.H = d->H; a.W = d->W; a.wh = wh; a.ww = ww; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.save = 1;
    a.scale = 0.17677669529663687f;
    NsArgs n = {};
    n.B = a.B; n.H = a.H; n.W = a.W; n.wh = wh; n.ww = ww; n.sh = a.sh; n.sw = a.sw; n.ngrp = a.ngrp;
    n.out = a.dxn;
    const size_t pstride = (size_t)a.ngrp * NS * (C / 128) * 16;
    int rc;
    if (phases & 1) { if ((rc = dispatch_deep(a, C, G, 2, stream))) return rc; }
    if (phases & 2) {       // fc1' (tulip.py:195 backwards): d(norm2 output), fp32
        n.in = a.dh; n.w = a.w1; n.prof = a.prof ? a.prof + pstride : nullptr;
        if ((rc = dispatch_ns(n, C, G, 4, 1, stream))) return rc;
    }
    if (phases & 4) {
        DeepArgs b = a;
        if (b.prof) b.prof += 2 * pstride;
        if ((rc = dispatch_deep(b, C, G, 3, stream))) return rc;
    }
    if (phases & 8) {       // qkv' (tulip.py:298 backwards): d(norm1 output), fp32
        n.in = a.dqkv; n.w = a.wqkv; n.prof = a.prof ? a.prof + 3 * pstride : nullptr;
        if ((rc = dispatch_ns(n, C, G, 3, 1, stream))) return rc;
    }
    return TULIP_OK;
}
