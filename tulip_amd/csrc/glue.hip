// The stage boundaries of the Swin U-Net -- PatchMerging, PatchUnmerging + skip Linear, and their backward mirrors -- as ONE launch
// each (round 6).  gfx950 only.
//
// Until round 5 these were the one part of the path still issued op by op, exactly as the reference's ATen stream is: LayerNorm with
// the 2x2 gather, then the reduction GEMM (tulip.py:101-106); the expand GEMM with its PixelShuffle scatter, then the skip
// Linear(cat[...]) (tulip.py:117-123, :713-715); and the same pairs (triples) in the backward -- 21 GEMM + 8 LayerNorm launches of
// 6-13 us "whatever the size" (profiles/r5_chain_gemms.txt).  Here a workgroup owns a block of 16 / 32 rows for the whole boundary:
// the row block's activations sit in LDS as the B operand of every MFMA (the fused block kernels' [k tile][token][64 B] layout,
// swin_stream.h), the weights stream from L2 straight into the A operand from their FRAGMENT-MAJOR copies (one contiguous 1-KiB wave
// load per MFMA operand; tulip_pack_bf16_multi, refreshed once per step with the block weights' copies), and what used to cross a
// kernel boundary through HBM crosses a __syncthreads() through LDS:
//
//   merge_fwd      x (fp32 stream) --2x2 gather + LayerNorm(4C)--> LDS panel --reduction GEMM--> next stage's fp32 stream
//                  (+ its bf16 copy into the skip concat buffer); xm / mean / rstd are still written (weight-gradient operand, backward)
//   merge_bwd      [dx + dy_skip . W_skip[:, C:]  (the x_save half of the skip Linear's input gradient, tulip.py:715 backwards)
//                  --> bf16 -->] reduction data gradient --> LayerNorm backward with the 2x2 scatter, d(gamma) / d(beta) partial rows
//   unmerge_skip_fwd   expand GEMM + bias --PixelShuffle(2), rounded to bf16 exactly where the two-launch form rounds it--> first half
//                  of the concat rows in LDS (+ written out: weight-gradient operand), x_save half loaded beside it --skip GEMM + bias-->
//                  decoder stage input.  Same rounding model as the reference's two autocast GEMMs (NOT their algebraic composition)
//   skip_unmerge_bwd   dy_skip . W_skip[:, :C] --inverse shuffle, bf16--> LDS (+ written out: weight-gradient operand) --expand
//                  data gradient--> coarse-level gradient (+ its bf16 copy for the next block's backward)
//
// A workgroup needs a whole ROW of the LayerNorm backward / of the second GEMM's contraction, so the row block is the unit and every
// workgroup streams the boundary's whole weight set (147 KB - 0.9 MB) through its CU: the forms exist where that is cheaper than the
// launches they replace (tulip_*_supported; the deepest boundary of each family, 512 rows against 2.4 MB of weights at batch 8, keeps
// the GEMM launches -- merge_fwd alone, whose LayerNorm prologue can be recomputed per column slice, covers it).
#include "swin_stream.h"
#include "tulip_hip.h"

namespace {

constexpr int PF = 4;          // weight stream runs this many 32-deep steps ahead of the MFMAs

// byte offset of channel c (a multiple of 8 for 16-B chunks) of token `tok` in a [k tile][T][64 B] panel
template <int T>
__device__ __forceinline__ int panel_off(int tok, int c) {
    return (c >> 5) * (T * 64) + tok * 64 + ((((c >> 3) & 3) ^ swz4(tok)) << 4) + (c & 7) * 2;
}

// PatchMerging gather (tulip.py:92-99): element offset of sub-position q (0: (0,0), 1: (1,0), 2: (0,1), 3: (1,1)) of merged row `row`
__device__ __forceinline__ size_t merge_src(int row, int q, int H, int W, int cin) {
    const int w2 = W >> 1, h2 = H >> 1;
    const int t = fast_div(row, w2), wq = row - t * w2;
    const int b = fast_div(t, h2), hq = t - b * h2;
    return (((size_t)b * H + 2 * hq + (q & 1)) * W + 2 * wq + (q >> 1)) * cin;
}

// fine token (2h+i, 2w+j), q = 2i + j, of coarse token m = (b, h, w) on an H x W coarse grid (PixelShuffle(2), tulip.py:120-122)
__device__ __forceinline__ size_t fine_row(int m, int q, int H, int W) {
    const int t = fast_div(m, W), w = m - t * W;
    const int b = fast_div(t, H), h = t - b * H;
    return ((size_t)b * 2 * H + 2 * h + (q >> 1)) * (2 * W) + 2 * w + (q & 1);
}

// L2 warm-up (the fused wide blocks' WeightWarm, csrc/swinw.hip): in the training step a boundary's weight copies were written a
// step ago and every workgroup's PF-deep fragment stream would walk them at HBM miss latency.  The first <= 256 workgroups of the
// launch that share an L2 (XCD = blockIdx % 8) and a weight region split it into KiB chunks and touch it with LDS-destination loads
// whose data is dropped (all in flight at once; they are older than the stream's own loads, so they have retired when those are
// waited for).  `j` of `sharers`: this workgroup's index among the ones that read the same region through the same L2.
#ifndef TULIP_GLUE_WARM
#define TULIP_GLUE_WARM 1
#endif
__device__ __forceinline__ void warm_region(const bf16_t* base, int kib, int j, int sharers, int wid, int lane, unsigned char* sink) {
    if (TULIP_GLUE_WARM && j < sharers)
        for (int c = j * 4 + wid; c < kib; c += sharers * 4) warm_touch16((const unsigned char*)(base + (size_t)c * 512) + lane * 16, sink);
}
// workgroups on this one's XCD among the first 256 of a launch whose workgroup b reads region b % nreg (nreg a power of two)
__device__ __forceinline__ void warm_geom(int nreg, int& j, int& sharers) {
    const int period = nreg > 8 ? nreg : 8;
    const int n = min((int)gridDim.x, 256);
    j = (int)blockIdx.x / period;
    sharers = max(n / period, 1);
    if ((int)blockIdx.x >= 256) j = sharers;        // (later workgroups find the L2 warm)
}

// The BACKWARD forms take a whole CU per workgroup (LDS padded to the CU's 160 KB).  They run beside the side queue's weight-gradient
// and fold launches, and every other kernel of the backward chain fills a CU by itself (a fused block workgroup holds 63-163 KB of LDS);
// skip_unmerge_bwd<192> -- 37 KB, 67 registers -- was the first chain kernel small enough to share a CU with the fold launches that take
// the optimizer step (reduce_rows_multi_kernel: 5 KB), and with it the captured step stopped being reproducible run to run: a few
// times per thousand fold launches, ONE component of `exp_avg` in 16 consecutive float4s (one register of one quarter-wave) came out as
// b1 m + g instead of b1 m + (1 - b1) g -- `exp_avg_sq` and the gradient sum untouched -- in the C = 96 weights whose fold ran beside it
// (tools/det_glue.py, tools/det_glue2.py: 8/8 repeats differ; no store-data hazard, no un-awaited load, no warm-up, no placement in the
// graph, no non-temporal access explains it; with the workgroup alone on its CU 0/16 differ).  Cause not found; the rule "a chain
// kernel of the backward never shares a CU with the side queue" is what every earlier kernel obeyed by size.  profiles/README.md round 6.
template <int USED>
struct WholeCU {
    static constexpr int PAD = 160 * 1024 - 512 - USED;
    static_assert(PAD > 0, "LDS");
    __device__ static __forceinline__ void take(int never_negative, float* out) {
        __shared__ unsigned char pad[PAD];
        if (never_negative < 0) {                          // (keeps the allocation; never executed)
            pad[threadIdx.x] = 1;
            __syncthreads();
            out[0] = pad[threadIdx.x ^ 1];
        }
    }
};

// one pass of a wave's GEMM: acc[i][g] = W tiles (first tile `tile0`, rows of a packed [*][K] matrix, k steps [ks0, ks0 + KS)) . panel
template <int NT, int G, int KS, int T>
struct WaveGemm {
    WStream<NT, KS, PF> ws;
    __device__ __forceinline__ void start(const bf16_t* wp, int tile0, int K, int ks0, int lane) {
#pragma unroll
        for (int i = 0; i < NT; ++i) ws.wt[i] = wtile_ptr(wp, tile0 + i, K, lane) + 512 * ks0;
        ws.start();
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[NT][G], const unsigned char* panel, int ks0, int tok0, int t, int gq) {
        ws.template run<G, T>(acc, panel + ks0 * (T * 64) + tok0 * 64, t, gq);      // (tok0 a multiple of 16: the swizzle has period 16)
    }
};

// ------------------------------------------------------------------------------------------------ PatchMerging forward
struct MergeFwdArgs {
    const float* x; const float* gamma; const float* beta; const bf16_t* wp;
    bf16_t* xm; float* mean; float* rstd; float* y; bf16_t* y16; int ld16;
    int B, H, W, rows; float eps;
    int xcd_rows;
};

// CIN input channels; K = 4 CIN, N = 2 CIN.  Workgroup: BM merged rows x NWG = (4 / KSPLIT) NT 16 output columns (column slice
// blockIdx % NSL: the slices of a row block sit on different XCDs and each XCD's L2 holds its slices' weights); the four waves are
// 4 / KSPLIT along N times KSPLIT along K.  Every slice recomputes the LayerNorm of its row block (the rows come from L2 after the
// first slice); slice 0 writes xm / mean / rstd.
template <int CIN, int BM, int NT, int KSPLIT>
__global__ __launch_bounds__(256) void merge_fwd_kernel(const MergeFwdArgs a) {
    constexpr int K = 4 * CIN, N = 2 * CIN, NW = 4 / KSPLIT, NWG = NW * NT * 16, NSL = N / NWG;
    static_assert(N % NWG == 0 && (K / 32) % KSPLIT == 0 && K % 64 == 0 && BM % 16 == 0, "shape");
    constexpr int KSW = K / 32 / KSPLIT, G = BM / 16, NCH = K / 64, PASSES = BM / 16;
    constexpr int PITCH = NWG * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char panel[BM * K * 2];
    __shared__ __attribute__((aligned(16))) unsigned char stg[BM * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char sink[1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    // workgroup -> (row block, column slice).  Row blocks are XCD-affine (XCD = blockIdx % 8 holds row blocks x, x + 8, ...; all their
    // slices follow one another on it): the rows are fetched into ONE L2 and every XCD streams the whole weight matrix -- with the slices
    // of a row block spread over the XCDs instead, each slice's XCD fetched the rows again (16 x 3 MB at the deepest level against
    // 8 x 2.4 MB of weights this way).  Needs the row-block count to be a multiple of 8 (the launcher falls back to slice-minor otherwise).
    int slice, rb;
    if (a.xcd_rows) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, rpx = (a.rows / BM) >> 3;     // row blocks per XCD
        slice = j / rpx;
        rb = (j - slice * rpx) * 8 + x;
    } else {
        slice = blockIdx.x % NSL;
        rb = blockIdx.x / NSL;
    }
    const int row0 = rb * BM;
    const int wn = wid % NW, wk = wid / NW;
    WaveGemm<NT, G, KSW, BM> gm;
    gm.start(a.wp, (slice * NWG) / 16 + wn * NT, K, wk * KSW, lane);
    {
        int j, sh;
        if (a.xcd_rows) {
            warm_geom(1, j, sh);
            warm_region(a.wp, N * K * 2 / 1024, j, sh, wid, lane, sink);
        } else {
            warm_geom(NSL, j, sh);
            warm_region(a.wp + (size_t)slice * NWG * K, NWG * K * 2 / 1024, j, sh, wid, lane, sink);
        }
    }
    // ---- gather + LayerNorm: 16 lanes per row, 4 rows per wave and pass
    const float invK = 1.0f / (float)K;
#pragma unroll 1
    for (int ps = 0; ps < PASSES; ++ps) {
        const int rl = ps * 16 + wid * 4 + gq, row = row0 + rl;
        size_t base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = merge_src(row, q, a.H, a.W, CIN);
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = 4 * (t + 16 * i), q = e / CIN, ci = e - q * CIN;
            const size_t o = (q == 0 ? base[0] : q == 1 ? base[1] : q == 2 ? base[2] : base[3]) + ci;
            v[i] = *(const float4*)(a.x + o);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = group_sum<16>(s) * invK;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float p = v[i].x - mu, q = v[i].y - mu, r = v[i].z - mu, u = v[i].w - mu;
            ss += (p * p + q * q) + (r * r + u * u);
        }
        const float rs = rsqrtf(group_sum<16>(ss) * invK + a.eps);
        if (slice == 0 && t == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = 4 * (t + 16 * i);
            const float4 ga = *(const float4*)(a.gamma + e), be = *(const float4*)(a.beta + e);
            const bf16x4 o = pack4((v[i].x - mu) * rs * ga.x + be.x, (v[i].y - mu) * rs * ga.y + be.y,
                                   (v[i].z - mu) * rs * ga.z + be.z, (v[i].w - mu) * rs * ga.w + be.w);
            *(bf16x4*)(panel + panel_off<BM>(rl, e)) = o;
            if (slice == 0) *(bf16x4*)(a.xm + (size_t)row * K + e) = o;
        }
    }
    __syncthreads();
    // ---- reduction GEMM
    f32x4 acc[NT][G];
    zero(acc);
    gm.run(acc, panel, wk * KSW, 0, t, gq);
#pragma unroll
    for (int kk = 0; kk < KSPLIT; ++kk) {
        if (wk == kk) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f32x4* p = (f32x4*)(stg + (16 * g + t) * PITCH + (wn * NT * 16 + 16 * i + 4 * gq) * 4);
                    *p = kk == 0 ? acc[i][g] : *p + acc[i][g];
                }
        }
        __syncthreads();
    }
    for (int c = tid; c < BM * (NWG / 4); c += 256) {
        const int rl = c / (NWG / 4), c4 = c - rl * (NWG / 4);
        const float4 v = *(const float4*)(stg + rl * PITCH + c4 * 16);
        const size_t row = row0 + rl;
        const int n = slice * NWG + c4 * 4;
        *(float4*)(a.y + row * N + n) = v;
        if (a.y16) *(bf16x4*)(a.y16 + row * a.ld16 + n) = pack4(v.x, v.y, v.z, v.w);
    }
}

// ------------------------------------------------------------------------------------------------ PatchMerging backward
struct MergeBwdArgs {
    const float* dx_in;       // SKIP: fp32 [rows][CS] gradient w.r.t. the stage input that came down the encoder
    const bf16_t* dys;        // SKIP: bf16 [rows][CS] gradient w.r.t. the skip Linear's output
    const bf16_t* w2t;        // SKIP: packed copy of W_skip^T ([2 CS][CS]); rows CS.. are the x_save half
    bf16_t* dyb;              // bf16 [rows][CS]: written (SKIP) / read: gradient w.r.t. the PatchMerging output
    const bf16_t* wrt;        // packed copy of reduction.weight^T ([4 CP][CS])
    const float* xprev; const float* mean; const float* rstd; const float* gamma;
    float* dxp;               // fp32 (B,H,W,CP): gradient w.r.t. the previous stage's output (overwritten)
    float* part;              // [gridDim.x][2 * 4 CP] partial rows [dgamma | dbeta]
    bf16_t* ycast; const float* cscale; int crps;
    int B, H, W, rows;        // H, W: grid of the PREVIOUS (finer) stage
};

template <int CP, int BM, bool SKIP>
__global__ __launch_bounds__(256) void merge_bwd_kernel(const MergeBwdArgs a) {
    constexpr int CS = 2 * CP, K4 = 4 * CP, G = BM / 16;
    constexpr int NTA = 3, PASS_A = CS / 64 / NTA;           // skip half: N = CS, four waves along N, NTA tiles per wave and pass
    constexpr int NTB = 6, PASS_B = K4 / 64 / NTB;           // reduction data gradient: N = 4 CP
    static_assert(CS % (64 * NTA) == 0 && K4 % (64 * NTB) == 0, "shape");
    constexpr int KS = CS / 32, NCH = K4 / 64;
    constexpr int DY_PITCH = K4 * 2 + 16, XH_PITCH = K4 * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char panelA[SKIP ? BM * CS * 2 : 16];
    __shared__ __attribute__((aligned(16))) unsigned char panelB[BM * CS * 2];
    __shared__ __attribute__((aligned(16))) unsigned char dyS[BM * DY_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char xhS[BM * XH_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char sink[1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int row0 = blockIdx.x * BM;
    WholeCU<(SKIP ? BM * CS * 2 : 16) + BM * CS * 2 + BM * DY_PITCH + BM * XH_PITCH + 1024>::take(a.rows, a.dxp);
    int wj, wsh;
    warm_geom(1, wj, wsh);
    if constexpr (SKIP) {
        WaveGemm<NTA, G, KS, BM> ga;
        ga.start(a.w2t, CS / 16 + wid * (CS / 64), CS, 0, lane);
        warm_region(a.w2t + (size_t)CS * CS, CS * CS * 2 / 1024, wj, wsh, wid, lane, sink);
        warm_region(a.wrt, K4 * CS * 2 / 1024, wj, wsh, wid, lane, sink);
        for (int c = tid; c < BM * (CS / 8); c += 256) {
            const int rl = c / (CS / 8), c8 = c - rl * (CS / 8);
            *(uint4*)(panelA + panel_off<BM>(rl, c8 * 8)) = *(const uint4*)(a.dys + (size_t)(row0 + rl) * CS + c8 * 8);
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASS_A; ++ps) {
            const int c0 = wid * (CS / 4) + ps * NTA * 16;
            f32x4 acc[NTA][G];
#pragma unroll
            for (int i = 0; i < NTA; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[i][g] = ld4(a.dx_in + (size_t)(row0 + 16 * g + t) * CS + c0 + 16 * i + 4 * gq);
            if (ps > 0) ga.start(a.w2t, CS / 16 + c0 / 16, CS, 0, lane);
            ga.run(acc, panelA, 0, 0, t, gq);
#pragma unroll
            for (int i = 0; i < NTA; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    *(bf16x4*)(panelB + panel_off<BM>(16 * g + t, c0 + 16 * i + 4 * gq)) = pack4(acc[i][g][0], acc[i][g][1], acc[i][g][2], acc[i][g][3]);
        }
        __syncthreads();
        for (int c = tid; c < BM * (CS / 8); c += 256) {         // the bf16 sum leaves: operand of the reduction's weight gradient
            const int rl = c / (CS / 8), c8 = c - rl * (CS / 8);
            *(uint4*)(a.dyb + (size_t)(row0 + rl) * CS + c8 * 8) = *(const uint4*)(panelB + panel_off<BM>(rl, c8 * 8));
        }
    } else {
        warm_region(a.wrt, K4 * CS * 2 / 1024, wj, wsh, wid, lane, sink);
        for (int c = tid; c < BM * (CS / 8); c += 256) {
            const int rl = c / (CS / 8), c8 = c - rl * (CS / 8);
            *(uint4*)(panelB + panel_off<BM>(rl, c8 * 8)) = *(const uint4*)(a.dyb + (size_t)(row0 + rl) * CS + c8 * 8);
        }
        __syncthreads();
    }
    // ---- d(LayerNorm output)[BM][4 CP] = dyb . W_red, rounded to bf16 where the GEMM launch stored it
    {
        WaveGemm<NTB, G, KS, BM> gb;
#pragma unroll
        for (int ps = 0; ps < PASS_B; ++ps) {
            const int e0 = wid * (K4 / 4) + ps * NTB * 16;
            gb.start(a.wrt, e0 / 16, CS, 0, lane);
            f32x4 acc[NTB][G];
            zero(acc);
            gb.run(acc, panelB, 0, 0, t, gq);
#pragma unroll
            for (int i = 0; i < NTB; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    *(bf16x4*)(dyS + (16 * g + t) * DY_PITCH + (e0 + 16 * i + 4 * gq) * 2) = pack4(acc[i][g][0], acc[i][g][1], acc[i][g][2], acc[i][g][3]);
        }
    }
    __syncthreads();
    // ---- LayerNorm backward, 16 lanes per row (tulip_layernorm_bwd's arithmetic), scatter through the 2x2 gather
    const float invK = 1.0f / (float)K4;
#pragma unroll 1
    for (int ps = 0; ps < G; ++ps) {
        const int rl = ps * 16 + wid * 4 + gq, row = row0 + rl;
        const float mu = a.mean[row], rs = a.rstd[row];
        size_t base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = merge_src(row, q, a.H, a.W, CP);
        float4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = 4 * (t + 16 * i), q = e / CP, ci = e - q * CP;
            const size_t o = (q == 0 ? base[0] : q == 1 ? base[1] : q == 2 ? base[2] : base[3]) + ci;
            const float4 xv = *(const float4*)(a.xprev + o);
            const uint2 d = *(const uint2*)(dyS + rl * DY_PITCH + e * 2);
            const float4 ga = *(const float4*)(a.gamma + e);
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            *(float4*)(xhS + rl * XH_PITCH + e * 4) = xh[i];
            const float d0 = bf2f((bf16_t)(d.x & 0xffff)), d1 = bf2f((bf16_t)(d.x >> 16));
            const float d2 = bf2f((bf16_t)(d.y & 0xffff)), d3 = bf2f((bf16_t)(d.y >> 16));
            gy[i] = make_float4(d0 * ga.x, d1 * ga.y, d2 * ga.z, d3 * ga.w);
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
        }
        const float m1 = group_sum<16>(s1) * invK, m2 = group_sum<16>(s2) * invK;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = 4 * (t + 16 * i), q = e / CP, ci = e - q * CP;
            const size_t o = (q == 0 ? base[0] : q == 1 ? base[1] : q == 2 ? base[2] : base[3]) + ci;
            const float4 r = make_float4(rs * (gy[i].x - m1 - xh[i].x * m2), rs * (gy[i].y - m1 - xh[i].y * m2),
                                         rs * (gy[i].z - m1 - xh[i].z * m2), rs * (gy[i].w - m1 - xh[i].w * m2));
            *(float4*)(a.dxp + o) = r;
            if (a.ycast) {
                const float sc = a.cscale ? a.cscale[fast_div((int)(o / CP), a.crps)] : 1.0f;
                *(bf16x4*)(a.ycast + o) = pack4(r.x * sc, r.y * sc, r.z * sc, r.w * sc);
            }
        }
    }
    __syncthreads();
    // ---- d(gamma), d(beta) of this row block: one partial row [2 * 4 CP]
    for (int c = tid; c < K4 / 4; c += 256) {
        float4 pg = make_float4(0, 0, 0, 0), pb = make_float4(0, 0, 0, 0);
#pragma unroll 4
        for (int r = 0; r < BM; ++r) {
            const float4 x4 = *(const float4*)(xhS + r * XH_PITCH + c * 16);
            const uint2 d = *(const uint2*)(dyS + r * DY_PITCH + c * 8);
            const float d0 = bf2f((bf16_t)(d.x & 0xffff)), d1 = bf2f((bf16_t)(d.x >> 16));
            const float d2 = bf2f((bf16_t)(d.y & 0xffff)), d3 = bf2f((bf16_t)(d.y >> 16));
            pg.x += d0 * x4.x; pg.y += d1 * x4.y; pg.z += d2 * x4.z; pg.w += d3 * x4.w;
            pb.x += d0; pb.y += d1; pb.z += d2; pb.w += d3;
        }
        float* o = a.part + (size_t)blockIdx.x * (2 * K4) + c * 4;
        *(float4*)o = pg;
        *(float4*)(o + K4) = pb;
    }
}

// ------------------------------------------------------------------------------------------------ PatchUnmerging -> skip Linear
struct UnmergeSkipArgs {
    const bf16_t* xb;         // bf16 [M][C]: the coarse stage's output
    const bf16_t* wexp; const float* bexp;     // packed expand weight [2C][C], bias [2C]
    bf16_t* cat;              // bf16 [4M][C]: fine-level concat rows; [:, :C/2] written here, [:, C/2:] (x_save) read
    const bf16_t* wskip; const float* bskip;   // packed skip weight [C/2][C], bias [C/2]
    float* out;               // fp32 [4M][C/2]: decoder stage input
    int B, H, W, M;           // coarse grid
};

template <int C>
__global__ __launch_bounds__(256) void unmerge_skip_fwd_kernel(const UnmergeSkipArgs a) {
    constexpr int CF = C / 2, BM = 16, TF = 64;
    constexpr int NT1 = 6, PASS1 = (2 * C) / 64 / NT1;       // expand: N = 2C, four waves along N
    constexpr int NT2 = CF / 32;                             // skip: N = CF over two waves, 64 fine rows over two waves
    static_assert((2 * C) % (64 * NT1) == 0 && CF % 32 == 0, "shape");
    constexpr int PITCH = CF * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char panelX[BM * C * 2];
    __shared__ __attribute__((aligned(16))) unsigned char panelC[TF * C * 2];
    __shared__ __attribute__((aligned(16))) unsigned char stg[TF * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char sink[1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    WaveGemm<NT1, 1, C / 32, BM> g1;
    g1.start(a.wexp, wid * (2 * C / 64), C, 0, lane);
    {
        int j, sh;
        warm_geom(1, j, sh);
        warm_region(a.wexp, 2 * C * C * 2 / 1024, j, sh, wid, lane, sink);
        warm_region(a.wskip, CF * C * 2 / 1024, j, sh, wid, lane, sink);
    }
    for (int c = tid; c < BM * (C / 8); c += 256) {
        const int rl = c / (C / 8), c8 = c - rl * (C / 8);
        *(uint4*)(panelX + panel_off<BM>(rl, c8 * 8)) = *(const uint4*)(a.xb + (size_t)(m0 + rl) * C + c8 * 8);
    }
    for (int c = tid; c < TF * (CF / 8); c += 256) {            // x_save half of the 64 fine rows
        const int f = c / (CF / 8), c8 = c - f * (CF / 8);
        const size_t fr = fine_row(m0 + (f >> 2), f & 3, a.H, a.W);
        *(uint4*)(panelC + panel_off<TF>(f, CF + c8 * 8)) = *(const uint4*)(a.cat + fr * C + CF + c8 * 8);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASS1; ++ps) {
        const int n0 = wid * (2 * C / 4) + ps * NT1 * 16;
        if (ps > 0) g1.start(a.wexp, n0 / 16, C, 0, lane);
        f32x4 acc[NT1][1];
        zero(acc);
        g1.run(acc, panelX, 0, 0, t, gq);
#pragma unroll
        for (int i = 0; i < NT1; ++i) {
            // column n = 4 c + q of coarse token t  ->  fine row 4 t + q, channel c (PixelShuffle(2) + BCHW -> BHWC)
            const int n = n0 + 16 * i + 4 * gq, cch = n >> 2;
            const float4 b = *(const float4*)(a.bexp + n);
            const float v[4] = {acc[i][0][0] + b.x, acc[i][0][1] + b.y, acc[i][0][2] + b.z, acc[i][0][3] + b.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) *(bf16_t*)(panelC + panel_off<TF>(4 * t + q, cch)) = f2bf(v[q]);
        }
    }
    __syncthreads();
    WaveGemm<NT2, 2, C / 32, TF> g2;
    const int wn = wid & 1, wt = wid >> 1;
    g2.start(a.wskip, wn * NT2, C, 0, lane);
    for (int c = tid; c < TF * (CF / 8); c += 256) {            // first half of the concat rows leaves (weight-gradient operand)
        const int f = c / (CF / 8), c8 = c - f * (CF / 8);
        const size_t fr = fine_row(m0 + (f >> 2), f & 3, a.H, a.W);
        *(uint4*)(a.cat + fr * C + c8 * 8) = *(const uint4*)(panelC + panel_off<TF>(f, c8 * 8));
    }
    f32x4 acc[NT2][2];
    zero(acc);
    g2.run(acc, panelC, 0, 32 * wt, t, gq);
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
        const int n = wn * (CF / 2) + 16 * i + 4 * gq;
        const f32x4 b = ld4(a.bskip + n);
#pragma unroll
        for (int g = 0; g < 2; ++g) *(f32x4*)(stg + (32 * wt + 16 * g + t) * PITCH + n * 4) = acc[i][g] + b;
    }
    __syncthreads();
    for (int c = tid; c < TF * (CF / 4); c += 256) {
        const int f = c / (CF / 4), c4 = c - f * (CF / 4);
        const size_t fr = fine_row(m0 + (f >> 2), f & 3, a.H, a.W);
        *(float4*)(a.out + fr * CF + c4 * 4) = *(const float4*)(stg + f * PITCH + c4 * 16);
    }
}

// ------------------------------------------------------------------------------------------------ its backward
struct SkipUnmergeBwdArgs {
    const bf16_t* dys;        // bf16 [4M][C/2]: gradient w.r.t. the skip Linear's output (fine level)
    const bf16_t* w1t;        // packed copy of W_skip^T ([C][C/2]); rows 0 .. C/2-1 are the unmerged-stream half
    bf16_t* dz2;              // bf16 [M][2C]: gradient w.r.t. the expand conv's output, un-shuffled (written: weight-gradient operand)
    const bf16_t* wet;        // packed copy of expand.weight^T ([C][2C])
    float* dx;                // fp32 [M][C]: gradient w.r.t. the coarse stage's output (overwritten)
    bf16_t* ycast; const float* cscale; int crps;
    int B, H, W, M;
};

template <int C>
__global__ __launch_bounds__(256) void skip_unmerge_bwd_kernel(const SkipUnmergeBwdArgs a) {
    constexpr int CF = C / 2, BM = 16, TF = 64;
    constexpr int NTA = CF / 32;                              // N = CF over two waves, 64 fine rows over two waves
    constexpr int NTB = 3, PASS_B = C / 64 / NTB;             // N = C, four waves along N
    static_assert(C % (64 * NTB) == 0, "shape");
    constexpr int PITCH = C * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char panelD[TF * CF * 2];
    __shared__ __attribute__((aligned(16))) unsigned char panelZ[BM * 2 * C * 2];
    __shared__ __attribute__((aligned(16))) unsigned char stg[BM * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char sink[1024];
    WholeCU<TF * CF * 2 + BM * 2 * C * 2 + BM * PITCH + 1024>::take(a.M, a.dx);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int wn = wid & 1, wt = wid >> 1;
    WaveGemm<NTA, 2, CF / 32, TF> ga;
    ga.start(a.w1t, wn * NTA, CF, 0, lane);
    {
        int j, sh;
        warm_geom(1, j, sh);
        warm_region(a.w1t, CF * CF * 2 / 1024, j, sh, wid, lane, sink);
        warm_region(a.wet, 2 * C * C * 2 / 1024, j, sh, wid, lane, sink);
    }
    for (int c = tid; c < TF * (CF / 8); c += 256) {
        const int f = c / (CF / 8), c8 = c - f * (CF / 8);
        const size_t fr = fine_row(m0 + (f >> 2), f & 3, a.H, a.W);
        *(uint4*)(panelD + panel_off<TF>(f, c8 * 8)) = *(const uint4*)(a.dys + fr * CF + c8 * 8);
    }
    __syncthreads();
    {
        f32x4 acc[NTA][2];
        zero(acc);
        ga.run(acc, panelD, 0, 32 * wt, t, gq);
#pragma unroll
        for (int i = 0; i < NTA; ++i)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // fine row f = (coarse f >> 2, q = f & 3), channel c  ->  column 4 c + q of the coarse row (inverse PixelShuffle(2))
                const int f = 32 * wt + 16 * g + t, cch = wn * (CF / 2) + 16 * i + 4 * gq;
#pragma unroll
                for (int r = 0; r < 4; ++r) *(bf16_t*)(panelZ + panel_off<BM>(f >> 2, 4 * (cch + r) + (f & 3))) = f2bf(acc[i][g][r]);
            }
    }
    __syncthreads();
    WaveGemm<NTB, 1, 2 * C / 32, BM> gb;
    gb.start(a.wet, wid * (C / 64), 2 * C, 0, lane);
    for (int c = tid; c < BM * (2 * C / 8); c += 256) {
        const int rl = c / (2 * C / 8), c8 = c - rl * (2 * C / 8);
        *(uint4*)(a.dz2 + (size_t)(m0 + rl) * (2 * C) + c8 * 8) = *(const uint4*)(panelZ + panel_off<BM>(rl, c8 * 8));
    }
#pragma unroll
    for (int ps = 0; ps < PASS_B; ++ps) {
        const int n0 = wid * (C / 4) + ps * NTB * 16;
        if (ps > 0) gb.start(a.wet, n0 / 16, 2 * C, 0, lane);
        f32x4 acc[NTB][1];
        zero(acc);
        gb.run(acc, panelZ, 0, 0, t, gq);
#pragma unroll
        for (int i = 0; i < NTB; ++i) *(f32x4*)(stg + t * PITCH + (n0 + 16 * i + 4 * gq) * 4) = acc[i][0];
    }
    __syncthreads();
    for (int c = tid; c < BM * (C / 4); c += 256) {
        const int rl = c / (C / 4), c4 = c - rl * (C / 4);
        const float4 v = *(const float4*)(stg + rl * PITCH + c4 * 16);
        const size_t o = (size_t)(m0 + rl) * C + c4 * 4;
        *(float4*)(a.dx + o) = v;
        if (a.ycast) {
            const float sc = a.cscale ? a.cscale[fast_div(m0 + rl, a.crps)] : 1.0f;
            *(bf16x4*)(a.ycast + o) = pack4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
    }
}

bool grid_ok(int B, int H, int W) { return B > 0 && H > 0 && W > 0 && (long long)B * H * W < (1ll << 29); }

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------------
extern "C" int tulip_merge_fwd_supported(int Cin, int B, int H, int W) {
    if (!grid_ok(B, H, W) || (H & 1) || (W & 1)) return 0;
    const int rows = B * (H / 2) * (W / 2);
    return (Cin == 96 || Cin == 192 || Cin == 384) && rows % 32 == 0;
}

extern "C" int tulip_merge_fwd(const tulip_merge_fwd_desc* d, hipStream_t stream) {
    if (!d || !tulip_merge_fwd_supported(d->Cin, d->B, d->H, d->W)) return TULIP_ERR_ARG;
    if (!d->x || !d->gamma || !d->beta || !d->w_packed || !d->xm || !d->mean || !d->rstd || !d->y) return TULIP_ERR_ARG;
    if (d->y_bf16 && (d->ld_bf16 & 3)) return TULIP_ERR_ARG;
    MergeFwdArgs a{d->x, d->gamma, d->beta, (const bf16_t*)d->w_packed, (bf16_t*)d->xm, d->mean, d->rstd, d->y,
                   (bf16_t*)d->y_bf16, d->ld_bf16, d->B, d->H, d->W, d->B * (d->H / 2) * (d->W / 2), d->eps, 0};
    const int rows = a.rows;
    // (row block, column slices) per width: few rows -> small row blocks and narrow slices, so that the launch has >= 256 workgroups
    // and a workgroup streams ~150 KB of weights; many rows -> the widest slice (fewest redundant LayerNorms, least L2 -> CU traffic)
#define MF(CIN, BM, NT, KSPLIT) do { \
        constexpr int NSL = (2 * CIN) / ((4 / KSPLIT) * NT * 16); \
        a.xcd_rows = NSL > 1 && (rows / BM) % 8 == 0; \
        hipLaunchKernelGGL((merge_fwd_kernel<CIN, BM, NT, KSPLIT>), dim3((rows / BM) * NSL), dim3(256), 0, stream, a); } while (0)
    if (d->Cin == 96) MF(96, 32, 3, 1);
    else if (d->Cin == 192) { if (rows >= 8192) MF(192, 32, 3, 1); else MF(192, 16, 3, 2); }
    else { if (rows >= 4096) MF(384, 16, 3, 1); else MF(384, 16, 3, 4); }
#undef MF
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_merge_bwd_supported(int Cp, int B, int H, int W) {
    if (!grid_ok(B, H, W) || (H & 1) || (W & 1)) return 0;
    const int rows = B * (H / 2) * (W / 2);
    return (Cp == 96 || Cp == 192) && rows % 32 == 0;
}
extern "C" int tulip_merge_bwd_partial_rows(int Cp, int B, int H, int W) {
    if (!tulip_merge_bwd_supported(Cp, B, H, W)) return 0;
    return B * (H / 2) * (W / 2) / (Cp == 96 ? 32 : 16);
}

extern "C" int tulip_merge_bwd(const tulip_merge_bwd_desc* d, hipStream_t stream) {
    if (!d || !tulip_merge_bwd_supported(d->Cp, d->B, d->H, d->W)) return TULIP_ERR_ARG;
    const bool skip = d->dy_skip != nullptr;
    if (!d->dyb || !d->w_red_t_packed || !d->x_prev || !d->mean || !d->rstd || !d->gamma || !d->dx_prev || !d->param_partials)
        return TULIP_ERR_ARG;
    if (skip && (!d->dx_in || !d->w_skip_t_packed)) return TULIP_ERR_ARG;
    if (d->dx_bf16 && d->cast_rowscale && d->cast_rows_per_sample <= 0) return TULIP_ERR_ARG;
    MergeBwdArgs a{d->dx_in, (const bf16_t*)d->dy_skip, (const bf16_t*)d->w_skip_t_packed, (bf16_t*)d->dyb,
                   (const bf16_t*)d->w_red_t_packed, d->x_prev, d->mean, d->rstd, d->gamma, d->dx_prev, d->param_partials,
                   (bf16_t*)d->dx_bf16, d->cast_rowscale, d->cast_rows_per_sample, d->B, d->H, d->W, d->B * (d->H / 2) * (d->W / 2)};
    const int grid = tulip_merge_bwd_partial_rows(d->Cp, d->B, d->H, d->W);
    if (d->Cp == 96) {
        if (skip) hipLaunchKernelGGL((merge_bwd_kernel<96, 32, true>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((merge_bwd_kernel<96, 32, false>), dim3(grid), dim3(256), 0, stream, a);
    } else {
        if (skip) hipLaunchKernelGGL((merge_bwd_kernel<192, 16, true>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((merge_bwd_kernel<192, 16, false>), dim3(grid), dim3(256), 0, stream, a);
    }
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_unmerge_skip_supported(int C, int B, int H, int W) {
    return grid_ok(B, H, W) && (C == 192 || C == 384) && (B * H * W) % 16 == 0;
}

extern "C" int tulip_unmerge_skip_fwd(const tulip_unmerge_skip_desc* d, hipStream_t stream) {
    if (!d || !tulip_unmerge_skip_supported(d->C, d->B, d->H, d->W)) return TULIP_ERR_ARG;
    if (!d->x_bf16 || !d->w_expand_packed || !d->b_expand || !d->cat || !d->w_skip_packed || !d->b_skip || !d->out) return TULIP_ERR_ARG;
    const int M = d->B * d->H * d->W;
    UnmergeSkipArgs a{(const bf16_t*)d->x_bf16, (const bf16_t*)d->w_expand_packed, d->b_expand, (bf16_t*)d->cat,
                      (const bf16_t*)d->w_skip_packed, d->b_skip, d->out, d->B, d->H, d->W, M};
    if (d->C == 192) hipLaunchKernelGGL((unmerge_skip_fwd_kernel<192>), dim3(M / 16), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((unmerge_skip_fwd_kernel<384>), dim3(M / 16), dim3(256), 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_skip_unmerge_bwd(const tulip_skip_unmerge_bwd_desc* d, hipStream_t stream) {
    if (!d || !tulip_unmerge_skip_supported(d->C, d->B, d->H, d->W)) return TULIP_ERR_ARG;
    if (!d->dy_skip || !d->w_skip_t_packed || !d->dz || !d->w_expand_t_packed || !d->dx) return TULIP_ERR_ARG;
    if (d->dx_bf16 && d->cast_rowscale && d->cast_rows_per_sample <= 0) return TULIP_ERR_ARG;
    const int M = d->B * d->H * d->W;
    SkipUnmergeBwdArgs a{(const bf16_t*)d->dy_skip, (const bf16_t*)d->w_skip_t_packed, (bf16_t*)d->dz,
                         (const bf16_t*)d->w_expand_t_packed, d->dx, (bf16_t*)d->dx_bf16, d->cast_rowscale,
                         d->cast_rows_per_sample, d->B, d->H, d->W, M};
    if (d->C == 192) hipLaunchKernelGGL((skip_unmerge_bwd_kernel<192>), dim3(M / 16), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((skip_unmerge_bwd_kernel<384>), dim3(M / 16), dim3(256), 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
