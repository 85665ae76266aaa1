// One launch for a whole Swin block at the wider stages (C = 192 and 384: stages 1 and 2 of every TULIP model; 32-wide
// heads, window 2x8, MLP C -> 4C -> C; tulip.py:338-352 with :289-323 and :194-200 inside), forward and backward.
//
// At these widths a block's weights (12 C^2 bf16 = 0.9 / 3.5 MB) no longer fit in LDS and one window per wave (the
// C = 96 design of swin96.hip) is far too serial: stage 1 of a batch of 8 has 512 windows for 1024 SIMDs.  Here a
// workgroup owns G neighbouring windows (T = 16 G tokens) and has one wave per HEAD (6 / 12 waves); every GEMM of the
// block is split along its OUTPUT channels across the waves:
//   * weights never touch LDS: each wave streams exactly its own rows from L2 straight into MFMA operand registers,
//     from a FRAGMENT-MAJOR bf16 copy of the weights in which the A operand of one MFMA (W . X^T form) is one fully
//     contiguous 1-KiB wave load (a ring of 12 loads in flight per wave); a fragment is reused for the G token tiles;
//   * activations go through LDS between the GEMMs ([32-k tile][token][64 B], 16-B XOR swizzle: conflict-free
//     ds_read_b128 fragments): LayerNorm output -> qkv (wave = head: q, k, v of its head stay in registers for the
//     attention core) -> attention output -> proj (+ residual; LayerNorm statistics combined across the waves with the
//     parallel-variance formula, one 8-byte exchange per token and wave) -> fc1 -> GELU -> fc2 (+ residual).
// Everything the backward needs is written exactly as the separate kernels write it.
//
// The backward mirrors it: fc2' -> GELU' -> fc1' -> norm2' -> proj' -> attention' -> qkv' -> norm1' with the same
// split.  Its data-gradient GEMMs contract over the output channels of each Linear, i.e. they stream W^T rows: the
// caller passes fragment-major copies of the four TRANSPOSED weights (tulip_pack_bf16_multi writes both kinds of copy
// once per optimizer step).
// Per-channel parameter partial sums (LayerNorm affine, relative-position bias) are disjoint between waves, so each
// wave writes its slice of the workgroup's partial row directly.
#include <type_traits>
// the activations saved for the backward / the operands left for the weight-gradient launches leave as non-temporal stores
// (common.h, store_late): same-box A/B of the step 2.021 vs 2.035 ms at batch 8, 9.07 vs 9.13 at batch 64 (four pairs each;
// isolated launches -1..-3 us).  The C = 96 kernels measured the other way (isolated forward 32.0 -> 35.4 us) and keep plain stores.
#ifndef TULIP_STORE_LATE_W
#define TULIP_STORE_LATE_W 1
#endif
#define TULIP_STORE_LATE TULIP_STORE_LATE_W
#ifdef TULIP_LOAD_SAVED_NT_W
#define TULIP_LOAD_SAVED_NT TULIP_LOAD_SAVED_NT_W
#endif
#include "common.h"
#include "tulip_hip.h"

#include "swin_stream.h"
// arrival tickets of the two-workgroups-per-window forms: the partial sums are published with write-through (sc1) stores that
// every wave drains (s_waitcnt vmcnt(0) + barrier) before the ticket is drawn, and the last arriver reads its partner's with sc1
// loads behind the ticket -- exactly the lines that cross the XCD boundary, nothing else.  Drawing the ticket acq_rel instead
// (-DTULIP_TICKET_ORDER=__ATOMIC_ACQ_REL: correct by the HIP memory model alone) makes the compiler write back and invalidate the
// whole L2 around it, with the block's saved activations dirty in it: +11 us per batch-8 step over four launches (same-box A/B,
// three pairs, profiles/r5_ab_ticket.txt) -- measured, not taken.  A sequence cut short re-zeroes the tickets (Plan.reset_exchange).
#ifndef TULIP_TICKET_ORDER
#define TULIP_TICKET_ORDER __ATOMIC_RELAXED
#endif

namespace {


// Cold caches.  In the training step a block's weights were last read a whole step ago (AdamW streams 0.8 GB through the
// Infinity Cache in between), and every workgroup of an XCD walks the SAME 24 C^2 bytes in the same order with 6-12 KiB in
// flight per wave: the stream then runs at (bytes in flight) / (miss latency) -- measured (tools/cold_probe.py) 73 us
// for the C = 384 forward against 48 us with the weights in L2.  So when the whole grid is resident at once (<= 256
// workgroups), the workgroups of an XCD (blockIdx % 8) first split the block's weights between them: every wave
// touches a few KiB chunks nobody else touches, in order of use, and drops the data; the chunks land in the XCD's L2
// within a few miss latencies and the streams behind them hit.
// ON = 1: every workgroup takes part (grids of at most 256 workgroups); ON = 2: larger grids -- the first 256 workgroups
// (the ones that find the caches cold) do it for the ones behind them.
template <int ON>
struct WeightWarm {
    unsigned char* scratch;
    int nslots, slot, ws, loff;
    bool on;
    // The touches are LDS-destination loads (global_load_lds_dwordx4 through the compiler's builtin): the data is dropped
    // into a 1-KiB LDS scratch that nothing ever reads (every wave of the workgroup writes the same KiB), so no VGPR is the
    // target of an in-flight load -- nothing the register allocator could move or reuse underneath it -- and the compiler
    // counts them (vmcnt) like any other memory operation.  Returns are in order: the first wait for a younger load
    // also waits for them, one exposed miss latency at the head of the kernel.
    __device__ __forceinline__ void init(int wid, int lane, unsigned char* lds_scratch) {
        if constexpr (ON) {
            nslots = (ON == 2 ? 256 : (int)gridDim.x) >> 3; slot = blockIdx.x >> 3;
            on = ON == 1 || blockIdx.x < 256;
            ws = __builtin_amdgcn_readfirstlane(wid); loff = lane * 16;
            scratch = lds_scratch;
        }
    }
    // one fragment-major matrix of KIB KiB; sized for >= 192 waves per XCD (batch 8), fewer waves leave a tail cold
    template <int NWV, int KIB>
    __device__ __forceinline__ void touch(const bf16_t* w) {
        if constexpr (ON) {
            if (on) {                                       // workgroup-uniform
#pragma unroll
                for (int r = 0; r < (KIB + 191) / 192; ++r) {   // chunk and address are wave-uniform: scalar arithmetic
                    int c = (r * nslots + slot) * NWV + ws;
                    c = c < KIB ? c : KIB - 1;
                    const unsigned char* p = (const unsigned char*)(w + (size_t)c * 512) + loff;
                    warm_touch16(p, scratch);
                }
            }
        }
    }
};

struct SwinWArgs {
    const float* xin; float* x1; float* xout;
    bf16_t *xn1, *qkv, *o, *xn2, *h, *g;
    float *mean1, *rstd1, *mean2, *rstd2;
    const bf16_t *wqkv, *wproj, *w1, *w2;
    const float *bqkv, *bproj, *b1, *b2, *g1, *be1, *g2, *be2;
    const float* bias_table; const int* rel_index;
    const float *ds0, *ds1;               // DropPath multipliers per sample (attention / MLP branch) or nullptr
    bf16_t* out_bf16;                     // optional bf16 copy of the block output (operand of a PatchUnmerging GEMM)
    unsigned long long* prof;             // optional: s_memtime stamps [workgroup][wave][16] at the phase boundaries
    int B, H, W, sh, sw, masked;
    float eps, scale;
    // split form (two workgroups per window, MODE bit 4): fc2 partial sums [window][half][thread][2] f32x4, arrival tickets [window]
    float* xws; unsigned* tick; unsigned xws_bytes;
};
#if TULIP_DEV_VARIANTS
#define TULIP_STAMP(k) do { if (a.prof && lane == 0) a.prof[((size_t)blockIdx.x * NWV + wid) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TULIP_STAMP(k) ((void)0)        // (the product library's kernels carry no stamp branches: tulip_swinw_block_fwd_profiled answers TULIP_ERR_NOT_BUILT)
#endif


template <int C, int G>
struct Geo {
    static constexpr int NH = C / 32, NWV = NH, NT = NWV * 64, T = 16 * G, KS = C / 32, HID = 4 * C;
    static constexpr int XN_BYTES = T * C * 2;                 // LayerNorm output (norm1, then norm2)
    static constexpr int BIG_BYTES = T * HID * 2;              // attention output | V tiles, later gelu(fc1)
    static constexpr int OFF_XN = 0, OFF_BIG = XN_BYTES, OFF_V = OFF_BIG + XN_BYTES;
    static constexpr int OFF_STAT = OFF_BIG + BIG_BYTES;
    static constexpr int OFF_WARM = OFF_STAT + NWV * T * 8;     // 1-KiB sink of the L2 warm-up loads (WeightWarm)
    static constexpr int SMEM = OFF_WARM + 1024;
    static_assert(OFF_V + NWV * 1024 <= OFF_STAT, "V tiles must fit beside the attention output");
    static_assert(SMEM <= 163840, "LDS");
};

// natural-order token of tile slot tt = 16 g + t (cyclic shift + window partition are address arithmetic, tulip.py:289-297)
struct TokMap {
    int b, wy, wx0, H, W, sh, sw;
    __device__ __forceinline__ void coords(int tt, int& hs, int& ws) const {
        const int t = tt & 15;
        hs = wy * 2 + (t >> 3);
        ws = (wx0 + (tt >> 4)) * 8 + (t & 7);
    }
    __device__ __forceinline__ size_t row(int tt) const {
        int hs, ws;
        coords(tt, hs, ws);
        int hh = hs + sh; if (hh >= H) hh -= H;
        int ww = ws + sw; if (ww >= W) ww -= W;
        return ((size_t)b * H + hh) * W + ww;
    }
    __device__ __forceinline__ int label(int tt) const {
        int hs, ws;
        coords(tt, hs, ws);
        return 3 * region(hs, H, 2, sh) + region(ws, W, 8, sw);
    }
};
template <int G>
__device__ __forceinline__ TokMap make_map(int B, int H, int W, int sh, int sw, int blk = (int)blockIdx.x) {
    const int nWx = W >> 3, nWy = H >> 1, gpr = nWx / G;
    TokMap m;
    m.b = blk / (nWy * gpr);
    blk -= m.b * nWy * gpr;
    m.wy = blk / gpr;
    m.wx0 = (blk - m.wy * gpr) * G;
    m.H = H; m.W = W; m.sh = sh; m.sw = sw;
    return m;
}

// MODE: bits 0-1 = the L2 warm-up (0 none, 1 every workgroup, 2 the first 256); bit 2 = the inference form (eval / MC-dropout
// forward: the activations a backward would need are not written); bit 3 = the fc1_pre buffer receives gelu'(h), not h
template <int C, int G, int MODE>
__global__ __launch_bounds__((Geo<C, G>::NT)) void swinw_fwd_kernel(const SwinWArgs a) {
    constexpr int WARM = MODE & 3;
    constexpr bool SAVE = !(MODE & 4);
    constexpr bool HGRAD = (MODE & 8) != 0;     // round 4 (TULIP_BLOCK_FC1_GRAD): the fc1_pre buffer receives bf16(gelu'(h)) instead of h
    // SPLIT (round 4, C = 384 with one window per workgroup: 128 workgroups on 256 CUs at batch 8, each streaming the block's
    // 3.5 MB of weights at the per-CU L1 rate): TWO workgroups per window.  Both run the attention half (norm1 .. norm2: a third
    // of the stream) redundantly -- workgroup 0 of the pair writes what it saves -- and each takes HALF of the hidden channels of
    // the MLP: fc1 / GELU for 768 of them, fc2 over that half of its k range.  The two fc2 partial sums meet through memory:
    // each workgroup publishes its 16 x 384 fp32 partial (write-through stores, drained; guide recipe R1 in its ticket form),
    // draws a ticket, and the LAST arriver adds its partner's partial to its own, applies bias / DropPath / residual and
    // writes the block output; the first arriver just leaves.  No workgroup ever waits for another.  a + b == b + a: the
    // output does not depend on the order of arrival.
    constexpr bool SPLIT = (MODE & 16) != 0;
    static_assert(!SPLIT || (C == 384 && G == 1 && (HGRAD || !SAVE)), "split form: C = 384, one window per workgroup pair");
    using Z = Geo<C, G>;
    constexpr int T = Z::T, KS = Z::KS, NWV = Z::NWV, HID = Z::HID, NH = Z::NH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::SMEM];
    unsigned char* const XN = smem + Z::OFF_XN;
    unsigned char* const XO = smem + Z::OFF_BIG;
    unsigned char* const GB = smem + Z::OFF_BIG;
    float2* const STAT = (float2*)(smem + Z::OFF_STAT);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    unsigned char* const ldsV = smem + Z::OFF_V + wid * 1024;
    const int half = SPLIT ? (int)(blockIdx.x & 1) : 0, wblk = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const bool sv = SAVE && (!SPLIT || half == 0);          // the tensors both workgroups of a pair hold: saved by the first
    const TokMap tm = make_map<G>(a.B, a.H, a.W, a.sh, a.sw, wblk);
    const float s0 = a.ds0 ? a.ds0[tm.b] : 1.0f, s1v = a.ds1 ? a.ds1[tm.b] : 1.0f;

    // loads in flight per wave: 12 with 6 waves per CU, 6-8 with 12 waves or 4 windows (register budget 168 / 256)
    constexpr int D = (C == 192 && G == 2) ? 2 : 1;
    TULIP_STAMP(0);
    // the qkv weight stream starts before anything else: it depends on nothing
    WStream<6, KS, D> wq;
#pragma unroll
    for (int i = 0; i < 6; ++i) wq.wt[i] = wtile_ptr(a.wqkv, (i >> 1) * (C / 16) + 2 * wid + (i & 1), C, lane);
    wq.start();
    WeightWarm<WARM> warm;                          // the block's weights into this XCD's L2, in order of use
    warm.init(wid, lane, smem + Z::OFF_WARM);
    warm.template touch<NWV, 6 * C * C / 1024>(a.wqkv);
    warm.template touch<NWV, 2 * C * C / 1024>(a.wproj);
    warm.template touch<NWV, 8 * C * C / 1024>(a.w1);
    warm.template touch<NWV, 8 * C * C / 1024>(a.w2);
    // ---- norm1 (tulip.py:340): 16 lanes per token, 4 tokens per wave pass; every pass's loads are issued first
    {
        constexpr int NP = (T + NWV * 4 - 1) / (NWV * 4);
        float4 xv[NP][C / 64];
        size_t rowp[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = wid * 4 + gq + p * NWV * 4;
            rowp[p] = tm.row(tt < T ? tt : T - 1);
#pragma unroll
            for (int j = 0; j < C / 64; ++j) xv[p][j] = *(const float4*)(a.xin + rowp[p] * C + 4 * t + 64 * j);
        }
        float4 ga[C / 64], be[C / 64];
#pragma unroll
        for (int j = 0; j < C / 64; ++j) { ga[j] = *(const float4*)(a.g1 + 4 * t + 64 * j); be[j] = *(const float4*)(a.be1 + 4 * t + 64 * j); }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = wid * 4 + gq + p * NWV * 4;
            if (tt < T) {
                const size_t row = rowp[p];
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < C / 64; ++j) sm += (xv[p][j].x + xv[p][j].y) + (xv[p][j].z + xv[p][j].w);
                const float mu = group_sum<16>(sm) * (1.0f / C);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < C / 64; ++j) {
                    const float d0 = xv[p][j].x - mu, d1 = xv[p][j].y - mu, d2 = xv[p][j].z - mu, d3 = xv[p][j].w - mu;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                const float rs = rsqrtf(group_sum<16>(q) * (1.0f / C) + a.eps);
                if (sv && t == 0) { a.mean1[row] = mu; a.rstd1[row] = rs; }
#pragma unroll
                for (int j = 0; j < C / 64; ++j) {
                    const int c = 4 * t + 64 * j;
                    const bf16x4 pk = pack4((xv[p][j].x - mu) * rs * ga[j].x + be[j].x, (xv[p][j].y - mu) * rs * ga[j].y + be[j].y,
                                            (xv[p][j].z - mu) * rs * ga[j].z + be[j].z, (xv[p][j].w - mu) * rs * ga[j].w + be[j].w);
                    if (sv) store_late((bf16x4*)(a.xn1 + row * C + c), pk);
                    put4<T>(XN, tt, c, pk);
                }
            }
        }
    }
    // this lane's tokens in the MFMA phases: token t of window g
    size_t rows[G];
    int lab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { rows[g] = tm.row(16 * g + t); lab[g] = tm.label(16 * g + t); }
    // relative-position bias of this wave's head for (query t, keys 4gq..4gq+3) (tulip.py:304-308), qkv biases
    float rpb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rpb[r] = a.bias_table[a.rel_index[t * 16 + gq * 4 + r] * NH + wid];
    f32x4 bq[D == 2 ? 6 : 1];                       // hoisted only where the register budget allows (6 waves per CU)
    if constexpr (D == 2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) bq[i] = ld4(a.bqkv + (i >> 1) * C + 32 * wid + 16 * (i & 1) + 4 * gq);
    }
    TULIP_STAMP(1);
    __syncthreads();
    TULIP_STAMP(2);

    // ---- qkv Linear (tulip.py:298), wave = head: q, k, v channels 32 wid .. +31 of each section
    bf16x4 qkvp[6][G];
    WStream<2, KS, 3 * D> wp;                       // proj weights: this wave's 32 output channels
#pragma unroll
    for (int i = 0; i < 2; ++i) wp.wt[i] = wtile_ptr(a.wproj, 2 * wid + i, C, lane);
    {
        f32x4 acc[6][G];
        zero(acc);
        wq.template run<G, T>(acc, XN, t, gq);
        TULIP_STAMP(3);
        wp.start();                                 // ahead of the qkv stores
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int n = (i >> 1) * C + 32 * wid + 16 * (i & 1) + 4 * gq;
            f32x4 bqi;
            if constexpr (D == 2) bqi = bq[i]; else bqi = ld4(a.bqkv + n);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                qkvp[i][g] = pack4(acc[i][g][0] + bqi[0], acc[i][g][1] + bqi[1], acc[i][g][2] + bqi[2], acc[i][g][3] + bqi[3]);
                if (sv && (i & 1))        // two adjacent tiles: one 16-byte store per lane (common.h)
                    store_bf16_tile_pair<true>(a.qkv + rows[g] * (3 * C) + n - 16 - 4 * gq, qkvp[i - 1][g], qkvp[i][g], gq);
            }
        }
    }
    // residual input and proj bias of this wave's channels: in flight during the attention core
    f32x4 x1v[2][G], bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bp[i] = ld4(a.bproj + 32 * wid + 16 * i + 4 * gq);
#pragma unroll
        for (int g = 0; g < G; ++g) x1v[i][g] = ld4(a.xin + rows[g] * C + 32 * wid + 16 * i + 4 * gq);
    }

    TULIP_STAMP(4);
    // ---- attention of this head, one window at a time (tulip.py:300-317); scores as K.Q^T: lane = query t, keys 4gq + r
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bf16x8 qf = cat8(qkvp[0][g], qkvp[1][g]);
        const bf16x8 kf = cat8(qkvp[2][g], qkvp[3][g]);
        *(bf16x4*)(ldsV + t * 64 + (4 * gq) * 2) = qkvp[4][g];             // V tile [token][d], d = 0..15
        *(bf16x4*)(ldsV + t * 64 + (16 + 4 * gq) * 2) = qkvp[5][g];        // d = 16..31
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
            put4<T>(XO, 16 * g + t, 32 * wid + 16 * dc + 4 * gq, opp[dc]);
        }
        if (sv) store_bf16_tile_pair<true>(a.o + rows[g] * C + 32 * wid, opp[0], opp[1], gq);
    }
    TULIP_STAMP(5);
    __syncthreads();
    TULIP_STAMP(6);

    // ---- proj Linear + DropPath + residual (tulip.py:318,344): this wave's 32 output channels; then norm2 (:347)
    WStream<4, KS, D + 1> w1a;                      // fc1, first 64 of this wave's 128 hidden channels
#pragma unroll
    for (int i = 0; i < 4; ++i) w1a.wt[i] = wtile_ptr(a.w1, SPLIT ? 48 * half + 4 * wid + i : 8 * wid + i, C, lane);
    f32x4 ga2[2], be2[2];
    {
        f32x4 acc[2][G];
        zero(acc);
        wp.template run<G, T>(acc, XO, t, gq);
        TULIP_STAMP(7);
        w1a.start();                                // ahead of the x1 / xn2 stores
#pragma unroll
        for (int i = 0; i < 2; ++i) { ga2[i] = ld4(a.g2 + 32 * wid + 16 * i + 4 * gq); be2[i] = ld4(a.be2 + 32 * wid + 16 * i + 4 * gq); }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c0 = 32 * wid + 16 * i + 4 * gq;
                x1v[i][g] = x1v[i][g] + s0 * (acc[i][g] + bp[i]);
                if (sv) store_late((float4*)(a.x1 + rows[g] * C + c0), make_float4(x1v[i][g][0], x1v[i][g][1], x1v[i][g][2], x1v[i][g][3]));
                sm += (x1v[i][g][0] + x1v[i][g][1]) + (x1v[i][g][2] + x1v[i][g][3]);
            }
            // statistics of this wave's 32 channels of token t: (mean, sum of squared deviations)
            sm = rows_sum(sm);
            const float mw = sm * (1.0f / 32);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = x1v[i][g][r] - mw; q += d * d; }
            q = rows_sum(q);
            if (gq == 0) STAT[wid * T + 16 * g + t] = make_float2(mw, q);
        }
    }
    TULIP_STAMP(8);
    __syncthreads();
    TULIP_STAMP(9);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // combine the NWV equal-sized groups: mean of means; M2 = sum M2_w + 32 sum (mean_w - mean)^2
        float2 st[NWV];
        float ms = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { st[w] = STAT[w * T + 16 * g + t]; ms += st[w].x; }
        const float mu = ms * (1.0f / NWV);
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const float d = st[w].x - mu; m2 += st[w].y + 32.0f * d * d; }
        const float rs = rsqrtf(m2 * (1.0f / C) + a.eps);
        if (sv && wid == 0 && gq == 0) { a.mean2[rows[g]] = mu; a.rstd2[rows[g]] = rs; }
        bf16x4 pk2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = 32 * wid + 16 * i + 4 * gq;
            pk2[i] = pack4((x1v[i][g][0] - mu) * rs * ga2[i][0] + be2[i][0], (x1v[i][g][1] - mu) * rs * ga2[i][1] + be2[i][1],
                           (x1v[i][g][2] - mu) * rs * ga2[i][2] + be2[i][2], (x1v[i][g][3] - mu) * rs * ga2[i][3] + be2[i][3]);
            put4<T>(XN, 16 * g + t, c0, pk2[i]);
        }
        if (sv) store_bf16_tile_pair<true>(a.xn2 + rows[g] * C + 32 * wid, pk2[0], pk2[1], gq);
    }
    f32x4 b1v[D == 2 ? 2 : 1][4];
    if constexpr (D == 2) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) b1v[ch][i] = ld4(a.b1 + 128 * wid + 64 * ch + 16 * i + 4 * gq);
    }
    TULIP_STAMP(10);
    __syncthreads();
    TULIP_STAMP(11);

    // ---- fc1 + exact-erf GELU (tulip.py:195-196): this wave's 128 hidden channels, 64 at a time
    WStream<4, KS, D + 1> w1b;
#pragma unroll
    for (int i = 0; i < 4; ++i) w1b.wt[i] = wtile_ptr(a.w1, 8 * wid + 4 + i, C, lane);
    constexpr int KS2 = SPLIT ? 2 * KS : 4 * KS;    // k steps of fc2 in this workgroup (split form: its half of the hidden channels)
    WStream<2, KS2, 3 * D> w2s;
#pragma unroll
    for (int i = 0; i < 2; ++i) w2s.wt[i] = wtile_ptr(a.w2, 2 * wid + i, HID, lane) + (SPLIT ? (size_t)half * KS2 * 512 : 0);
    auto fc1_out = [&](const f32x4 (&acc)[4][G], int ch) {
        bf16x4 hprev[G], gprev[G];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // nl: the channel's k index of fc2 inside this workgroup; n: the hidden channel
            const int nl = SPLIT ? 64 * wid + 16 * i + 4 * gq : 128 * wid + 64 * ch + 16 * i + 4 * gq;
            const int n = SPLIT ? 768 * half + nl : nl;
            f32x4 bb;
            if constexpr (D == 2) bb = b1v[ch][i]; else bb = ld4(a.b1 + n);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                bf16x4 hp = pack4(acc[i][g][0] + bb[0], acc[i][g][1] + bb[1], acc[i][g][2] + bb[2], acc[i][g][3] + bb[3]);
                const f32x2 h01 = {bf2f((bf16_t)hp[0]), bf2f((bf16_t)hp[1])}, h23 = {bf2f((bf16_t)hp[2]), bf2f((bf16_t)hp[3])};
                f32x2 g01, g23;                                                                        // GELU of the (rounded) h
                if constexpr (HGRAD && SAVE) {      // ... and its derivative: all the backward wants from h
                    f32x2 d01, d23;
                    gelu_exact_and_grad2(h01, g01, d01);
                    gelu_exact_and_grad2(h23, g23, d23);
                    hp = pack4(d01.x, d01.y, d23.x, d23.y);
                } else {
                    g01 = gelu_exact2(h01); g23 = gelu_exact2(h23);
                }
                const bf16x4 gp = pack4(g01.x, g01.y, g23.x, g23.y);
                put4<T>(GB, 16 * g + t, nl, gp);
                if (i & 1) {    // two adjacent tiles: one 16-byte store per lane (common.h)
                    // (uniform part spelled out: with the lane's 4 gq added and subtracted again the C = 192 kernel spilled 16 bytes)
                    const size_t off = rows[g] * HID + (SPLIT ? 768 * half + 64 * wid : 128 * wid + 64 * ch) + 16 * (i - 1);
                    if constexpr (SAVE) {
                        store_bf16_tile_pair<true>(a.h + off, hprev[g], hp, gq);
                        store_bf16_tile_pair<true>(a.g + off, gprev[g], gp, gq);
                    }
                } else {
                    hprev[g] = hp; gprev[g] = gp;
                }
            }
        }
    };
    {
        f32x4 acc[4][G];
        zero(acc);
        w1a.template run<G, T>(acc, XN, t, gq);
        if constexpr (SPLIT) {
            w2s.start();
            fc1_out(acc, 0);
        } else {
            w1b.start();
            fc1_out(acc, 0);
            zero(acc);
            w1b.template run<G, T>(acc, XN, t, gq);
            w2s.start();
            fc1_out(acc, 1);
        }
    }
    f32x4 b2v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b2v[i] = ld4(a.b2 + 32 * wid + 16 * i + 4 * gq);
    TULIP_STAMP(12);
    __syncthreads();
    TULIP_STAMP(13);

    // ---- fc2 + DropPath + residual (tulip.py:198,351)
    {
        f32x4 acc[2][G];
        zero(acc);
        w2s.template run<G, T>(acc, GB, t, gq);
        TULIP_STAMP(14);
        if constexpr (SPLIT) {
            typedef unsigned u32x4_x __attribute__((ext_vector_type(4)));
            // (descriptor from wave-uniform kernel arguments; every per-lane part in the offset)
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xws, 0, (int)a.xws_bytes, 0x00020000);
            const unsigned mine = (unsigned)(((wblk * 2 + half) * 2) * Z::NT + tid) * 16u;
            const unsigned other = (unsigned)(((wblk * 2 + (half ^ 1)) * 2) * Z::NT + tid) * 16u;
#pragma unroll
            for (int i = 0; i < 2; ++i)     // write-through (sc1) stores: no release fence, no L2 write-back
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_x, acc[i][0]), rsrc, mine + i * (Z::NT * 16), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains
            __syncthreads();
            unsigned* flag = (unsigned*)STAT;                          // (the statistics exchange is long over)
            if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tick + wblk, 1u, TULIP_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (flag[0] == 0u) return;                                 // first of the pair: the partner finishes the block
            if (tid == 0) __hip_atomic_store(a.tick + wblk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // for the next launch
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4_x o = __builtin_amdgcn_raw_buffer_load_b128(rsrc, other + i * (Z::NT * 16), 0, 16);
                acc[i][0] += __builtin_bit_cast(f32x4, o);
            }
        }
        bf16x4 ob[2][G];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = 32 * wid + 16 * i + 4 * gq;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 o = x1v[i][g] + s1v * (acc[i][g] + b2v[i]);
                *(float4*)(a.xout + rows[g] * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
                ob[i][g] = pack4(o[0], o[1], o[2], o[3]);
            }
        }
        if (a.out_bf16) {
#pragma unroll
            for (int g = 0; g < G; ++g) store_bf16_tile_pair(a.out_bf16 + rows[g] * C + 32 * wid, ob[0][g], ob[1][g], gq);
        }
    }
    TULIP_STAMP(15);
}

// two workgroups per window (swinw_fwd_kernel, SPLIT): the training form with gelu'(h) handed over, or the inference form
int launch_fwd_split(const SwinWArgs& a, hipStream_t stream) {
    const int windows = a.B * (a.H / 2) * (a.W / 8);
    const int warm = (a.prof || (a.masked & TULIP_BLOCK_NO_WARM)) ? 0 : 1;
    const dim3 grid(2 * windows), block(Geo<384, 1>::NT);
    if (!a.qkv) {
        if (warm) hipLaunchKernelGGL((swinw_fwd_kernel<384, 1, 21>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_fwd_kernel<384, 1, 20>), grid, block, 0, stream, a);
    } else if (warm) hipLaunchKernelGGL((swinw_fwd_kernel<384, 1, 25>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((swinw_fwd_kernel<384, 1, 24>), grid, block, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
template <int C, int G>
int launch_fwd(const SwinWArgs& a, hipStream_t stream) {
    const int blocks = a.B * (a.H / 2) * (a.W / (8 * G));
    // the workgroups that are resident first warm their XCD's L2 with the block's weights (WeightWarm)
    const int warm = (a.prof || (a.masked & TULIP_BLOCK_NO_WARM) || blocks < 8) ? 0 : (blocks <= 256 ? 1 : 2);
    const dim3 grid(blocks), block(Geo<C, G>::NT);
    if (a.qkv && (a.masked & TULIP_BLOCK_FC1_GRAD)) {       // training form, gelu'(h) handed to the backward
        if (warm == 1) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 9>), grid, block, 0, stream, a);
        else if (warm == 2) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 10>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 8>), grid, block, 0, stream, a);
    } else if (a.qkv) {                                     // training form that saves h itself: development build only
#if TULIP_DEV_VARIANTS
        if (warm == 1) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 1>), grid, block, 0, stream, a);
        else if (warm == 2) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 2>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 0>), grid, block, 0, stream, a);
#else
        return TULIP_ERR_NOT_BUILT;
#endif
    } else {                                                // inference form: no saved activations
        if (warm == 1) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 5>), grid, block, 0, stream, a);
        else if (warm == 2) hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 6>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_fwd_kernel<C, G, 4>), grid, block, 0, stream, a);
    }
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}


// =====================================================================================================================
// backward
struct SwinWBwdArgs {
    float* dx;                                   // in: d(block output); out: d(block input)
    const float *xin, *x1;
    const bf16_t *qkv, *h;
    const float *mean1, *rstd1, *mean2, *rstd2;
    const bf16_t *wqkvt, *wprojt, *w1t, *w2t;    // fragment-major TRANSPOSED weights: of [C][3C], [C][C], [C][4C], [4C][C]
    const float *g1, *g2;
    const float* bias_table; const int* rel_index;
    const float *ds0, *ds1;
    bf16_t *dyb_m, *dh, *dyb_a, *dqkv;           // bf16 operands of the fc2 / fc1 / proj / qkv weight gradients
    bf16_t* dx_bf16; const float* dx_scale;      // optional bf16(dx * per-sample scale) for the consumer of dx
    float *lnpart1, *lnpart2, *biaspart;         // [workgroups][2C], [workgroups][2C], [workgroups][NH*256]
    int B, H, W, sh, sw, masked;
    float scale;
    float* xws; unsigned* tick; unsigned xws_bytes;     // split form (BMODE bit 3): as SwinWArgs
};

template <int C, int G>
struct GeoB {
    static constexpr int NH = C / 32, NWV = NH, NT = NWV * 64, T = 16 * G, KS = C / 32, HID = 4 * C;
    static constexpr int DY_BYTES = T * C * 2;                 // bf16(dy s_mlp), then bf16(d(x1) s_attn)
    static constexpr int BIG_BYTES = T * HID * 2;              // d(fc1 pre-activation), later d(qkv) (3/4 of it)
    static constexpr int OFF_DY = 0, OFF_BIG = DY_BYTES, OFF_ATT = OFF_BIG + BIG_BYTES;   // NWV x (Q | K | dO) 1-KiB tiles
    static constexpr int OFF_STAT = OFF_ATT + NWV * 3072;
    static constexpr int OFF_WARM = OFF_STAT + NWV * T * 8;     // 1-KiB sink of the L2 warm-up loads (WeightWarm)
    static constexpr int SMEM = OFF_WARM + 1024;
    static_assert(SMEM <= 163840, "LDS");
};

// LayerNorm backward of this wave's 32-channel slice (2 tiles x 4 channels per lane, G token tiles; x, the row
// statistics and gamma were fetched while the GEMM in front was running).  Phase 1: the
// affine-gradient sums of the slice (over the workgroup's T tokens) go straight to the partial row, d <- d * gamma,
// and the per-token partial sums (sum d, sum d*xhat over the 32 channels) to STAT.  Phase 2 (after the barrier):
// d <- rstd * (d - m1 - xhat * m2) with the sums over all NWV waves.
template <int C, int G, int T>
__device__ __forceinline__ void ln_bwd_part1(f32x4 (&d)[2][G], f32x4 (&xh)[2][G], const f32x4 (&xv)[2][G],
                                             const float (&mu)[G], const float (&rs)[G], const f32x4 (&gam)[2],
                                             float* __restrict__ part, float2* STAT, int wid, int t, int gq) {
    f32x4 pg[2], pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pg[i] = pb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[i][g][r] = (xv[i][g][r] - mu[g]) * rs[g];
                pg[i][r] += d[i][g][r] * xh[i][g][r];
                pb[i][r] += d[i][g][r];
                d[i][g][r] *= gam[i][r];
                s1 += d[i][g][r];
                s2 += d[i][g][r] * xh[i][g][r];
            }
        }
        s1 = rows_sum(s1);
        s2 = rows_sum(s2);
        if (gq == 0) STAT[wid * T + 16 * g + t] = make_float2(s1, s2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 sg, sb;
        sg.x = group_sum<16>(pg[i][0]); sg.y = group_sum<16>(pg[i][1]); sg.z = group_sum<16>(pg[i][2]); sg.w = group_sum<16>(pg[i][3]);
        sb.x = group_sum<16>(pb[i][0]); sb.y = group_sum<16>(pb[i][1]); sb.z = group_sum<16>(pb[i][2]); sb.w = group_sum<16>(pb[i][3]);
        if (t == 0) {
            const int c0 = 32 * wid + 16 * i + 4 * gq;
            *(float4*)(part + c0) = sg;
            *(float4*)(part + C + c0) = sb;
        }
    }
}
template <int C, int G, int T, int NWV>
__device__ __forceinline__ void ln_bwd_part2(f32x4 (&d)[2][G], const f32x4 (&xh)[2][G], const float (&rs)[G],
                                             const float2* STAT, int t) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const float2 v = STAT[w * T + 16 * g + t]; s1 += v.x; s2 += v.y; }
        const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[i][g][r] = rs[g] * (d[i][g][r] - m1 - xh[i][g][r] * m2);
    }
}

// BMODE: bits 0-1 = the L2 warm-up as in the forward; bit 2 (round 4, TULIP_BLOCK_FC1_GRAD) = the fc1_pre buffer holds
// bf16(gelu'(h)), written by the forward's MODE bit 3, and the MLP half multiplies with it instead of evaluating erf / exp
template <int C, int G, int BMODE>
__global__ __launch_bounds__((GeoB<C, G>::NT)) void swinw_bwd_kernel(const SwinWBwdArgs a) {
    constexpr int WARM = BMODE & 3;
    constexpr bool HGRAD = (BMODE & 4) != 0;
    // SPLIT (round 4; the forward's MODE bit 4): two workgroups per window.  The MLP half comes first here: each workgroup takes
    // half of the hidden channels (fc2', GELU', and its half of fc1's contraction), publishes its 16 x 384 fp32 partial of
    // d(xn2) and draws a ticket; the LAST arriver adds its partner's partial and runs the rest of the block (norm2' ... norm1')
    // alone, the first one leaves.  Nobody waits; a + b == b + a.
    constexpr bool SPLIT = (BMODE & 8) != 0;
    static_assert(!SPLIT || (C == 384 && G == 1 && HGRAD), "split form: C = 384, one window per workgroup pair, gelu'(h) handed over");
    using Z = GeoB<C, G>;
    constexpr int T = Z::T, KS = Z::KS, NWV = Z::NWV, HID = Z::HID, NH = Z::NH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Z::SMEM];
    unsigned char* const DY = smem + Z::OFF_DY;
    unsigned char* const DH = smem + Z::OFF_BIG;
    unsigned char* const DQ = smem + Z::OFF_BIG;
    float2* const STAT = (float2*)(smem + Z::OFF_STAT);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    const int half = SPLIT ? (int)(blockIdx.x & 1) : 0, wblk = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const TokMap tm = make_map<G>(a.B, a.H, a.W, a.sh, a.sw, wblk);
    const float s0 = a.ds0 ? a.ds0[tm.b] : 1.0f, s1v = a.ds1 ? a.ds1[tm.b] : 1.0f;

    constexpr int D = (C == 192 && G == 2) ? 2 : 1;       // loads in flight per wave, as in the forward
    // the first weight stream (fc2^T, first 64 of this wave's 128 hidden channels) starts before anything else
    WStream<4, KS, D + 1> w2a;
#pragma unroll
    for (int i = 0; i < 4; ++i) w2a.wt[i] = wtile_ptr(a.w2t, SPLIT ? 48 * half + 4 * wid + i : 8 * wid + i, C, lane);
    w2a.start();
    WeightWarm<WARM> warm;                          // the four transposed weights into this XCD's L2, in order of use
    warm.init(wid, lane, smem + Z::OFF_WARM);
    warm.template touch<NWV, 8 * C * C / 1024>(a.w2t);
    warm.template touch<NWV, 8 * C * C / 1024>(a.w1t);
    warm.template touch<NWV, 2 * C * C / 1024>(a.wprojt);
    warm.template touch<NWV, 6 * C * C / 1024>(a.wqkvt);
    // ---- bf16(dy * s_mlp): operand of fc2's weight gradient and of the first data-gradient GEMM
    {
        constexpr int NP = (T + NWV * 4 - 1) / (NWV * 4);
        float4 v[NP][C / 64];
        size_t rowp[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = wid * 4 + gq + p * NWV * 4;
            rowp[p] = tm.row(tt < T ? tt : T - 1);
#pragma unroll
            for (int j = 0; j < C / 64; ++j) v[p][j] = *(const float4*)(a.dx + rowp[p] * C + 4 * t + 64 * j);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int tt = wid * 4 + gq + p * NWV * 4;
            if (tt < T) {
#pragma unroll
                for (int j = 0; j < C / 64; ++j) {
                    const int c = 4 * t + 64 * j;
                    const bf16x4 pk = pack4(v[p][j].x * s1v, v[p][j].y * s1v, v[p][j].z * s1v, v[p][j].w * s1v);
                    if (!SPLIT || half == 0) store_late((bf16x4*)(a.dyb_m + rowp[p] * C + c), pk);
                    put4<T>(DY, tt, c, pk);
                }
            }
        }
    }
    size_t rows[G];
    int lab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { rows[g] = tm.row(16 * g + t); lab[g] = tm.label(16 * g + t); }
    // relative-position bias seen from the query side (query t, key 4gq+r) and from the key side (query 4gq+r, key t)
    float bias_q[4], bias_k[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bias_q[r] = a.bias_table[a.rel_index[t * 16 + gq * 4 + r] * NH + wid];
        bias_k[r] = a.bias_table[a.rel_index[(gq * 4 + r) * 16 + t] * NH + wid];
    }
    __syncthreads();

    // ---- fc2' and GELU' (tulip.py:196-198 backwards): d(h) for this wave's 128 hidden channels = (dy . W2)[hid] * gelu'(h)
    WStream<4, KS, D + 1> w2b;
#pragma unroll
    for (int i = 0; i < 4; ++i) w2b.wt[i] = wtile_ptr(a.w2t, 8 * wid + 4 + i, C, lane);
    constexpr int KS1 = SPLIT ? 2 * KS : 4 * KS;    // k steps of fc1' in this workgroup (split form: its half of the hidden channels)
    WStream<2, KS1, 3 * D> w1s;                     // fc1^T: this wave's 32 channels of d(xn2)
#pragma unroll
    for (int i = 0; i < 2; ++i) w1s.wt[i] = wtile_ptr(a.w1t, 2 * wid + i, HID, lane) + (SPLIT ? (size_t)half * KS1 * 512 : 0);
    auto dh_out = [&](const f32x4 (&acc)[4][G], const bf16x4 (&hv)[4][G], int ch) {
        bf16x4 dprev[G];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // nl: the channel's k index of fc1' inside this workgroup; n: the hidden channel
            const int nl = SPLIT ? 64 * wid + 16 * i + 4 * gq : 128 * wid + 64 * ch + 16 * i + 4 * gq;
            const int n = SPLIT ? 768 * half + nl : nl;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x2 d01 = {bf2f((bf16_t)hv[i][g][0]), bf2f((bf16_t)hv[i][g][1])}, d23 = {bf2f((bf16_t)hv[i][g][2]), bf2f((bf16_t)hv[i][g][3])};
                if constexpr (!HGRAD) { d01 = gelu_exact_grad2(d01); d23 = gelu_exact_grad2(d23); }    // HGRAD: the forward left gelu'(h) there
                const bf16x4 dp = pack4(acc[i][g][0] * d01.x, acc[i][g][1] * d01.y, acc[i][g][2] * d23.x, acc[i][g][3] * d23.y);
                put4<T>(DH, 16 * g + t, nl, dp);
                if (i & 1) store_bf16_tile_pair<true>(a.dh + rows[g] * HID + (n - 16 - 4 * gq), dprev[g], dp, gq);
                else dprev[g] = dp;
            }
        }
    };
    auto load_h = [&](bf16x4 (&hv)[4][G], int ch) {          // the saved fc1 pre-activation
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g)
                hv[i][g] = ld_saved((const bf16x4*)(a.h + rows[g] * HID + (SPLIT ? 768 * half + 64 * wid : 128 * wid + 64 * ch) + 16 * i + 4 * gq));
    };
    {
        // (the activation loads stay BEHIND the GEMM they follow: vmcnt retires in order, so a strided HBM load issued
        // in front of the loop would sit in front of every weight refill the loop then waits for -- measured 35 -> 50 us)
        f32x4 acc[4][G];
        bf16x4 hv[4][G];
        zero(acc);
        w2a.template run<G, T>(acc, DY, t, gq);
        if constexpr (SPLIT) {
            w1s.start();
            load_h(hv, 0);
            dh_out(acc, hv, 0);
        } else {
            w2b.start();
            load_h(hv, 0);
            dh_out(acc, hv, 0);
            zero(acc);
            w2b.template run<G, T>(acc, DY, t, gq);
            w1s.start();
            load_h(hv, 1);
            dh_out(acc, hv, 1);
        }
    }
    f32x4 xv[2][G], gam[2], dyv[2][G];
    float mu[G], rs[G];
    __syncthreads();

    // ---- fc1' (tulip.py:195 backwards): d(xn2)[c] for this wave's 32 channels, then norm2' and the residual
    WStream<2, KS, 3 * D> wps;                      // proj^T: dO of this wave's head
#pragma unroll
    for (int i = 0; i < 2; ++i) wps.wt[i] = wtile_ptr(a.wprojt, 2 * wid + i, C, lane);
    f32x4 dx1[2][G];
    {
        f32x4 xh[2][G];
        zero(dx1);
        w1s.template run<G, T>(dx1, DH, t, gq);
        if constexpr (SPLIT) {
            typedef unsigned u32x4_x __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xws, 0, (int)a.xws_bytes, 0x00020000);
            const unsigned mine = (unsigned)(((wblk * 2 + half) * 2) * Z::NT + tid) * 16u;
            const unsigned other = (unsigned)(((wblk * 2 + (half ^ 1)) * 2) * Z::NT + tid) * 16u;
#pragma unroll
            for (int i = 0; i < 2; ++i)     // write-through (sc1) stores, drained by every wave (guide recipe R1, ticket form)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_x, dx1[i][0]), rsrc, mine + i * (Z::NT * 16), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = (unsigned*)STAT;
            if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tick + wblk, 1u, TULIP_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (flag[0] == 0u) return;                                 // first of the pair: the partner runs the rest of the block
            if (tid == 0) __hip_atomic_store(a.tick + wblk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4_x o = __builtin_amdgcn_raw_buffer_load_b128(rsrc, other + i * (Z::NT * 16), 0, 16);
                dx1[i][0] += __builtin_bit_cast(f32x4, o);
            }
            __syncthreads();                                           // (flag word = STAT[0]: read by all before norm2' writes it)
        }
        wps.start();
        // what norm2' needs: x1 slice, row statistics, gamma; and dy for the residual
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gam[i] = ld4(a.g2 + 32 * wid + 16 * i + 4 * gq);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xv[i][g] = ld_saved((const f32x4*)(a.x1 + rows[g] * C + 32 * wid + 16 * i + 4 * gq));
                dyv[i][g] = ld4(a.dx + rows[g] * C + 32 * wid + 16 * i + 4 * gq);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) { mu[g] = a.mean2[rows[g]]; rs[g] = a.rstd2[rows[g]]; }
        ln_bwd_part1<C, G, T>(dx1, xh, xv, mu, rs, gam, a.lnpart2 + (size_t)wblk * 2 * C, STAT, wid, t, gq);
        __syncthreads();
        ln_bwd_part2<C, G, T, NWV>(dx1, xh, rs, STAT, t);
        // d(x1) = dy + norm2'(d(xn2))  (residual, tulip.py:351); its bf16 copy * s_attn feeds proj' and proj's wgrad
        bf16x4 pka[2][G];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = 32 * wid + 16 * i + 4 * gq;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                dx1[i][g] = dx1[i][g] + dyv[i][g];
                pka[i][g] = pack4(dx1[i][g][0] * s0, dx1[i][g][1] * s0, dx1[i][g][2] * s0, dx1[i][g][3] * s0);
                put4<T>(DY, 16 * g + t, c0, pka[i][g]);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) store_bf16_tile_pair<true>(a.dyb_a + rows[g] * C + 32 * wid, pka[0][g], pka[1][g], gq);
    }
    __syncthreads();

    // ---- proj' (tulip.py:318 backwards): dO of this wave's head
    WStream<2, 3 * KS, 3 * D> wqs;                  // qkv^T: this wave's 32 channels of d(xn1)
#pragma unroll
    for (int i = 0; i < 2; ++i) wqs.wt[i] = wtile_ptr(a.wqkvt, 2 * wid + i, 3 * C, lane);
    bf16x4 dop[2][G];
    {
        f32x4 acc[2][G];
        zero(acc);
        wps.template run<G, T>(acc, DY, t, gq);
        wqs.start();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) dop[i][g] = pack4(acc[i][g][0], acc[i][g][1], acc[i][g][2], acc[i][g][3]);
    }
    // q, k, v of this head for the attention backward
    bf16x4 qkvr[6][G];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g)
            qkvr[i][g] = ld_saved((const bf16x4*)(a.qkv + rows[g] * (3 * C) + (i >> 1) * C + 32 * wid + 16 * (i & 1) + 4 * gq));
    // ---- attention' of this head, one window at a time (tulip.py:300-317 backwards; algebra of attn_bwd_kernel)
    {
        unsigned char* ldsQ = smem + Z::OFF_ATT + wid * 3072;
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
                    const int n = sec * C + 32 * wid + 16 * dc + 4 * gq;
                    put4<T>(DQ, 16 * g + t, n, o[sec]);
                    if (dc) store_bf16_tile_pair<true>(a.dqkv + rows[g] * (3 * C) + sec * C + 32 * wid, oprev[sec], o[sec], gq);
                    else oprev[sec] = o[sec];
                }
            }
        }
        // dense relative-position-bias gradient of this head, summed over the workgroup's windows: [NH][16 q][16 k]
        *(float4*)(a.biaspart + (size_t)wblk * (NH * 256) + wid * 256 + t * 16 + gq * 4) =
            make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
    }
    __syncthreads();

    // ---- qkv' (tulip.py:298 backwards), norm1' and the residual
    {
        f32x4 acc[2][G], xh[2][G];
        zero(acc);
        wqs.template run<G, T>(acc, DQ, t, gq);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gam[i] = ld4(a.g1 + 32 * wid + 16 * i + 4 * gq);
#pragma unroll
            for (int g = 0; g < G; ++g) xv[i][g] = ld_saved((const f32x4*)(a.xin + rows[g] * C + 32 * wid + 16 * i + 4 * gq));
        }
#pragma unroll
        for (int g = 0; g < G; ++g) { mu[g] = a.mean1[rows[g]]; rs[g] = a.rstd1[rows[g]]; }
        ln_bwd_part1<C, G, T>(acc, xh, xv, mu, rs, gam, a.lnpart1 + (size_t)wblk * 2 * C, STAT, wid, t, gq);
        __syncthreads();
        ln_bwd_part2<C, G, T, NWV>(acc, xh, rs, STAT, t);
        const float cs = a.dx_scale ? a.dx_scale[tm.b] : 1.0f;
        bf16x4 oc[2][G];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = 32 * wid + 16 * i + 4 * gq;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 o = dx1[i][g] + acc[i][g];
                *(float4*)(a.dx + rows[g] * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
                oc[i][g] = pack4(o[0] * cs, o[1] * cs, o[2] * cs, o[3] * cs);
            }
        }
        if (a.dx_bf16) {
#pragma unroll
            for (int g = 0; g < G; ++g) store_bf16_tile_pair(a.dx_bf16 + rows[g] * C + 32 * wid, oc[0][g], oc[1][g], gq);
        }
    }
}

int launch_bwd_split(const SwinWBwdArgs& a, hipStream_t stream) {
#if TULIP_DEV_VARIANTS      // built, bit-tested, 5.6 us faster per launch in isolation and 34 us slower per step (profiles/README.md): not shipped
    const int windows = a.B * (a.H / 2) * (a.W / 8);
    const dim3 grid(2 * windows), block(GeoB<384, 1>::NT);
    if (!(a.masked & TULIP_BLOCK_NO_WARM)) hipLaunchKernelGGL((swinw_bwd_kernel<384, 1, 13>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((swinw_bwd_kernel<384, 1, 12>), grid, block, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
#else
    return TULIP_ERR_NOT_BUILT;
#endif
}
template <int C, int G>
int launch_bwd(const SwinWBwdArgs& a, hipStream_t stream) {
    const int blocks = a.B * (a.H / 2) * (a.W / (8 * G));
    const bool warm_on = !(a.masked & TULIP_BLOCK_NO_WARM);
    const int warm = (blocks <= 256 && blocks >= 8 && warm_on) ? 1 : (blocks > 256 && warm_on) ? 2 : 0;
    const dim3 grid(blocks), block(GeoB<C, G>::NT);
    if (a.masked & TULIP_BLOCK_FC1_GRAD) {      // the fc1_pre buffer holds gelu'(h) (bit 2 of the template's mode)
        if (warm == 1) hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 5>), grid, block, 0, stream, a);
        else if (warm == 2) hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 6>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 4>), grid, block, 0, stream, a);
    } else {                                    // h itself in the fc1_pre buffer: development build only
#if TULIP_DEV_VARIANTS
        if (warm == 1) hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 1>), grid, block, 0, stream, a);
        else if (warm == 2) hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 2>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((swinw_bwd_kernel<C, G, 0>), grid, block, 0, stream, a);
#else
        return TULIP_ERR_NOT_BUILT;
#endif
    }
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

__host__ __device__ inline bool wide_g4(int C, int B, int H, int W) {
    // windows per workgroup: 4 once that still gives every CU a workgroup, else 2 (every TULIP grid has W % 16 == 0)
    return C == 192 && (W % 32 == 0) && (B * (H / 2) * (W / 8)) / 4 >= 256;
}
#ifndef TULIP_SWINW_G1_BELOW
#define TULIP_SWINW_G1_BELOW 128
#endif
__host__ __device__ inline int wide_g(int C, int B, int H, int W) {
    if (wide_g4(C, B, H, W)) return 4;
    // C = 384 with few windows (batch 8: 128): one window per workgroup, twice the workgroups streaming the weights
    // (forward 66 -> 50 us, backward 65 -> 51 us at batch 8)
    if (C == 384 && (B * (H / 2) * (W / 8)) / 2 < TULIP_SWINW_G1_BELOW) return 1;
    return 2;
}

// ---- fragment-major copies of a list of bf16 matrices: mode 0: dst = packed(src [rows][cols]);  mode 1: dst =
// packed(src^T) (a [cols][rows] matrix), through an LDS tile transpose so that both sides move 16-byte pieces
struct TrItem { const bf16_t* src; bf16_t* dst; int rows, cols, mode, first; };
struct TrList { TrItem it[TULIP_PACK_MAX]; int n; };
__device__ __forceinline__ size_t packed_offset(int n, int k, int K) {      // element (n, k) of a [N][K] matrix
    return ((size_t)(n >> 4) * (K >> 5) + (k >> 5)) * 512 + ((n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
}
#ifndef TULIP_PACK_LOAD_NT
#define TULIP_PACK_LOAD_NT 0
#endif
__device__ __forceinline__ uint4 pack_src(const uint4* p) {
#if TULIP_PACK_LOAD_NT
    typedef unsigned u32x4_p __attribute__((ext_vector_type(4)));
    const u32x4_p v = __builtin_nontemporal_load((const u32x4_p*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *p;
#endif
}
__global__ __launch_bounds__(256) void pack_multi_kernel(const TrList L) {
    __shared__ bf16_t tile[64][66];
    int i = 0;
    while (i + 1 < L.n && (int)blockIdx.x >= L.it[i + 1].first) ++i;
    const TrItem& m = L.it[i];
    const int tc = (m.cols + 63) / 64;
    const int b = blockIdx.x - m.first, r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    if (m.mode == 0) {
        for (int e = threadIdx.x; e < 64 * 8; e += 256) {
            const int r = r0 + (e >> 3), c8 = c0 + (e & 7) * 8;
            if (r < m.rows && c8 < m.cols)
                *(uint4*)(m.dst + packed_offset(r, c8, m.cols)) = pack_src((const uint4*)(m.src + (size_t)r * m.cols + c8));
        }
        return;
    }
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
        const int r = e >> 3, c8 = (e & 7) * 8;
        if (r0 + r < m.rows && c0 + c8 < m.cols) {
            const uint4 v = pack_src((const uint4*)(m.src + (size_t)(r0 + r) * m.cols + c0 + c8));
            const bf16_t* pv = (const bf16_t*)&v;
#pragma unroll
            for (int k = 0; k < 8; ++k) tile[r][c8 + k] = pv[k];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
        const int c = e >> 3, r8 = (e & 7) * 8;
        if (c0 + c < m.cols && r0 + r8 < m.rows) {
            bf16_t o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = tile[r8 + k][c];
            *(uint4*)(m.dst + packed_offset(c0 + c, r0 + r8, m.rows)) = *(const uint4*)o;    // transposed: [cols][rows]
        }
    }
}

}  // namespace


extern "C" int tulip_swinw_supported(int C, int H, int W) {
    return (C == 192 || C == 384) && H > 0 && !(H & 1) && W > 0 && !(W & 15);
}

static int swinw_fwd_impl(const tulip_swin96_desc* d, int C, void* out_bf16, unsigned long long* prof, void* exchange,
                          size_t exchange_bytes, hipStream_t stream);
extern "C" int tulip_swinw_block_fwd(const tulip_swin96_desc* d, int C, void* out_bf16, hipStream_t stream) {
    return swinw_fwd_impl(d, C, out_bf16, nullptr, nullptr, 0, stream);
}
extern "C" int tulip_swinw_block_fwd_profiled(const tulip_swin96_desc* d, int C, void* out_bf16, uint64_t* stamps,
                                              hipStream_t stream) {
    return swinw_fwd_impl(d, C, out_bf16, (unsigned long long*)stamps, nullptr, 0, stream);
}
// bytes of the exchange buffer of the split form (two workgroups per window), 0 where that form does not exist: per window two
// 16 x C fp32 partial tiles in accumulator order, then one arrival ticket per window (zero before the first launch; the
// kernels leave it zero)
extern "C" int tulip_swinw_split_bytes(int C, int B, int H, int W) {
    if (C != 384 || B <= 0 || !tulip_swinw_supported(C, H, W) || wide_g(C, B, H, W) != 1) return 0;
    const int windows = B * (H / 2) * (W / 8);          // (< TULIP_SWINW_G1_BELOW * 2)
    return windows * (2 * 2 * Geo<384, 1>::NT * 16) + windows * 4;
}
extern "C" int tulip_swinw_block_fwd_split(const tulip_swin96_desc* d, int C, void* out_bf16, void* exchange, size_t exchange_bytes,
                                           uint64_t* stamps, hipStream_t stream) {
    if (!d || !exchange) return TULIP_ERR_ARG;
    const size_t need = (size_t)tulip_swinw_split_bytes(C, d->B, d->H, d->W);
    if (need == 0 || exchange_bytes < need || ((uintptr_t)exchange & 15)) return TULIP_ERR_ARG;
    if (d->qkv && !(d->masked & TULIP_BLOCK_FC1_GRAD)) return TULIP_ERR_ARG;         // training form: with gelu'(h) handed over only
    return swinw_fwd_impl(d, C, out_bf16, (unsigned long long*)stamps, exchange, exchange_bytes, stream);
}
static int swinw_fwd_impl(const tulip_swin96_desc* d, int C, void* out_bf16, unsigned long long* prof, void* exchange,
                          size_t exchange_bytes, hipStream_t stream) {
    if (!d || d->B <= 0 || !tulip_swinw_supported(C, d->H, d->W) || d->shift_h < 0 || d->shift_h >= d->H ||
        d->shift_w < 0 || d->shift_w >= d->W)
        return TULIP_ERR_ARG;
    SwinWArgs a;
    a.xin = d->x_in; a.x1 = d->x1; a.xout = d->x_out;
    a.xn1 = (bf16_t*)d->xn1; a.qkv = (bf16_t*)d->qkv; a.o = (bf16_t*)d->attn_out; a.xn2 = (bf16_t*)d->xn2;
    a.h = (bf16_t*)d->fc1_pre; a.g = (bf16_t*)d->fc1_act;
    a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.wqkv = (const bf16_t*)d->w_qkv; a.wproj = (const bf16_t*)d->w_proj; a.w1 = (const bf16_t*)d->w_fc1;
    a.w2 = (const bf16_t*)d->w_fc2;
    a.bqkv = d->b_qkv; a.bproj = d->b_proj; a.b1 = d->b_fc1; a.b2 = d->b_fc2;
    a.g1 = d->norm1_weight; a.be1 = d->norm1_bias; a.g2 = d->norm2_weight; a.be2 = d->norm2_bias;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.out_bf16 = (bf16_t*)out_bf16;
    a.prof = prof;
#if !TULIP_DEV_VARIANTS
    if (prof) return TULIP_ERR_NOT_BUILT;
#endif
    {   // every saved-activation pointer NULL: the inference form
        const bool any = d->xn1 || d->qkv || d->attn_out || d->x1 || d->xn2 || d->fc1_pre || d->fc1_act || d->mean1 || d->rstd1 ||
                         d->mean2 || d->rstd2;
        const bool all = d->xn1 && d->qkv && d->attn_out && d->x1 && d->xn2 && d->fc1_pre && d->fc1_act && d->mean1 && d->rstd1 &&
                         d->mean2 && d->rstd2;
        if (any && !all) return TULIP_ERR_ARG;
    }
    a.B = d->B; a.H = d->H; a.W = d->W; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.eps = d->eps; a.scale = 0.17677669529663687f;        // head_dim^-0.5 = 32^-0.5 (tulip.py:220)
    a.xws = nullptr; a.tick = nullptr; a.xws_bytes = 0;
    if (exchange) {
        const size_t windows = (size_t)d->B * (d->H / 2) * (d->W / 8);
        a.xws = (float*)exchange;
        a.xws_bytes = (unsigned)(windows * (2 * 2 * Geo<384, 1>::NT * 16));
        a.tick = (unsigned*)((unsigned char*)exchange + a.xws_bytes);
        return launch_fwd_split(a, stream);
    }
    if (C == 192) return wide_g4(C, d->B, d->H, d->W) ? launch_fwd<192, 4>(a, stream) : launch_fwd<192, 2>(a, stream);
    return wide_g(C, d->B, d->H, d->W) == 1 ? launch_fwd<384, 1>(a, stream) : launch_fwd<384, 2>(a, stream);
}

extern "C" int tulip_swinw_bwd_partial_rows(int C, int B, int H, int W) {
    if (B <= 0 || !tulip_swinw_supported(C, H, W)) return 0;
    return B * (H / 2) * (W / (8 * wide_g(C, B, H, W)));
}

static int swinw_bwd_impl(const tulip_swin96_bwd_desc* d, int C, void* exchange, hipStream_t stream);
extern "C" int tulip_swinw_block_bwd(const tulip_swin96_bwd_desc* d, int C, hipStream_t stream) {
    return swinw_bwd_impl(d, C, nullptr, stream);
}
extern "C" int tulip_swinw_block_bwd_split(const tulip_swin96_bwd_desc* d, int C, void* exchange, size_t exchange_bytes,
                                           hipStream_t stream) {
    if (!d || !exchange) return TULIP_ERR_ARG;
    const size_t need = (size_t)tulip_swinw_split_bytes(C, d->B, d->H, d->W);
    if (need == 0 || exchange_bytes < need || ((uintptr_t)exchange & 15)) return TULIP_ERR_ARG;
    if (!(d->masked & TULIP_BLOCK_FC1_GRAD)) return TULIP_ERR_ARG;
    return swinw_bwd_impl(d, C, exchange, stream);
}
static int swinw_bwd_impl(const tulip_swin96_bwd_desc* d, int C, void* exchange, hipStream_t stream) {
    if (!d || d->B <= 0 || !tulip_swinw_supported(C, d->H, d->W) || d->shift_h < 0 || d->shift_h >= d->H ||
        d->shift_w < 0 || d->shift_w >= d->W)
        return TULIP_ERR_ARG;
    SwinWBwdArgs a;
    a.dx = d->dx; a.xin = d->x_in; a.x1 = d->x1;
    a.qkv = (const bf16_t*)d->qkv; a.h = (const bf16_t*)d->fc1_pre;
    a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.wqkvt = (const bf16_t*)d->w_qkv; a.wprojt = (const bf16_t*)d->w_proj; a.w1t = (const bf16_t*)d->w_fc1;
    a.w2t = (const bf16_t*)d->w_fc2;
    a.g1 = d->norm1_weight; a.g2 = d->norm2_weight;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.dyb_m = (bf16_t*)d->d_out_mlp; a.dh = (bf16_t*)d->d_fc1_pre; a.dyb_a = (bf16_t*)d->d_out_attn;
    a.dqkv = (bf16_t*)d->d_qkv;
    a.dx_bf16 = (bf16_t*)d->dx_bf16; a.dx_scale = d->dx_bf16_scale;
    a.lnpart1 = d->norm1_partials; a.lnpart2 = d->norm2_partials; a.biaspart = d->bias_partials;
    a.B = d->B; a.H = d->H; a.W = d->W; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.scale = 0.17677669529663687f;
    a.xws = nullptr; a.tick = nullptr; a.xws_bytes = 0;
    if (exchange) {
        const size_t windows = (size_t)d->B * (d->H / 2) * (d->W / 8);
        a.xws = (float*)exchange;
        a.xws_bytes = (unsigned)(windows * (2 * 2 * GeoB<384, 1>::NT * 16));
        a.tick = (unsigned*)((unsigned char*)exchange + a.xws_bytes);
        return launch_bwd_split(a, stream);
    }
    if (C == 192) return wide_g4(C, d->B, d->H, d->W) ? launch_bwd<192, 4>(a, stream) : launch_bwd<192, 2>(a, stream);
    return wide_g(C, d->B, d->H, d->W) == 1 ? launch_bwd<384, 1>(a, stream) : launch_bwd<384, 2>(a, stream);
}

extern "C" int tulip_pack_bf16_multi(const tulip_pack_item* items, int n, hipStream_t stream) {
    if (n < 0 || n > TULIP_PACK_MAX || (n && !items)) return TULIP_ERR_ARG;
    TrList L;
    L.n = 0;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        const tulip_pack_item& it = items[i];
        if (it.rows <= 0 || it.cols <= 0) continue;
        const int N = it.transpose ? it.cols : it.rows, K = it.transpose ? it.rows : it.cols;
        if (!it.src || !it.dst || (N & 15) || (K & 31)) return TULIP_ERR_ARG;
        L.it[L.n++] = TrItem{(const bf16_t*)it.src, (bf16_t*)it.dst, it.rows, it.cols, it.transpose ? 1 : 0, first};
        first += ((it.rows + 63) / 64) * ((it.cols + 63) / 64);
    }
    if (L.n == 0) return TULIP_OK;
    hipLaunchKernelGGL(pack_multi_kernel, dim3(first), dim3(256), 0, stream, L);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
