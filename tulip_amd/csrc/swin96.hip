// One launch for the forward of a whole Swin block at embed width 96 (stage 0 of every TULIP model: 3 heads x 32,
// window 2x8, MLP 96 -> 384 -> 96; tulip.py:338-352 with :289-323 and :194-200 inside).
//
// A workgroup owns 8 neighbouring windows = 128 tokens, one window per wave, and never talks to another workgroup:
// norm1 -> qkv -> shifted-window attention -> proj -> +residual -> norm2 -> fc1 -> GELU -> fc2 -> +residual all stay
// in registers.  The trick that makes the chain LDS-free for activations: with the MFMA issued as W . X^T, a lane of
// the accumulator holds 4 consecutive output channels of ONE token (its column); two such accumulators, rounded to
// bf16, are exactly a 32-deep operand fragment of the next GEMM up to a fixed permutation of k -- and a permutation
// of k is free if the weight fragment is read with the same permutation (two 8-byte LDS reads instead of one
// 16-byte read).  Only V goes through a 1-KiB per-wave LDS tile (transpose read for P.V), and the weights are
// staged in LDS (W2 | {Wqkv,Wproj} then W1 in the same region, plus the bias / norm2 vectors: 163.7 of 163.8 KB).
// Everything the backward needs (xn1, qkv, o, x1, xn2, h, g, the LayerNorm statistics) is written exactly as the
// separate kernels write it, so the backward is unchanged.
#ifndef TULIP_STORE_LATE_96
#define TULIP_STORE_LATE_96 0          // plain stores measured faster here (see swinw.hip)
#endif
#define TULIP_STORE_LATE TULIP_STORE_LATE_96
#include "common.h"
#include "tulip_hip.h"

namespace {

#ifndef TULIP_SWIN96_IGLP
#define TULIP_SWIN96_IGLP 0          // dev: -DTULIP_SWIN96_IGLP=1 asks the scheduler for its DS-read / MFMA interleave (iglp_opt 0)
#endif
#if TULIP_SWIN96_IGLP
#define IGLP() __builtin_amdgcn_iglp_opt(0)
#else
#define IGLP() ((void)0)
#endif
constexpr int C = 96, HID = 384;
constexpr int NW = 8;                     // windows (= waves) per workgroup: 128 tokens share one copy of the weights
constexpr int NT = NW * 64;
constexpr int PW = 200;                   // LDS pitch of a 96-wide bf16 weight row (192 B + 8): conflict-free 8-B reads
constexpr int PW2 = 776;                  // LDS pitch of a 384-wide bf16 weight row (768 B + 8)
constexpr int OFF_W2 = 0;                 // fc2.weight  [96][384]
constexpr int OFF_A = 96 * PW2;           // phase 1: qkv.weight [288][96] | proj.weight [96][96]; phase 2: fc1.weight [384][96]
constexpr int OFF_WPROJ = OFF_A + 288 * PW;
constexpr int OFF_V = OFF_A + 384 * PW;   // NW x 1 KiB V tiles
constexpr int OFF_P = OFF_V + NW * 1024;  // fp32 vectors: b_fc1[384] b_qkv[288] b_proj[96] b_fc2[96] norm2.weight[96] norm2.bias[96]
constexpr int P_B1 = 0, P_BQKV = 384, P_BPROJ = 672, P_B2 = 768, P_G2 = 864, P_BE2 = 960, P_N = 1056;
constexpr int SMEM = OFF_P + P_N * 4;     // 163712 B of 163840

struct Swin96Args {
    const float* xin; float* x1; float* xout;
    bf16_t *xn1, *qkv, *o, *xn2, *h, *g;
    float *mean1, *rstd1, *mean2, *rstd2;
    const bf16_t *wqkv, *wproj, *w1, *w2;
    const float *bqkv, *bproj, *b1, *b2, *g1, *be1, *g2, *be2;
    const float* bias_table; const int* rel_index;
    const float *ds0, *ds1;               // DropPath multipliers per sample (attention / MLP branch) or nullptr
    int B, H, W, sh, sw, masked;
    float eps, scale;
    unsigned long long* prof;             // optional: s_memtime stamps [workgroup][wave][16] at the phase boundaries (dev)
};
#define TULIP_STAMP(k) do { if constexpr (PROF) { if (lane == 0) a.prof[((size_t)blockIdx.x * NW + wid) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)

__device__ __forceinline__ int region(int x, int X, int wsz, int ssz) {       // create_mask slices, tulip.py:261-266
    return (ssz == 0 || x >= X - ssz) ? 2 : (x >= X - wsz ? 1 : 0);
}
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    return __builtin_bit_cast(bf16x4, (u32x2_t){pack_bf16x2(a, b), pack_bf16x2(c, d)});
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x4 lo4(bf16x8 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ __forceinline__ bf16x4 hi4(bf16x8 v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }
// ds_read_b64_tr_b16 through the builtin: the compiler batches the waits of consecutive reads
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)p);
}
// weight fragment in the chained-operand k order: k slots 0..3 <- columns c0..c0+3, slots 4..7 <- c0+16..c0+19
__device__ __forceinline__ bf16x8 wfrag(const unsigned char* rowp, int c0) {
    return cat8(*(const bf16x4*)(rowp + c0 * 2), *(const bf16x4*)(rowp + (c0 + 16) * 2));
}
// LayerNorm output of one token row held as 6 x 4 channels (16n + 4gq + r), rounded to bf16: the ONE place the forward and the
// backward's recomputation (swin96_bwd_kernel<true>) go through, with floating-point contraction pinned -- the backward must see
// exactly the operand bits the forward's GEMMs saw, or its recomputed qkv / fc1 pre-activation would differ from the forward's
// in the last bf16 digit now and then
__device__ __forceinline__ void ln_apply(const f32x4 (&xv)[6], float mu, float rs, const float* gam, const float* bet, int gq,
                                         bf16x4 (&p)[6]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const int c0 = 16 * n + 4 * gq;
        const float4 ga = *(const float4*)(gam + c0), be = *(const float4*)(bet + c0);
        p[n] = pack4(__builtin_fmaf((xv[n][0] - mu) * rs, ga.x, be.x), __builtin_fmaf((xv[n][1] - mu) * rs, ga.y, be.y),
                     __builtin_fmaf((xv[n][2] - mu) * rs, ga.z, be.z), __builtin_fmaf((xv[n][3] - mu) * rs, ga.w, be.w));
    }
}
// global [ROWS][COLS] bf16 -> LDS rows of PITCH bytes; every thread keeps PER 16-byte loads in flight
template <int ROWS, int COLS, int PITCH>
__device__ __forceinline__ void stage_weights(const bf16_t* __restrict__ w, unsigned char* dst, int tid) {
    constexpr int CPR = COLS / 8, N = ROWS * CPR, PER = (N + NT - 1) / NT;
    uint4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * NT;
        if (N % NT == 0 || c < N) v[i] = *(const uint4*)(w + (size_t)(c / CPR) * COLS + (c % CPR) * 8);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * NT;
        if (N % NT == 0 || c < N) {
            unsigned char* d = dst + (c / CPR) * PITCH + (c % CPR) * 16;
            *(uint2*)d = make_uint2(v[i].x, v[i].y);
            *(uint2*)(d + 8) = make_uint2(v[i].z, v[i].w);
        }
    }
}

// the same in two steps: the loads early (before a burst of stores -- vmcnt retires in order, a load issued behind stores
// cannot be consumed before they are acknowledged), the LDS writes once the region is free
template <int ROWS, int COLS>
struct StagedWeights {
    static constexpr int CPR = COLS / 8, N = ROWS * CPR, PER = (N + NT - 1) / NT;
    uint4 v[PER];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ w, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            v[i] = *(const uint4*)(w + (size_t)((N % NT == 0 || c < N) ? c : N - 1) * 8);     // rows are contiguous: chunk c
        }
    }
    template <int PITCH>
    __device__ __forceinline__ void store(unsigned char* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            if (N % NT == 0 || c < N) {
                unsigned char* d = dst + (c / CPR) * PITCH + (c % CPR) * 16;
                *(uint2*)d = make_uint2(v[i].x, v[i].y);
                *(uint2*)(d + 8) = make_uint2(v[i].z, v[i].w);
            }
        }
    }
};

// SAVE = 0: the inference form (eval / MC-dropout forward: nothing is kept for a backward) -- 116 of the 129 MB a launch
// moves at batch 8 are the saved activations.  SAVE = 1 (round 4): what the fused backward with recomputation needs -- x1, the
// statistics and the four bf16 operands of the weight-gradient GEMMs (xn1, attention output, xn2, gelu(h)); qkv and the fc1
// pre-activation h (1344 of 3472 B per token) are recomputed by swin96_bwd_kernel<true> from x / x1.  SAVE = 2: everything,
// as the separate kernels write it (the unfused backward reads qkv and h).
// SAVE = 3: as 2, but the fc1_pre buffer receives bf16(gelu'(h)) instead of h (TULIP_BLOCK_FC1_GRAD): the only thing the
// backward does with h is that derivative (~12 vector instructions per element there, two more here where erf and the
// Gaussian are at hand anyway).  PROF: the diagnostic twin with shader-clock stamps (tools/swin96_phases.py).
// HAND (the two-block launch below), bits: 1 = the block output leaves with write-through (sc1) stores, 2 = the block input is read
// with sc1 loads -- the lines another XCD's workgroup wrote in this same launch; 0 = plain (a launch of its own)
template <int SAVE, bool PROF, int HAND, class ARGS>
__device__ __forceinline__ void swin96_fwd_tile(ARGS& a, int blk, unsigned char* smem) {
    typedef unsigned u32x4_h __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    unsigned char* ldsV = smem + OFF_V + wid * 1024;
    float* prm = (float*)(smem + OFF_P);

    // ---- which window, which token (cyclic shift + window partition are address arithmetic, tulip.py:289-297)
    const int nWx = a.W >> 3, nWy = a.H >> 1, gpr = nWx / NW;
    const int b = blk / (nWy * gpr);
    blk -= b * nWy * gpr;
    const int wy = blk / gpr, wx = (blk - wy * gpr) * NW + wid;
    const int hs = wy * 2 + (t >> 3), ws = wx * 8 + (t & 7);
    int hh = hs + a.sh; if (hh >= a.H) hh -= a.H;
    int ww = ws + a.sw; if (ww >= a.W) ww -= a.W;
    const size_t row = ((size_t)b * a.H + hh) * a.W + ww;
    const int lab = 3 * region(hs, a.H, 2, a.sh) + region(ws, a.W, 8, a.sw);
    TULIP_STAMP(0);

    // the token's row: lane owns channels 16n + 4gq .. +3 (n = 0..5) -- the accumulator layout of every GEMM below
    f32x4 xv[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        if constexpr (HAND & 2) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xin), 0, 0x7FFFFFFF, 0x00020000);
            xv[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(row * C + 16 * n + 4 * gq) * 4u, 0, 16));
        } else {
            const float4 u = *(const float4*)(a.xin + row * C + 16 * n + 4 * gq);
            xv[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
    }
    // relative-position bias of this lane's (query t, keys 4gq..4gq+3), all heads (tulip.py:304-308)
    float rpb[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = a.rel_index[t * 16 + gq * 4 + r] * 3;
#pragma unroll
        for (int h = 0; h < 3; ++h) rpb[h][r] = a.bias_table[e + h];
    }
    // Every load of the prologue is issued before the first LDS write (round 4: three stage_weights calls and the parameter loop
    // were six dependent L2 round trips, 12 k of the kernel's 50 k cycles at batch 8 -- tools/swin96_phases.py)
    {
        StagedWeights<288, C> sq; StagedWeights<C, C> sp; StagedWeights<C, HID> s2;
        sq.load(a.wqkv, tid);
        sp.load(a.wproj, tid);
        s2.load(a.w2, tid);
        // b_fc1[384] b_qkv[288] b_proj[96] b_fc2[96] norm2.weight[96] norm2.bias[96] as 264 float4
        const int i4 = tid * 4;
        const float* src = i4 < P_BQKV ? a.b1 + i4 : i4 < P_BPROJ ? a.bqkv + (i4 - P_BQKV) : i4 < P_B2 ? a.bproj + (i4 - P_BPROJ)
                         : i4 < P_G2 ? a.b2 + (i4 - P_B2) : i4 < P_BE2 ? a.g2 + (i4 - P_G2) : a.be2 + (i4 - P_BE2);
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < P_N) pv = *(const float4*)src;
        sq.template store<PW>(smem + OFF_A, tid);
        sp.template store<PW>(smem + OFF_WPROJ, tid);
        s2.template store<PW2>(smem + OFF_W2, tid);
        if (i4 < P_N) *(float4*)(prm + i4) = pv;
    }

    // ---- norm1 (tulip.py:340)
    bf16x8 xfrag[3];
    {
        float s1 = 0.f;
#pragma unroll
        for (int n = 0; n < 6; ++n) s1 += (xv[n][0] + xv[n][1]) + (xv[n][2] + xv[n][3]);
        s1 = rows_sum(s1);
        const float mu = s1 * (1.0f / C);
        float s2 = 0.f;
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = xv[n][r] - mu; s2 += d * d; }
        s2 = rows_sum(s2);
        const float rs = rsqrtf(s2 * (1.0f / C) + a.eps);
        if (SAVE && gq == 0) { a.mean1[row] = mu; a.rstd1[row] = rs; }
        bf16x4 p1[6];
        ln_apply(xv, mu, rs, a.g1, a.be1, gq, p1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (SAVE != 0) store_bf16_tile_pair<true>(a.xn1 + row * C + 32 * s, p1[2 * s], p1[2 * s + 1], gq);
            xfrag[s] = cat8(p1[2 * s], p1[2 * s + 1]);   // k order within 32s: 4gq.., 16+4gq..
        }
    }
    TULIP_STAMP(1);
    __syncthreads();                                        // weights of phase 1, W2 and the parameter vectors are in LDS
    TULIP_STAMP(2);

    IGLP();
    // ---- qkv Linear (tulip.py:298): acc lane = 4 consecutive output channels 16j + 4gq + r of token t
    bf16x4 qkvp[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* wr = smem + OFF_A + (16 * j + t) * PW;
#pragma unroll
        for (int s = 0; s < 3; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(wr, 32 * s + 4 * gq), xfrag[s], acc, 0, 0, 0);
        const float4 bq = *(const float4*)(prm + P_BQKV + 16 * j + 4 * gq);
        qkvp[j] = pack4(acc[0] + bq.x, acc[1] + bq.y, acc[2] + bq.z, acc[3] + bq.w);
        if (SAVE >= 2 && (j & 1)) store_bf16_tile_pair<true>(a.qkv + row * 288 + 16 * (j - 1), qkvp[j - 1], qkvp[j], gq);
    }

    // fc1.weight rows 0..287 replace qkv.weight in LDS once every wave is through the qkv GEMM: fetched now, written
    // behind the attention core (the stall of a restage between two barriers was ~10 % of this kernel)
    TULIP_STAMP(3);
    StagedWeights<288, C> w1a;
    w1a.load(a.w1, tid);

    // ---- attention, one head at a time (tulip.py:300-317); scores issued as K.Q^T: lane = query t, keys 4gq + r
    bf16x8 ofrag[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const bf16x8 qf = cat8(qkvp[2 * h], qkvp[2 * h + 1]);
        const bf16x8 kf = cat8(qkvp[6 + 2 * h], qkvp[7 + 2 * h]);
        *(bf16x4*)(ldsV + t * 64 + (4 * gq) * 2) = qkvp[12 + 2 * h];            // V tile [token][d], d = 0..15
        *(bf16x4*)(ldsV + t * 64 + (16 + 4 * gq) * 2) = qkvp[13 + 2 * h];       // d = 16..31
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        if (a.masked & TULIP_ATTN_FP8) sc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf16x8_to_fp8(kf), bf16x8_to_fp8(qf), sc, 0, 0, 0);
        else sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sc, 0, 0, 0);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = sc[r] * a.scale + rpb[h][r];
            if (a.masked & TULIP_ATTN_MASKED) {
                const int kl = __shfl(lab, gq * 4 + r, 64);
                if (kl != lab) x += -100.0f;
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
        bf16x4 op[2];
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const bf16x4 vt = tr_read(ldsV + (gq * 4 + (t >> 2)) * 64 + dc * 32 + (t & 3) * 8);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb, o, 0, 0, 0);   // o[r] = O[t][16dc + 4gq + r]
            op[dc] = pack4(o[0], o[1], o[2], o[3]);
        }
        if constexpr (SAVE != 0) store_bf16_tile_pair<true>(a.o + row * C + 32 * h, op[0], op[1], gq);
        ofrag[h] = cat8(op[0], op[1]);                    // k order: d = 4gq+0..3, 16+4gq+0..3
    }

    TULIP_STAMP(4);
    __syncthreads();                                        // nobody reads qkv.weight any more
    TULIP_STAMP(5);
    w1a.template store<PW>(smem + OFF_A, tid);
    StagedWeights<96, C> w1b;                               // rows 288..383 go where proj.weight is: behind the proj GEMM
    w1b.load(a.w1 + 288 * C, tid);

    IGLP();
    // ---- proj Linear + DropPath + residual (tulip.py:318,344), then norm2 (:347); x1 replaces x in xv
    const float s0 = a.ds0 ? a.ds0[b] : 1.0f, s1v = a.ds1 ? a.ds1[b] : 1.0f;
    bf16x8 x2frag[3];
    {
        float sum = 0.f;
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* wr = smem + OFF_WPROJ + (16 * n2 + t) * PW;
#pragma unroll
            for (int h = 0; h < 3; ++h)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(wr, 32 * h + 4 * gq), ofrag[h], acc, 0, 0, 0);
            const int c0 = 16 * n2 + 4 * gq;
            const float4 bp = *(const float4*)(prm + P_BPROJ + c0);
            xv[n2] = (f32x4){xv[n2][0] + s0 * (acc[0] + bp.x), xv[n2][1] + s0 * (acc[1] + bp.y),
                             xv[n2][2] + s0 * (acc[2] + bp.z), xv[n2][3] + s0 * (acc[3] + bp.w)};
            if constexpr (SAVE != 0) store_late((float4*)(a.x1 + row * C + c0), make_float4(xv[n2][0], xv[n2][1], xv[n2][2], xv[n2][3]));
            sum += (xv[n2][0] + xv[n2][1]) + (xv[n2][2] + xv[n2][3]);
        }
        sum = rows_sum(sum);
        const float mu = sum * (1.0f / C);
        float s2 = 0.f;
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = xv[n2][r] - mu; s2 += d * d; }
        s2 = rows_sum(s2);
        const float rs = rsqrtf(s2 * (1.0f / C) + a.eps);
        if (SAVE && gq == 0) { a.mean2[row] = mu; a.rstd2[row] = rs; }
        bf16x4 p2[6];
        ln_apply(xv, mu, rs, prm + P_G2, prm + P_BE2, gq, p2);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (SAVE != 0) store_bf16_tile_pair<true>(a.xn2 + row * C + 32 * s, p2[2 * s], p2[2 * s + 1], gq);
            x2frag[s] = cat8(p2[2 * s], p2[2 * s + 1]);
        }
    }

    // ---- the rest of fc1.weight replaces proj.weight
    TULIP_STAMP(6);
    __syncthreads();
    w1b.template store<PW>(smem + OFF_WPROJ, tid);
    __syncthreads();
    TULIP_STAMP(7);

    // ---- fc1 -> exact-erf GELU -> fc2 (tulip.py:195-198), 32 hidden channels at a time, chained in registers
    f32x4 acc3[6];
#pragma unroll
    for (int n2 = 0; n2 < 6; ++n2) acc3[n2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int p = 0; p < 12; ++p) {
        IGLP();
        bf16x4 gp[2], hq[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * p + jj;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* wr = smem + OFF_A + (16 * j + t) * PW;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(wr, 32 * s + 4 * gq), x2frag[s], acc, 0, 0, 0);
            const int c0 = 16 * j + 4 * gq;
            const float4 bb = *(const float4*)(prm + P_B1 + c0);
            const bf16x4 hp = pack4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
            const f32x2 h01 = {bf2f((bf16_t)hp[0]), bf2f((bf16_t)hp[1])}, h23 = {bf2f((bf16_t)hp[2]), bf2f((bf16_t)hp[3])};
            if constexpr (SAVE == 3) {                                                           // GELU of the stored h + its derivative
                f32x2 g01, g23, d01, d23;
                gelu_exact_and_grad2(h01, g01, d01);
                gelu_exact_and_grad2(h23, g23, d23);
                gp[jj] = pack4(g01.x, g01.y, g23.x, g23.y);
                hq[jj] = pack4(d01.x, d01.y, d23.x, d23.y);
            } else {
                hq[jj] = hp;
                const f32x2 g01 = gelu_exact2(h01), g23 = gelu_exact2(h23);                      // GELU of the stored h
                gp[jj] = pack4(g01.x, g01.y, g23.x, g23.y);
            }
        }
        if constexpr (SAVE >= 2) store_bf16_tile_pair<true>(a.h + row * HID + 32 * p, hq[0], hq[1], gq);       // 16-byte stores (common.h)
        if constexpr (SAVE != 0) store_bf16_tile_pair<true>(a.g + row * HID + 32 * p, gp[0], gp[1], gq);
        const bf16x8 gf = cat8(gp[0], gp[1]);
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2)
            acc3[n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(smem + OFF_W2 + (16 * n2 + t) * PW2, 32 * p + 4 * gq), gf,
                                                               acc3[n2], 0, 0, 0);
    }
    TULIP_STAMP(8);
#pragma unroll
    for (int n2 = 0; n2 < 6; ++n2) {
        const int c0 = 16 * n2 + 4 * gq;
        const float4 bb = *(const float4*)(prm + P_B2 + c0);
        const f32x4 y = {xv[n2][0] + s1v * (acc3[n2][0] + bb.x), xv[n2][1] + s1v * (acc3[n2][1] + bb.y),
                         xv[n2][2] + s1v * (acc3[n2][2] + bb.z), xv[n2][3] + s1v * (acc3[n2][3] + bb.w)};
        if constexpr (HAND & 1) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xout, 0, 0x7FFFFFFF, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_h, y), rsrc, (unsigned)(row * C + c0) * 4u, 0, 16);
        } else {
            *(float4*)(a.xout + row * C + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
    TULIP_STAMP(9);
}

template <int SAVE, bool PROF>
__global__ __launch_bounds__(NT) void swin96_fwd_kernel(const Swin96Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    swin96_fwd_tile<SAVE, PROF, 0>(a, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Two blocks of a stage in ONE launch (tulip.py:399-436 runs them back to back: the un-shifted block, then the shifted one).
// A tile of the second block reads its own rows plus those of at most three neighbouring tiles of the first: no grid barrier --
// every workgroup publishes its first-block tile (sc1 stores, drained, then one flag per tile), polls the <= 4 flags of the tiles
// its second-block tile reads, and goes on.  First-block tiles never wait, so the launch cannot deadlock whatever is resident.
// sync: [0] epoch (a flag is "set" when it holds epoch + 1; the last workgroup to leave advances it: no clearing between graph
// replays), [1] workgroups that left, [2] / [3] dev: tiles of parity [2] - 1 idle [3] sleep quanta before their first block
// (forces the arrival order in the tests), [4] polls that gave up (stays 0), [16 + 4 tile] flags.
#ifndef TULIP_PAIR_HAND
#define TULIP_PAIR_HAND 3              // dev (tools/bench_pair96.py): 0 = plain loads / stores, with TULIP_PAIR_NOSYNC=1 the bare cost of the form
#endif
#ifndef TULIP_PAIR_NOSYNC
#define TULIP_PAIR_NOSYNC 0
#endif
struct Swin96Pair { Swin96Args a0, a1; unsigned* sync; int ntiles; };
__device__ __forceinline__ int pair_source_tile(int H, int W, int dsh, int dsw, int tile, int corner) {
    const int nWy = H >> 1, gpr = (W >> 3) / NW;
    const int b = tile / (nWy * gpr);
    const int r = tile - b * nWy * gpr, wy = r / gpr, g = r - wy * gpr;
    int hh = 2 * wy + (corner & 1) + dsh, ww = 64 * g + 63 * (corner >> 1) + dsw;
    hh = (hh % H + H) % H;
    ww = (ww % W + W) % W;
    return (b * nWy + (hh >> 1)) * gpr + (ww >> 6);
}
// One tile per workgroup (the launch is one round of the chip), straight-line: first block, publish, poll, second block.  The second
// block's arguments are read through a pointer the compiler cannot see through until the first block is over -- hoisted to the top,
// the two argument sets (2 x 70 scalar registers) spilled into vector lanes and came back through ~950 v_readlane (+5 us per block).
typedef const __attribute__((address_space(4))) Swin96Args KSwin96Args;
template <int SAVE>
__global__ __launch_bounds__(NT) void swin96_pair_fwd_kernel(const Swin96Pair P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const unsigned want = __hip_atomic_load(P.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const unsigned hold = __hip_atomic_load(P.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hold && (unsigned)(tile & 1) == hold - 1u) {
        const unsigned n = __hip_atomic_load(P.sync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64);
    }
    swin96_fwd_tile<SAVE, false, TULIP_PAIR_HAND & 1>(P.a0, tile, smem);
    if (!TULIP_PAIR_NOSYNC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
    __syncthreads();
    // (the kernel's arguments start at offset 0 of the kernarg segment; &P.a1 would make the compiler copy P to scratch)
    KSwin96Args* a1 = (KSwin96Args*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Swin96Pair, a1));
    unsigned* sync = P.sync;
    asm volatile("" : "+s"(a1), "+s"(sync) :: "memory");
    if (!TULIP_PAIR_NOSYNC) {
        if (tid == 0) __hip_atomic_store(sync + 16 + 4 * tile, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 4) {
            const unsigned* f = sync + 16 + 4 * pair_source_tile(a1->H, a1->W, a1->sh - P.a0.sh, a1->sw - P.a0.sw, tile, tid);
            // (bounded: ~0.5 s of polling, then on with whatever is there -- a wrong result a test can see instead of a hung queue)
            unsigned spin = 0;
            for (; spin < (1u << 24) && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want; ++spin)
                __builtin_amdgcn_s_sleep(1);
            if (spin == (1u << 24)) __hip_atomic_fetch_add(sync + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // gave up: the host can see it
        }
        __syncthreads();
    }
    swin96_fwd_tile<SAVE, false, TULIP_PAIR_HAND & 2>(*a1, tile, smem);
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gridDim.x - 1) {
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// =====================================================================================================================
// backward of the same block in one launch: the whole chain fc2' -> GELU' -> fc1' -> norm2' -> proj' -> attention' ->
// qkv' -> norm1' stays in registers, one window per wave.  The four data-gradient GEMMs contract over the OUTPUT
// channels of their Linear, i.e. they need W^T as the MFMA A operand: the weights sit in LDS in their natural
// [out][in] layout and are read through ds_read_b64_tr_b16 (rows padded to 32 B x odd so that the 16 x 32-byte
// blocks of a transpose read are bank-conflict free).  What leaves the kernel: the input gradient (fp32, in place),
// the bf16 operands of the four weight-gradient GEMMs (which run beside the chain as before), and one partial row per
// workgroup for the LayerNorm affine gradients and the dense relative-position-bias gradient.
constexpr int PT = 224;                          // 96-wide bf16 row read transposed
constexpr int PT2 = 800;                         // 384-wide bf16 row read transposed
constexpr int BOFF_W2 = 0;                       // MLP half: fc2.weight [96][384] | fc1.weight [384][96]
constexpr int BOFF_W1 = 96 * PT2;
constexpr int BOFF_WQKV = 0;                     // attention half: qkv.weight [288][96] | proj.weight [96][96]
constexpr int BOFF_WPROJ = 288 * PT;
constexpr int BOFF_T = BOFF_WPROJ + 96 * PT;     // NW x (Q | K | dO) 1-KiB tiles
constexpr int BOFF_RED = BOFF_T + NW * 3072;     // fp32 [NW][192] norm2 | [NW][192] norm1 | [NW][768] bias partial sums
constexpr int BOFF_GAM = BOFF_W1 + 384 * PT;     // 162816: norm2.weight[96] norm1.weight[96] fp32 (beyond both layouts)
constexpr int BSMEM = BOFF_GAM + 2 * C * 4;
#ifndef TULIP_SWIN96_PREFETCH
#define TULIP_SWIN96_PREFETCH 0         // measured: +6.5 us per launch warm, L2-cold and all-cold alike (profiles/README.md): off
#endif
constexpr int BOFF_BQKV = BOFF_RED + NW * (192 + 192 + 768) * 4;   // qkv.bias[288] fp32 (attention half, recomputation form)
static_assert(BOFF_BQKV + 288 * 4 <= BOFF_GAM, "attention-half layout overlaps the norm weights");
static_assert(BSMEM <= 163840, "LDS");

struct Swin96BwdArgs {
    float* dx;                                   // in: d(block output); out: d(block input)
    const float *xin, *x1;
    const bf16_t *qkv, *h;
    const float *mean1, *rstd1, *mean2, *rstd2;
    const bf16_t *wqkv, *wproj, *w1, *w2;
    const float *g1, *g2;
    const float *bqkv, *b1, *be1, *be2;          // recomputation form (qkv == h == nullptr): the biases the forward added
    const float* bias_table; const int* rel_index;
    const float *ds0, *ds1;
    bf16_t *dyb_m, *dh, *dyb_a, *dqkv;           // bf16 operands of the fc2 / fc1 / proj / qkv weight gradients
    bf16_t* dx_bf16; const float* dx_scale;      // optional bf16(dx * per-sample scale) for the consumer of dx
    float *lnpart1, *lnpart2, *biaspart;         // [workgroups][192], [workgroups][192], [workgroups][768]
    int B, H, W, sh, sw, masked;
    float scale;
    unsigned long long* prof;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 trr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)p);
}
// A fragment of W^T in the chained k order from W stored [k][n]: lane (t, gq) gets W[k0 + 4gq + e][n0 + t] in k slots
// e = 0..3 and W[k0 + 16 + 4gq + e][n0 + t] in slots 4..7
__device__ __forceinline__ bf16x8 wfrag_t(const unsigned char* w, int pitch, int k0, int n0, int t, int gq) {
    const unsigned char* p = w + (k0 + 4 * gq + (t >> 2)) * pitch + (n0 + (t & 3) * 4) * 2;
    return cat8(trr(p), trr(p + 16 * pitch));
}
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int ROWS, int COLS>
struct Staged { u32x4 v[(ROWS * COLS / 8 + NT - 1) / NT]; };
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_load(const bf16_t* __restrict__ w, Staged<ROWS, COLS>& st, int tid) {
    constexpr int CPR = COLS / 8, N = ROWS * CPR, PER = (N + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = (N % NT == 0 || tid + i * NT < N) ? tid + i * NT : N - 1;     // tail threads re-read the last chunk
        st.v[i] = *(const u32x4*)(w + (size_t)(c / CPR) * COLS + (c % CPR) * 8);
    }
}
template <int ROWS, int COLS, int PITCH>
__device__ __forceinline__ void stage_store(const Staged<ROWS, COLS>& st, unsigned char* dst, int tid) {
    constexpr int CPR = COLS / 8, N = ROWS * CPR, PER = (N + NT - 1) / NT;
    static_assert(PITCH % 16 == 0, "16-byte LDS stores");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * NT;
        if (N % NT == 0 || c < N) *(u32x4*)(dst + (c / CPR) * PITCH + (c % CPR) * 16) = st.v[i];
    }
}
// one halving step of the 16-lane reduce-scatter: after the steps with m = 8, 4, 2, 1 on 48 values, lane t holds the
// sums over the 16 token lanes of values 3t .. 3t+2 in v[0..2].  The partner of a step only has to sit in the other half of
// the 2m-lane group (each lane keeps the half its OWN bit m selects, so any perfect matching across the halves gives the
// same sums): row_mirror / row_half_mirror / quad_perm are modifiers of a vector-ALU move, where lane ^ m through
// __shfl_xor was a ds_bpermute round trip per value (45 per LayerNorm backward, two per block)
template <int HALF, int M>
__device__ __forceinline__ void halve(float (&v)[48], int t) {
    const bool up = (t & M) != 0;
    constexpr int CTRL = M == 8 ? 0x140 : M == 4 ? 0x141 : M == 2 ? 0x4E : 0xB1;   // row_mirror, row_half_mirror, quad [2,3,0,1], [1,0,3,2]
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float a = v[i], b = v[HALF + i];
        v[i] = (up ? b : a) + dpp_move<CTRL>(up ? a : b);
    }
}
// LayerNorm backward of one token row held as 6 x 4 channels (16n + 4gq + r): d <- rstd*(d*gamma - m1 - xhat*m2);
// the affine-gradient terms d*xhat | d of the 16 tokens of the wave are reduce-scattered into red[0..2]
__device__ __forceinline__ void ln_bwd_row(f32x4 (&d)[6], const f32x4 (&xv)[6], float mu, float rs, const float* gam,
                                           int t, int gq, float (&red)[3]) {
    float v[48];
    float s1 = 0.f, s2 = 0.f;
    f32x4 xh[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const float4 ga = *(const float4*)(gam + 16 * n + 4 * gq);
        const float gv[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[n][r] = (xv[n][r] - mu) * rs;
            v[4 * n + r] = d[n][r] * xh[n][r];
            v[24 + 4 * n + r] = d[n][r];
            d[n][r] *= gv[r];
            s1 += d[n][r];
            s2 += d[n][r] * xh[n][r];
        }
    }
    s1 = rows_sum(s1);
    s2 = rows_sum(s2);
    const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[n][r] = rs * (d[n][r] - m1 - xh[n][r] * m2);
    halve<24, 8>(v, t); halve<12, 4>(v, t); halve<6, 2>(v, t); halve<3, 1>(v, t);
    red[0] = v[0]; red[1] = v[1]; red[2] = v[2];
}
// value q = 3t + i of the reduce-scatter is [d*xhat | d][channel 16n + 4gq + r] with q = 24*which + 4n + r
__device__ __forceinline__ void put_red(float* row, const float (&red)[3], int t, int gq) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = 3 * t + i, which = q / 24, n = (q % 24) >> 2, r = q & 3;
        row[which * C + 16 * n + 4 * gq + r] = red[i];
    }
}

// RECOMP (round 4): qkv and the fc1 pre-activation h are not read -- the forward no longer writes them (swin96_fwd_kernel<1>) --
// but recomputed from x / x1, the saved statistics and the weights that are in LDS anyway: norm1 -> qkv (54 MFMAs per window)
// and norm2 -> fc1 (72) with the forward's own operand fragments, accumulation order and bias adds, i.e. the forward's bits.
// The [out][in] LDS copy that the data-gradient GEMMs read transposed is read PLAIN for it (two 8-byte reads per fragment; with
// the 32 B x odd pitch of the transpose reads tokens t and t + 8 share a bank: 2-way).  2.7 KB per token less HBM traffic
// for the forward / backward pair, 126 more MFMAs per window.
// HGRAD (round 4): the fc1_pre buffer holds bf16(gelu'(h)), written by swin96_fwd_kernel<3> (TULIP_BLOCK_FC1_GRAD) -- the MLP
// half then multiplies with it instead of evaluating erf / exp for 384 values per token (100 of the loop's 134 vector
// instructions per 32 hidden channels).
template <bool RECOMP, bool HGRAD, bool PROF>
__global__ __launch_bounds__(NT) void swin96_bwd_kernel(const Swin96BwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[BSMEM];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    float* gam = (float*)(smem + BOFF_GAM);

    const int nWx = a.W >> 3, nWy = a.H >> 1, gpr = nWx / NW;
    int blk = blockIdx.x;
    const int b = blk / (nWy * gpr);
    blk -= b * nWy * gpr;
    const int wy = blk / gpr, wx = (blk - wy * gpr) * NW + wid;
    const int hs = wy * 2 + (t >> 3), ws = wx * 8 + (t & 7);
    int hh = hs + a.sh; if (hh >= a.H) hh -= a.H;
    int ww = ws + a.sw; if (ww >= a.W) ww -= a.W;
    const size_t row = ((size_t)b * a.H + hh) * a.W + ww;
    const int lab = 3 * region(hs, a.H, 2, a.sh) + region(ws, a.W, 8, a.sw);
    const float s0 = a.ds0 ? a.ds0[b] : 1.0f, s1v = a.ds1 ? a.ds1[b] : 1.0f;
    TULIP_STAMP(0);

    // ---- stage fc2 / fc1 weights.  Every load of the prologue -- weights, the incoming gradient row, statistics, the norm
    // weights, the relative-position index -- is issued before the first wait (round 4: they were four dependent round trips)
    f32x4 dy[6], x1v[6];
    int ridx_q[4], ridx_k[4];
    float mu2, rs2;
    unsigned pf[5] = {0u, 0u, 0u, 0u, 0u};
    {
        Staged<C, HID> w2s; Staged<HID, C> w1s;
        stage_load<C, HID>(a.w2, w2s, tid);
        stage_load<HID, C>(a.w1, w1s, tid);
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const float4 u = *(const float4*)(a.dx + row * C + 16 * n + 4 * gq);
            dy[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
        mu2 = a.mean2[row]; rs2 = a.rstd2[row];
        if constexpr (HGRAD && !RECOMP) {     // the norm2 input row now: behind the (short) MLP loop its loads would queue behind the dh stores
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                const float4 u = *(const float4*)(a.x1 + row * C + 16 * n + 4 * gq);
                x1v[n] = (f32x4){u.x, u.y, u.z, u.w};
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ridx_q[r] = a.rel_index[t * 16 + gq * 4 + r]; ridx_k[r] = a.rel_index[(gq * 4 + r) * 16 + t]; }
        float gv = 0.f;
        if (tid < 2 * C) gv = tid < C ? a.g2[tid] : a.g1[tid - C];
#if TULIP_SWIN96_PREFETCH
        // In the training step everything the forward saved for this launch was written a millisecond and ~2 GB of traffic
        // ago: every read of it is an HBM miss (isolated: 35 us warm, 54 us behind a 1-GB fill = the in-step figure), and the
        // MLP loop below fetches its gelu'(h) fragments only one iteration (~1.3 k cycles) ahead.  The rows of this wave's 16
        // tokens are touched here, behind the prologue's own loads: one 4-byte load per 128-byte line (6 lines of h, 4.5 of
        // qkv, 3 of x per token, the norm1 statistics) into registers that are "used" at the barrier below.  (The same touches
        // as LDS-destination loads -- no registers -- cost 6 us per launch: every LDS access behind them waits for vmcnt(0).)
        if constexpr (!RECOMP) {
            const unsigned char* hb = (const unsigned char*)(a.h + row * HID);
            const unsigned char* qb = (const unsigned char*)(a.qkv + row * 288);
            pf[0] = *(const unsigned*)(hb + 128 * gq);
            pf[1] = *(const unsigned*)(hb + 512 + 128 * (gq & 1));
            pf[2] = *(const unsigned*)(qb + 128 * gq);
            pf[3] = *(const unsigned*)(gq < 2 ? qb + 512 + 60 * gq : (const unsigned char*)(gq == 2 ? a.mean1 + row : a.rstd1 + row));
            pf[4] = *(const unsigned*)((const unsigned char*)(a.xin + row * C) + 128 * (gq < 2 ? gq : 2));
        }
#endif
        stage_store<C, HID, PT2>(w2s, smem + BOFF_W2, tid);
        stage_store<HID, C, PT>(w1s, smem + BOFF_W1, tid);
        if (tid < 2 * C) gam[tid] = gv;
    }
    // relative-position bias seen from the query side (query t, key 4gq+r) and from the key side (query 4gq+r, key t)
    float bias_q[3][4], bias_k[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int eq = ridx_q[r] * 3, ek = ridx_k[r] * 3;
#pragma unroll
        for (int h = 0; h < 3; ++h) { bias_q[h][r] = a.bias_table[eq + h]; bias_k[h][r] = a.bias_table[ek + h]; }
    }
    bf16x8 dyf[3];
    {
        bf16x4 pk[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            pk[n] = pack4(dy[n][0] * s1v, dy[n][1] * s1v, dy[n][2] * s1v, dy[n][3] * s1v);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            store_bf16_tile_pair<true>(a.dyb_m + row * C + 32 * s, pk[2 * s], pk[2 * s + 1], gq);
            dyf[s] = cat8(pk[2 * s], pk[2 * s + 1]);
        }
    }
    // recomputation form: xn2 = norm2(x1) as the forward formed it (the row is fetched again for norm2' behind the loop: L2)
    bf16x8 x2frag[3];
    if constexpr (RECOMP) {
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const float4 u = *(const float4*)(a.x1 + row * C + 16 * n + 4 * gq);
            x1v[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
        bf16x4 p2[6];
        ln_apply(x1v, mu2, rs2, a.g2, a.be2, gq, p2);
#pragma unroll
        for (int s = 0; s < 3; ++s) x2frag[s] = cat8(p2[2 * s], p2[2 * s + 1]);
    }
    TULIP_STAMP(1);
#if TULIP_SWIN96_PREFETCH
    asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]));
#endif
    __syncthreads();
    TULIP_STAMP(2);

    // ---- MLP half (tulip.py:346-351 backwards): per 32 hidden channels  dg = dy.W2 -> dh = dg*gelu'(h) -> dxn2 += dh.W1
    f32x4 acc2[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) acc2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* hrow = RECOMP ? nullptr : a.h + row * HID + 4 * gq;
    bf16x4 hn[2];
    float4 bn[2];
    if constexpr (RECOMP) { bn[0] = *(const float4*)(a.b1 + 4 * gq); bn[1] = *(const float4*)(a.b1 + 16 + 4 * gq); }
    else { hn[0] = *(const bf16x4*)(hrow); hn[1] = *(const bf16x4*)(hrow + 16); }
#pragma unroll 2
    for (int p = 0; p < 12; ++p) {
        IGLP();
        bf16x4 hc[2];
        float4 bc[2];
        if constexpr (RECOMP) {
            bc[0] = bn[0]; bc[1] = bn[1];
            if (p + 1 < 12) { bn[0] = *(const float4*)(a.b1 + 32 * (p + 1) + 4 * gq); bn[1] = *(const float4*)(a.b1 + 32 * (p + 1) + 16 + 4 * gq); }
        } else {
            hc[0] = hn[0]; hc[1] = hn[1];
            if (p + 1 < 12) { hn[0] = *(const bf16x4*)(hrow + 32 * (p + 1)); hn[1] = *(const bf16x4*)(hrow + 32 * (p + 1) + 16); }
        }
        bf16x4 dp[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j0 = 32 * p + 16 * jj;
            if constexpr (RECOMP) {     // h tile = fc1(xn2) + bias, rounded to bf16: swin96_fwd_kernel's fc1 step
                f32x4 ah = {0.f, 0.f, 0.f, 0.f};
                const unsigned char* wr = smem + BOFF_W1 + (j0 + t) * PT;
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(wr, 32 * s + 4 * gq), x2frag[s], ah, 0, 0, 0);
                hc[jj] = pack4(ah[0] + bc[jj].x, ah[1] + bc[jj].y, ah[2] + bc[jj].z, ah[3] + bc[jj].w);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag_t(smem + BOFF_W2, PT2, 32 * s, j0, t, gq), dyf[s], acc, 0, 0, 0);
            f32x2 d01 = {bf2f((bf16_t)hc[jj][0]), bf2f((bf16_t)hc[jj][1])}, d23 = {bf2f((bf16_t)hc[jj][2]), bf2f((bf16_t)hc[jj][3])};
            if constexpr (!HGRAD || RECOMP) { d01 = gelu_exact_grad2(d01); d23 = gelu_exact_grad2(d23); }
            dp[jj] = pack4(acc[0] * d01.x, acc[1] * d01.y, acc[2] * d23.x, acc[3] * d23.y);
        }
        store_bf16_tile_pair<true>(a.dh + row * HID + 32 * p, dp[0], dp[1], gq);
        const bf16x8 df = cat8(dp[0], dp[1]);
#pragma unroll
        for (int n = 0; n < 6; ++n)
            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag_t(smem + BOFF_W1, PT, 32 * p, 16 * n, t, gq), df, acc2[n], 0, 0, 0);
    }

    TULIP_STAMP(3);
    // ---- the attention-half weights are fetched while norm2 is differentiated
    Staged<288, C> wqs; Staged<C, C> wps;
    stage_load<288, C>(a.wqkv, wqs, tid);
    stage_load<C, C>(a.wproj, wps, tid);
    if constexpr (!HGRAD || RECOMP) {
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const float4 u = *(const float4*)(a.x1 + row * C + 16 * n + 4 * gq);
            x1v[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
    }
    if constexpr (RECOMP) {     // the incoming gradient row again (L2): 24 registers that need not live through the loop above
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const float4 u = *(const float4*)(a.dx + row * C + 16 * n + 4 * gq);
            dy[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
    }
    float red2[3];
    ln_bwd_row(acc2, x1v, mu2, rs2, gam, t, gq, red2);
    // dx1 = d(x1) = dy + norm2'(dxn2)   (residual, tulip.py:351); from here on dy holds dx1
    bf16x8 daf[3];
    {
        bf16x4 pk[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            dy[n] += acc2[n];
            pk[n] = pack4(dy[n][0] * s0, dy[n][1] * s0, dy[n][2] * s0, dy[n][3] * s0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            store_bf16_tile_pair<true>(a.dyb_a + row * C + 32 * s, pk[2 * s], pk[2 * s + 1], gq);
            daf[s] = cat8(pk[2 * s], pk[2 * s + 1]);
        }
    }
    // q, k, v of this token, all heads, in the chained k order (dims 4gq.., 16+4gq.. of each head): read back, or (RECOMP)
    // recomputed behind the barrier from xn1 = norm1(x) as the forward formed it
    bf16x4 qkvr[18];
    bf16x8 xfrag[3];
    const float mu1 = a.mean1[row], rs1 = a.rstd1[row];
    if constexpr (!RECOMP) {
#pragma unroll
        for (int j = 0; j < 18; ++j) qkvr[j] = *(const bf16x4*)(a.qkv + row * 288 + 16 * j + 4 * gq);
    }
    TULIP_STAMP(4);
    __syncthreads();                                          // every wave is done with fc1 / fc2 weights
    TULIP_STAMP(5);
    stage_store<288, C, PT>(wqs, smem + BOFF_WQKV, tid);
    stage_store<C, C, PT>(wps, smem + BOFF_WPROJ, tid);
    float* redw = (float*)(smem + BOFF_RED);
    put_red(redw + wid * 192, red2, t, gq);
    if constexpr (RECOMP) {     // (the x row is fetched here, not in front of the barrier: the staged weights hold 40 registers there)
        if (tid < 288) ((float*)(smem + BOFF_BQKV))[tid] = a.bqkv[tid];
        f32x4 xr[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const float4 u = *(const float4*)(a.xin + row * C + 16 * n + 4 * gq);
            xr[n] = (f32x4){u.x, u.y, u.z, u.w};
        }
        bf16x4 p1[6];
        ln_apply(xr, mu1, rs1, a.g1, a.be1, gq, p1);
#pragma unroll
        for (int s = 0; s < 3; ++s) xfrag[s] = cat8(p1[2 * s], p1[2 * s + 1]);
    }
    __syncthreads();
    TULIP_STAMP(6);

    IGLP();
    if constexpr (RECOMP) {     // qkv = xn1 . Wqkv^T + bias, rounded to bf16: swin96_fwd_kernel's qkv step
        const float* bq = (const float*)(smem + BOFF_BQKV);
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* wr = smem + BOFF_WQKV + (16 * j + t) * PT;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(wr, 32 * s + 4 * gq), xfrag[s], acc, 0, 0, 0);
            const float4 b4 = *(const float4*)(bq + 16 * j + 4 * gq);
            qkvr[j] = pack4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w);
        }
    }

    TULIP_STAMP(7);
    // ---- proj' : dO = dyb_a . Wproj   (tulip.py:318 backwards)
    bf16x4 dop[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag_t(smem + BOFF_WPROJ, PT, 32 * s, 16 * n, t, gq), daf[s], acc, 0, 0, 0);
        dop[n] = pack4(acc[0], acc[1], acc[2], acc[3]);
    }
    TULIP_STAMP(8);
    // ---- attention' per head (tulip.py:300-317 backwards; same algebra as attn_bwd_kernel)
    unsigned char* ldsQ = smem + BOFF_T + wid * 3072;
    unsigned char* ldsK = ldsQ + 1024;
    unsigned char* ldsD = ldsK + 1024;
    float* redb = redw + NW * 384 + wid * 768;
    bf16x4 dqkvp[18];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int troff = (gq * 4 + (t >> 2)) * 64 + (t & 3) * 8;
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        bf16x8 qf = cat8(qkvr[2 * h], qkvr[2 * h + 1]);
        bf16x8 kf = cat8(qkvr[6 + 2 * h], qkvr[7 + 2 * h]);
        if (a.masked & TULIP_ATTN_FP8) { qf = round_through_fp8(qf); kf = round_through_fp8(kf); }   // what the forward's scores saw
        const bf16x8 vf = cat8(qkvr[12 + 2 * h], qkvr[13 + 2 * h]);
        const bf16x8 df = cat8(dop[2 * h], dop[2 * h + 1]);
        const int o0 = t * 64 + (4 * gq) * 2, o1 = t * 64 + (16 + 4 * gq) * 2;
        *(bf16x4*)(ldsQ + o0) = lo4(qf);  *(bf16x4*)(ldsQ + o1) = hi4(qf);
        *(bf16x4*)(ldsK + o0) = lo4(kf);  *(bf16x4*)(ldsK + o1) = hi4(kf);
        *(bf16x4*)(ldsD + o0) = dop[2 * h];       *(bf16x4*)(ldsD + o1) = dop[2 * h + 1];
        f32x4 sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);    // S[t][4gq+r]
        f32x4 sk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);    // S[4gq+r][t]
        f32x4 dpq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df, z, 0, 0, 0);   // dP[t][4gq+r]
        f32x4 dpk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, z, 0, 0, 0);   // dP[4gq+r][t]
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float xq = sq[r] * a.scale + bias_q[h][r];
            float xk = sk[r] * a.scale + bias_k[h][r];
            if (a.masked & TULIP_ATTN_MASKED) {
                const int ol = __shfl(lab, gq * 4 + r, 64);
                if (ol != lab) { xq += -100.0f; xk += -100.0f; }
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
        }
        *(float4*)(redb + h * 256 + t * 16 + gq * 4) = make_float4(dsq[0], dsq[1], dsq[2], dsq[3]);
        const bf16x4 dsq_b = pack4(dsq[0], dsq[1], dsq[2], dsq[3]);
        const bf16x4 dsk_b = pack4(dsk[0], dsk[1], dsk[2], dsk[3]);
        const bf16x4 pk_b = pack4(pk[0], pk[1], pk[2], pk[3]);
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const bf16x4 kt = trr(ldsK + troff + dc * 32);     // K[4gq+e][16dc+t]
            const bf16x4 qt = trr(ldsQ + troff + dc * 32);
            const bf16x4 dt = trr(ldsD + troff + dc * 32);
            const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, dsq_b, z, 0, 0, 0);   // dQ[t][16dc+4gq+r] / scale
            const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dsk_b, z, 0, 0, 0);
            const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dt, pk_b, z, 0, 0, 0);
            dqkvp[2 * h + dc] = pack4(dq[0] * a.scale, dq[1] * a.scale, dq[2] * a.scale, dq[3] * a.scale);
            dqkvp[6 + 2 * h + dc] = pack4(dk[0] * a.scale, dk[1] * a.scale, dk[2] * a.scale, dk[3] * a.scale);
            dqkvp[12 + 2 * h + dc] = pack4(dv[0], dv[1], dv[2], dv[3]);
        }
    }
    TULIP_STAMP(9);
#pragma unroll
    for (int j = 0; j < 9; ++j) store_bf16_tile_pair<true>(a.dqkv + row * 288 + 32 * j, dqkvp[2 * j], dqkvp[2 * j + 1], gq);

    // ---- qkv' : dxn1 = dqkv . Wqkv  (tulip.py:298 backwards), then norm1' and the residual
    f32x4 acc1[6], xv[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const float4 u = *(const float4*)(a.xin + row * C + 16 * n + 4 * gq);
        xv[n] = (f32x4){u.x, u.y, u.z, u.w};
        acc1[n] = z;
    }
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const bf16x8 f = cat8(dqkvp[2 * s], dqkvp[2 * s + 1]);
#pragma unroll
        for (int n = 0; n < 6; ++n)
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag_t(smem + BOFF_WQKV, PT, 32 * s, 16 * n, t, gq), f, acc1[n], 0, 0, 0);
    }
    TULIP_STAMP(10);
    float red1[3];
    ln_bwd_row(acc1, xv, mu1, rs1, gam + C, t, gq, red1);
    const float cs = a.dx_scale ? a.dx_scale[b] : 1.0f;
    bf16x4 ob[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const f32x4 o = dy[n] + acc1[n];
        *(float4*)(a.dx + row * C + 16 * n + 4 * gq) = make_float4(o[0], o[1], o[2], o[3]);
        ob[n] = pack4(o[0] * cs, o[1] * cs, o[2] * cs, o[3] * cs);
    }
    if (a.dx_bf16) {
#pragma unroll
        for (int n = 0; n < 3; ++n) store_bf16_tile_pair(a.dx_bf16 + row * C + 32 * n, ob[2 * n], ob[2 * n + 1], gq);
    }
    put_red(redw + NW * 192 + wid * 192, red1, t, gq);
    TULIP_STAMP(11);
    __syncthreads();
    // ---- one partial row per workgroup: waves summed in a fixed order
    for (int i = tid; i < 192 + 192 + 768; i += NT) {
        const float* src; float* dst; int stride;
        if (i < 192) { src = redw + i; stride = 192; dst = a.lnpart2 + (size_t)blockIdx.x * 192 + i; }
        else if (i < 384) { src = redw + NW * 192 + (i - 192); stride = 192; dst = a.lnpart1 + (size_t)blockIdx.x * 192 + (i - 192); }
        else { src = redw + NW * 384 + (i - 384); stride = 768; dst = a.biaspart + (size_t)blockIdx.x * 768 + (i - 384); }
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += src[w * stride];
        *dst = sacc;
    }
    TULIP_STAMP(12);
}

}  // namespace

// descriptor -> kernel arguments; the form of the launch: -1 argument error, 0 inference, 1 lean (recomputing backward), 2 everything
// saved, 3 everything with gelu'(h) in fc1_pre
static int swin96_fwd_args(const tulip_swin96_desc* d, Swin96Args& a) {
    if (!d || d->B <= 0 || d->H <= 0 || (d->H & 1) || d->W <= 0 || (d->W & 63) || d->shift_h < 0 ||
        d->shift_h >= d->H || d->shift_w < 0 || d->shift_w >= d->W)
        return -1;
    a.xin = d->x_in; a.x1 = d->x1; a.xout = d->x_out;
    a.xn1 = (bf16_t*)d->xn1; a.qkv = (bf16_t*)d->qkv; a.o = (bf16_t*)d->attn_out; a.xn2 = (bf16_t*)d->xn2;
    a.h = (bf16_t*)d->fc1_pre; a.g = (bf16_t*)d->fc1_act;
    a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.wqkv = (const bf16_t*)d->w_qkv; a.wproj = (const bf16_t*)d->w_proj; a.w1 = (const bf16_t*)d->w_fc1;
    a.w2 = (const bf16_t*)d->w_fc2;
    a.bqkv = d->b_qkv; a.bproj = d->b_proj; a.b1 = d->b_fc1; a.b2 = d->b_fc2;
    a.g1 = d->norm1_weight; a.be1 = d->norm1_bias; a.g2 = d->norm2_weight; a.be2 = d->norm2_bias;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.B = d->B; a.H = d->H; a.W = d->W; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.eps = d->eps; a.scale = 0.17677669529663687f;        // head_dim^-0.5 = 32^-0.5 (tulip.py:220)
    a.prof = nullptr;
    // every saved-activation pointer NULL: the inference form; qkv and fc1_pre alone NULL: the backward recomputes them
    const bool core = d->xn1 && d->attn_out && d->x1 && d->xn2 && d->fc1_act && d->mean1 && d->rstd1 && d->mean2 && d->rstd2;
    const bool any = d->xn1 || d->qkv || d->attn_out || d->x1 || d->xn2 || d->fc1_pre || d->fc1_act || d->mean1 || d->rstd1 ||
                     d->mean2 || d->rstd2;
    if (any && !(core && (!d->qkv == !d->fc1_pre))) return -1;
    if (!any) return 0;
    if (!d->qkv) return 1;
    return (d->masked & TULIP_BLOCK_FC1_GRAD) ? 3 : 2;
}

static int swin96_fwd_impl(const tulip_swin96_desc* d, unsigned long long* prof, hipStream_t stream) {
    Swin96Args a;
    const int form = swin96_fwd_args(d, a);
    if (form < 0) return TULIP_ERR_ARG;
    a.prof = prof;
    const int blocks = d->B * (d->H / 2) * (d->W / (8 * NW));
    const bool any = form != 0, hgrad = form == 3;
    const dim3 g(blocks), b(NT);
#if TULIP_DEV_VARIANTS
    if (prof) {                                         // diagnostic twin: the full training forms only
        if (!any || !d->qkv) return TULIP_ERR_ARG;
        if (hgrad) hipLaunchKernelGGL((swin96_fwd_kernel<3, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((swin96_fwd_kernel<2, true>), g, b, 0, stream, a);
    } else if (any && !d->qkv) hipLaunchKernelGGL((swin96_fwd_kernel<1, false>), g, b, 0, stream, a);    // the lean form of the recomputing backward
    else
#else
    if (prof || (any && !d->qkv)) return TULIP_ERR_NOT_BUILT;
#endif
    if (!any) hipLaunchKernelGGL((swin96_fwd_kernel<0, false>), g, b, 0, stream, a);
    else if (hgrad) hipLaunchKernelGGL((swin96_fwd_kernel<3, false>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((swin96_fwd_kernel<2, false>), g, b, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_swin96_block_fwd(const tulip_swin96_desc* d, hipStream_t stream) { return swin96_fwd_impl(d, nullptr, stream); }

// two blocks, one launch (swin96_pair_fwd_kernel)
extern "C" int tulip_swin96_pair_sync_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || (H & 1) || W <= 0 || (W & 63)) return 0;
    return (16 + 4 * B * (H / 2) * (W / (8 * NW))) * (int)sizeof(unsigned);
}
extern "C" int tulip_swin96_pair_fwd(const tulip_swin96_desc* d0, const tulip_swin96_desc* d1, void* sync, size_t sync_bytes,
                                     hipStream_t stream) {
    Swin96Pair P;
    const int f0 = swin96_fwd_args(d0, P.a0), f1 = swin96_fwd_args(d1, P.a1);
    if (f0 < 0 || f1 < 0 || f0 != f1 || !sync) return TULIP_ERR_ARG;
    if (d0->B != d1->B || d0->H != d1->H || d0->W != d1->W || d0->x_out != d1->x_in) return TULIP_ERR_ARG;
    if (sync_bytes < (size_t)tulip_swin96_pair_sync_bytes(d0->B, d0->H, d0->W) || ((uintptr_t)sync & 15)) return TULIP_ERR_ARG;
    if (f0 != 0 && f0 != 3) return TULIP_ERR_NOT_BUILT;      // the two forms the engine launches: inference, training with gelu'(h)
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return TULIP_ERR_LAUNCH;
    }
    P.sync = (unsigned*)sync;
    P.ntiles = d0->B * (d0->H / 2) * (d0->W / (8 * NW));
    if (P.ntiles > cus) return TULIP_ERR_ARG;                  // one workgroup per CU (LDS) and every one of them resident: a polling
    const dim3 g(P.ntiles), b(NT);                             // workgroup must never hold the CU its producer is waiting for
    if (f0 == 0) hipLaunchKernelGGL((swin96_pair_fwd_kernel<0>), g, b, 0, stream, P);
    else hipLaunchKernelGGL((swin96_pair_fwd_kernel<3>), g, b, 0, stream, P);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
extern "C" int tulip_swin96_block_fwd_profiled(const tulip_swin96_desc* d, uint64_t* stamps, hipStream_t stream) {
    return swin96_fwd_impl(d, (unsigned long long*)stamps, stream);
}

extern "C" int tulip_swin96_bwd_partial_rows(int B, int H, int W) {
    if (B <= 0 || H <= 0 || (H & 1) || W <= 0 || (W & 63)) return 0;
    return B * (H / 2) * (W / (8 * NW));
}

static int swin96_bwd_impl(const tulip_swin96_bwd_desc* d, unsigned long long* prof, hipStream_t stream) {
    if (!d || d->B <= 0 || d->H <= 0 || (d->H & 1) || d->W <= 0 || (d->W & 63) || d->shift_h < 0 ||
        d->shift_h >= d->H || d->shift_w < 0 || d->shift_w >= d->W)
        return TULIP_ERR_ARG;
    Swin96BwdArgs a;
    a.dx = d->dx; a.xin = d->x_in; a.x1 = d->x1;
    a.qkv = (const bf16_t*)d->qkv; a.h = (const bf16_t*)d->fc1_pre;
    a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.wqkv = (const bf16_t*)d->w_qkv; a.wproj = (const bf16_t*)d->w_proj; a.w1 = (const bf16_t*)d->w_fc1;
    a.w2 = (const bf16_t*)d->w_fc2;
    a.g1 = d->norm1_weight; a.g2 = d->norm2_weight;
    a.bqkv = d->b_qkv; a.b1 = d->b_fc1; a.be1 = d->norm1_bias; a.be2 = d->norm2_bias;
    // qkv and fc1_pre both NULL: the recomputation form (needs the four bias vectors the forward added)
    const bool recomp = !d->qkv && !d->fc1_pre;
    if (recomp ? !(d->b_qkv && d->b_fc1 && d->norm1_bias && d->norm2_bias) : !(d->qkv && d->fc1_pre)) return TULIP_ERR_ARG;
    a.bias_table = d->bias_table; a.rel_index = d->rel_index; a.ds0 = d->drop_scale_attn; a.ds1 = d->drop_scale_mlp;
    a.dyb_m = (bf16_t*)d->d_out_mlp; a.dh = (bf16_t*)d->d_fc1_pre; a.dyb_a = (bf16_t*)d->d_out_attn;
    a.dqkv = (bf16_t*)d->d_qkv;
    a.dx_bf16 = (bf16_t*)d->dx_bf16; a.dx_scale = d->dx_bf16_scale;
    a.lnpart1 = d->norm1_partials; a.lnpart2 = d->norm2_partials; a.biaspart = d->bias_partials;
    a.B = d->B; a.H = d->H; a.W = d->W; a.sh = d->shift_h; a.sw = d->shift_w; a.masked = d->masked;
    a.scale = 0.17677669529663687f;
    a.prof = prof;
    const int blocks = d->B * (d->H / 2) * (d->W / (8 * NW));
    const bool hgrad = (d->masked & TULIP_BLOCK_FC1_GRAD) != 0;
    const dim3 g(blocks), b(NT);
#if TULIP_DEV_VARIANTS
    if (prof) {
        if (recomp) hipLaunchKernelGGL((swin96_bwd_kernel<true, false, true>), g, b, 0, stream, a);
        else if (hgrad) hipLaunchKernelGGL((swin96_bwd_kernel<false, true, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((swin96_bwd_kernel<false, false, true>), g, b, 0, stream, a);
    } else if (recomp) hipLaunchKernelGGL((swin96_bwd_kernel<true, false, false>), g, b, 0, stream, a);
    else
#else
    if (prof || recomp) return TULIP_ERR_NOT_BUILT;
#endif
    if (hgrad) hipLaunchKernelGGL((swin96_bwd_kernel<false, true, false>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((swin96_bwd_kernel<false, false, false>), g, b, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
extern "C" int tulip_swin96_block_bwd(const tulip_swin96_bwd_desc* d, hipStream_t stream) { return swin96_bwd_impl(d, nullptr, stream); }
extern "C" int tulip_swin96_block_bwd_profiled(const tulip_swin96_bwd_desc* d, uint64_t* stamps, hipStream_t stream) {
    return swin96_bwd_impl(d, (unsigned long long*)stamps, stream);
}
