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
#include <cstdlib>
#include "common.h"
#include "tulip_hip.h"

namespace {

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
    int stop;     // dev: return after phase N (0 = run everything)
};

__device__ __forceinline__ int region(int x, int X, int wsz, int ssz) {       // create_mask slices, tulip.py:261-266
    return (ssz == 0 || x >= X - ssz) ? 2 : (x >= X - wsz ? 1 : 0);
}
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (short)f2bf(a); r[1] = (short)f2bf(b); r[2] = (short)f2bf(c); r[3] = (short)f2bf(d);
    return r;
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    bf16x4 v;
    const unsigned a = (unsigned)(uintptr_t)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
// weight fragment in the chained-operand k order: k slots 0..3 <- columns c0..c0+3, slots 4..7 <- c0+16..c0+19
__device__ __forceinline__ bf16x8 wfrag(const unsigned char* rowp, int c0) {
    return cat8(*(const bf16x4*)(rowp + c0 * 2), *(const bf16x4*)(rowp + (c0 + 16) * 2));
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

__global__ __launch_bounds__(NT) void swin96_fwd_kernel(const Swin96Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, t = lane & 15, gq = lane >> 4;
    unsigned char* ldsV = smem + OFF_V + wid * 1024;
    float* prm = (float*)(smem + OFF_P);

    // ---- which window, which token (cyclic shift + window partition are address arithmetic, tulip.py:289-297)
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

    // the token's row: lane owns channels 16n + 4gq .. +3 (n = 0..5) -- the accumulator layout of every GEMM below
    f32x4 xv[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const float4 u = *(const float4*)(a.xin + row * C + 16 * n + 4 * gq);
        xv[n] = (f32x4){u.x, u.y, u.z, u.w};
    }
    // relative-position bias of this lane's (query t, keys 4gq..4gq+3), all heads (tulip.py:304-308)
    float rpb[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = a.rel_index[t * 16 + gq * 4 + r] * 3;
#pragma unroll
        for (int h = 0; h < 3; ++h) rpb[h][r] = a.bias_table[e + h];
    }
    stage_weights<288, C, PW>(a.wqkv, smem + OFF_A, tid);
    stage_weights<C, C, PW>(a.wproj, smem + OFF_WPROJ, tid);
    stage_weights<C, HID, PW2>(a.w2, smem + OFF_W2, tid);
    for (int i = tid; i < P_N; i += NT) {
        float v;
        if (i < P_BQKV) v = a.b1[i];
        else if (i < P_BPROJ) v = a.bqkv[i - P_BQKV];
        else if (i < P_B2) v = a.bproj[i - P_BPROJ];
        else if (i < P_G2) v = a.b2[i - P_B2];
        else if (i < P_BE2) v = a.g2[i - P_G2];
        else v = a.be2[i - P_BE2];
        prm[i] = v;
    }

    // ---- norm1 (tulip.py:340)
    bf16x8 xfrag[3];
    {
        float s1 = 0.f;
#pragma unroll
        for (int n = 0; n < 6; ++n) s1 += (xv[n][0] + xv[n][1]) + (xv[n][2] + xv[n][3]);
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        const float mu = s1 * (1.0f / C);
        float s2 = 0.f;
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = xv[n][r] - mu; s2 += d * d; }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float rs = rsqrtf(s2 * (1.0f / C) + a.eps);
        if (gq == 0) { a.mean1[row] = mu; a.rstd1[row] = rs; }
        bf16x4 p1[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const int c0 = 16 * n + 4 * gq;
            const float4 ga = *(const float4*)(a.g1 + c0), be = *(const float4*)(a.be1 + c0);
            p1[n] = pack4((xv[n][0] - mu) * rs * ga.x + be.x, (xv[n][1] - mu) * rs * ga.y + be.y,
                          (xv[n][2] - mu) * rs * ga.z + be.z, (xv[n][3] - mu) * rs * ga.w + be.w);
            *(bf16x4*)(a.xn1 + row * C + c0) = p1[n];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) xfrag[s] = cat8(p1[2 * s], p1[2 * s + 1]);   // k order within 32s: 4gq.., 16+4gq..
    }
    __syncthreads();                                        // weights of phase 1, W2 and the parameter vectors are in LDS
    if (a.stop == 1) return;

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
        *(bf16x4*)(a.qkv + row * 288 + 16 * j + 4 * gq) = qkvp[j];
    }
    if (a.stop == 2) return;

    // ---- attention, one head at a time (tulip.py:300-317); scores issued as K.Q^T: lane = query t, keys 4gq + r
    bf16x8 ofrag[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const bf16x8 qf = cat8(qkvp[2 * h], qkvp[2 * h + 1]);
        const bf16x8 kf = cat8(qkvp[6 + 2 * h], qkvp[7 + 2 * h]);
        *(bf16x4*)(ldsV + t * 64 + (4 * gq) * 2) = qkvp[12 + 2 * h];            // V tile [token][d], d = 0..15
        *(bf16x4*)(ldsV + t * 64 + (16 + 4 * gq) * 2) = qkvp[13 + 2 * h];       // d = 16..31
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sc, 0, 0, 0);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = sc[r] * a.scale + rpb[h][r];
            if (a.masked) {
                const int kl = __shfl(lab, gq * 4 + r, 64);
                if (kl != lab) x += -100.0f;
            }
            sc[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = __expf(sc[r] - mx); sum += sc[r]; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const bf16x4 pb = pack4(sc[0] * inv, sc[1] * inv, sc[2] * inv, sc[3] * inv);
        bf16x4 op[2];
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const bf16x4 vt = tr_read(ldsV + (gq * 4 + (t >> 2)) * 64 + dc * 32 + (t & 3) * 8);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb, o, 0, 0, 0);   // o[r] = O[t][16dc + 4gq + r]
            op[dc] = pack4(o[0], o[1], o[2], o[3]);
            *(bf16x4*)(a.o + row * C + 32 * h + 16 * dc + 4 * gq) = op[dc];
        }
        ofrag[h] = cat8(op[0], op[1]);                    // k order: d = 4gq+0..3, 16+4gq+0..3
    }
    if (a.stop == 3) return;

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
            *(float4*)(a.x1 + row * C + c0) = make_float4(xv[n2][0], xv[n2][1], xv[n2][2], xv[n2][3]);
            sum += (xv[n2][0] + xv[n2][1]) + (xv[n2][2] + xv[n2][3]);
        }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        const float mu = sum * (1.0f / C);
        float s2 = 0.f;
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = xv[n2][r] - mu; s2 += d * d; }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float rs = rsqrtf(s2 * (1.0f / C) + a.eps);
        if (gq == 0) { a.mean2[row] = mu; a.rstd2[row] = rs; }
        bf16x4 p2[6];
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2) {
            const int c0 = 16 * n2 + 4 * gq;
            const float4 ga = *(const float4*)(prm + P_G2 + c0), be = *(const float4*)(prm + P_BE2 + c0);
            p2[n2] = pack4((xv[n2][0] - mu) * rs * ga.x + be.x, (xv[n2][1] - mu) * rs * ga.y + be.y,
                           (xv[n2][2] - mu) * rs * ga.z + be.z, (xv[n2][3] - mu) * rs * ga.w + be.w);
            *(bf16x4*)(a.xn2 + row * C + c0) = p2[n2];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) x2frag[s] = cat8(p2[2 * s], p2[2 * s + 1]);
    }
    if (a.stop == 4) return;

    // ---- fc1 weights replace qkv/proj weights in LDS
    __syncthreads();
    stage_weights<HID, C, PW>(a.w1, smem + OFF_A, tid);
    __syncthreads();
    if (a.stop == 5) return;

    // ---- fc1 -> exact-erf GELU -> fc2 (tulip.py:195-198), 32 hidden channels at a time, chained in registers
    f32x4 acc3[6];
#pragma unroll
    for (int n2 = 0; n2 < 6; ++n2) acc3[n2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int p = 0; p < 12; ++p) {
        bf16x4 gp[2];
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
            *(bf16x4*)(a.h + row * HID + c0) = hp;
            gp[jj] = pack4(gelu_exact(bf2f((bf16_t)hp[0])), gelu_exact(bf2f((bf16_t)hp[1])),
                           gelu_exact(bf2f((bf16_t)hp[2])), gelu_exact(bf2f((bf16_t)hp[3])));   // GELU of the stored h
            *(bf16x4*)(a.g + row * HID + c0) = gp[jj];
        }
        const bf16x8 gf = cat8(gp[0], gp[1]);
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2)
            acc3[n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(smem + OFF_W2 + (16 * n2 + t) * PW2, 32 * p + 4 * gq), gf,
                                                               acc3[n2], 0, 0, 0);
    }
#pragma unroll
    for (int n2 = 0; n2 < 6; ++n2) {
        const int c0 = 16 * n2 + 4 * gq;
        const float4 bb = *(const float4*)(prm + P_B2 + c0);
        *(float4*)(a.xout + row * C + c0) =
            make_float4(xv[n2][0] + s1v * (acc3[n2][0] + bb.x), xv[n2][1] + s1v * (acc3[n2][1] + bb.y),
                        xv[n2][2] + s1v * (acc3[n2][2] + bb.z), xv[n2][3] + s1v * (acc3[n2][3] + bb.w));
    }
}

}  // namespace

extern "C" int tulip_swin96_block_fwd(const tulip_swin96_desc* d, hipStream_t stream) {
    if (!d || d->B <= 0 || d->H <= 0 || (d->H & 1) || d->W <= 0 || (d->W & 63) || d->shift_h < 0 ||
        d->shift_h >= d->H || d->shift_w < 0 || d->shift_w >= d->W)
        return TULIP_ERR_ARG;
    Swin96Args a;
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
    a.stop = getenv("TULIP_SWIN96_STOP") ? atoi(getenv("TULIP_SWIN96_STOP")) : 0;
    const int blocks = d->B * (d->H / 2) * (d->W / (8 * NW));
    hipLaunchKernelGGL(swin96_fwd_kernel, dim3(blocks), dim3(NT), 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
