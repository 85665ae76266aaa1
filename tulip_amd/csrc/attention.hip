// Shifted-window attention core for 16-token windows (2x8, or the 1x16 backup window), forward and
// backward.   gfx950 only.
//
// One wave owns one (window, head) pair at a time.  Cyclic shift, window partition and their
// inverses (tulip.py:289-290, 248-252, 320-323) are pure address arithmetic: the 16 token rows of a
// window are gathered from / scattered to their natural (b,h,w) positions in the [B*H*W][3C] qkv
// tensor, so no roll / permute copies exist.
//
// A 16x16 score tile with head_dim 32 is exactly one v_mfma_f32_16x16x32_bf16 (head_dim 16: upper
// k-half zero).  The MFMA is issued as K.Q^T so lane l holds S[query=l&15][key=4*(l>>4)+r]: the
// row softmax is 4 in-lane values + two cross-lane steps (xor 16, 32), and the probabilities are
// already the B operand of the 16x16x16 P.V MFMA.  V / K / Q / dO tiles that must be consumed
// "token-major" go through a 1 KiB per-wave LDS tile and ds_read_b64_tr_b16 (LDS transpose read).
#include "common.h"
#include "tulip_hip.h"

namespace {

struct AttnGeom {
    int B, H, W, C, nh;
    int wh, ww, sh, sw, masked, fp8;
    int nWy, nWx;
    float scale;
};

__device__ __forceinline__ int region(int x, int X, int wsz, int ssz) {
    // create_mask slices (tulip.py:261-266): [0:-wsz]=0, [-wsz:-ssz]=1, [-ssz:]=2; later wins, and
    // ssz==0 makes the last slice [0:] (Python -0) cover everything.
    return (ssz == 0 || x >= X - ssz) ? 2 : (x >= X - wsz ? 1 : 0);
}

__device__ __forceinline__ void slot_info(const AttnGeom& g, int b, int wy, int wx, int s, int& row, int& label) {
    const int i = fast_div(s, g.ww), j = s - i * g.ww;
    const int hs = wy * g.wh + i, ws = wx * g.ww + j;  // coordinates in the rolled image
    int h = hs + g.sh; if (h >= g.H) h -= g.H;         // rolled[hs] = x[(hs+sh) mod H]
    int w = ws + g.sw; if (w >= g.W) w -= g.W;
    row = (b * g.H + h) * g.W + w;
    label = 3 * region(hs, g.H, g.wh, g.sh) + region(ws, g.W, g.ww, g.sw);
}

// ds_read_b64_tr_b16 through the builtin: the compiler batches the waits of consecutive reads
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)p);
}

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    return __builtin_bit_cast(bf16x4, (u32x2_t){pack_bf16x2(a, b), pack_bf16x2(c, d)});
}

__device__ __forceinline__ void store4(bf16_t* p, f32x4 v, float s) {
    *(uint2*)p = make_uint2(pack_bf16x2(v[0] * s, v[1] * s), pack_bf16x2(v[2] * s, v[3] * s));
}

// wave -> (head, window group); every wave keeps one head so d(bias) accumulates in registers
struct WaveMap {
    int h, grp, ngrp;
};
__device__ __forceinline__ WaveMap wave_map(int nh, int ngrp) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    WaveMap m;
    m.h = wave % nh; m.grp = wave / nh; m.ngrp = ngrp;
    return m;
}

template <int P>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                       const float* __restrict__ bias_table,
                                                       const int* __restrict__ rel_index, bf16_t* __restrict__ out,
                                                       AttnGeom g, int ngrp) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 16 * P * 2];
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4;
    unsigned char* ldsV = smem + (threadIdx.x >> 6) * (16 * P * 2);
    const WaveMap wm = wave_map(g.nh, ngrp);
    if (wm.grp >= ngrp) return;
    const int h = wm.h;
    const int nW = g.nWy * g.nWx, total = g.B * nW;
    const bool dvalid = gq * 8 < P;
    const int C3 = 3 * g.C;

    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = bias_table[rel_index[li * 16 + gq * 4 + r] * g.nh + h];

    for (int win = wm.grp; win < total; win += ngrp) {
        const int b = fast_div(win, nW), wloc = win - b * nW;
        const int wy = fast_div(wloc, g.nWx), wx = wloc - wy * g.nWx;
        int row, lab;
        slot_info(g, b, wy, wx, li, row, lab);
        const bf16_t* src = qkv + (size_t)row * C3 + h * P + gq * 8;
        bf16x8 q = {0, 0, 0, 0, 0, 0, 0, 0}, k = q, v = q;
        if (dvalid) {
            q = *(const bf16x8*)src;
            k = *(const bf16x8*)(src + g.C);
            v = *(const bf16x8*)(src + 2 * g.C);
            *(bf16x8*)(ldsV + li * (P * 2) + gq * 16) = v;
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (g.fp8) s = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf16x8_to_fp8(k), bf16x8_to_fp8(q), s, 0, 0, 0);
        else s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k, q, s, 0, 0, 0);  // s[r] = q_li . k_(4gq+r)
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = s[r] * g.scale + bias[r];
            if (g.masked) {
                const int kl = __shfl(lab, gq * 4 + r, 64);
                if (kl != lab) x += -100.0f;
            }
            s[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - mx); sum += s[r]; }
        sum = rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        const bf16x4 pb = pack4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
        bf16_t* dst = out + (size_t)row * g.C + h * P + gq * 4;
#pragma unroll
        for (int dc = 0; dc < P / 16; ++dc) {
            const bf16x4 vt = tr_read(ldsV + (gq * 4 + (li >> 2)) * (P * 2) + dc * 32 + (li & 3) * 8);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb, o, 0, 0, 0);  // o[r] = O[li][16dc+4gq+r]
            store4(dst + dc * 16, o, 1.0f);
        }
    }
}

template <int P>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                       const float* __restrict__ bias_table,
                                                       const int* __restrict__ rel_index, bf16_t* __restrict__ dqkv,
                                                       float* dbias_part, AttnGeom g, int ngrp) {
    // All 4 waves of a workgroup serve the SAME head (blockIdx.x % nh), so d(bias) is summed over the
    // workgroup in LDS and leaves as one plain 256-float partial row per workgroup:
    // dbias_part[blockIdx.x][i*16+j].  Rows of one head are nh apart -> viewed as [gridDim.x/nh][nh*256]
    // the partials fold into the dense [nh][16][16] gradient with a single row reduction.
    constexpr int TILE = 16 * P * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 3 * TILE];
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, li = lane & 15, gq = lane >> 4, wid = threadIdx.x >> 6;
    unsigned char* ldsQ = smem + wid * (3 * TILE);
    unsigned char* ldsK = ldsQ + TILE;
    unsigned char* ldsD = ldsK + TILE;
    const int h = blockIdx.x % g.nh;
    WaveMap wm;
    wm.h = h; wm.grp = (blockIdx.x / g.nh) * 4 + wid; wm.ngrp = ngrp;
    const int nW = g.nWy * g.nWx, total = g.B * nW;
    const bool dvalid = gq * 8 < P;
    const int C3 = 3 * g.C;

    float bias_q[4], bias_k[4];  // Lq: (query li, key 4gq+r)   Lk: (query 4gq+r, key li)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bias_q[r] = bias_table[rel_index[li * 16 + gq * 4 + r] * g.nh + h];
        bias_k[r] = bias_table[rel_index[(gq * 4 + r) * 16 + li] * g.nh + h];
    }
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};

    for (int win = wm.grp; win < total; win += ngrp) {
        const int b = fast_div(win, nW), wloc = win - b * nW;
        const int wy = fast_div(wloc, g.nWx), wx = wloc - wy * g.nWx;
        int row, lab;
        slot_info(g, b, wy, wx, li, row, lab);
        const bf16_t* src = qkv + (size_t)row * C3 + h * P + gq * 8;
        bf16x8 q = {0, 0, 0, 0, 0, 0, 0, 0}, k = q, v = q, d = q;
        if (dvalid) {
            q = *(const bf16x8*)src;
            k = *(const bf16x8*)(src + g.C);
            v = *(const bf16x8*)(src + 2 * g.C);
            d = *(const bf16x8*)(dout + (size_t)row * g.C + h * P + gq * 8);
            if (g.fp8) { q = round_through_fp8(q); k = round_through_fp8(k); }     // the values the forward's scores saw
            const int off = li * (P * 2) + gq * 16;
            *(bf16x8*)(ldsQ + off) = q;
            *(bf16x8*)(ldsK + off) = k;
            *(bf16x8*)(ldsD + off) = d;
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k, q, z, 0, 0, 0);   // Lq: S[li][4gq+r]
        f32x4 sk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, k, z, 0, 0, 0);   // Lk: S[4gq+r][li]
        f32x4 dpq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, d, z, 0, 0, 0);  // Lq: dP[li][4gq+r] = dO_li . V_key
        f32x4 dpk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d, v, z, 0, 0, 0);  // Lk: dP[4gq+r][li]

        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ol = __shfl(lab, gq * 4 + r, 64);
            float xq = sq[r] * g.scale + bias_q[r];
            float xk = sk[r] * g.scale + bias_k[r];
            if (g.masked && ol != lab) { xq += -100.0f; xk += -100.0f; }
            sq[r] = xq; sk[r] = xk;
            mx = fmaxf(mx, xq);
        }
        mx = rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += __expf(sq[r] - mx);
        sum = rows_sum(sum);
        const float lse = mx + __logf(sum);  // for query li
        float pq[4], pk[4], delta = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pq[r] = __expf(sq[r] - lse);
            pk[r] = __expf(sk[r] - __shfl(lse, gq * 4 + r, 64));
            delta += pq[r] * dpq[r];
        }
        delta = rows_sum(delta);  // sum_key P*dP for query li
        float dsq[4], dsk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dsq[r] = pq[r] * (dpq[r] - delta);
            dsk[r] = pk[r] * (dpk[r] - __shfl(delta, gq * 4 + r, 64));
            dbacc[r] += dsq[r];
        }
        const bf16x4 dsq_b = pack4(dsq[0], dsq[1], dsq[2], dsq[3]);
        const bf16x4 dsk_b = pack4(dsk[0], dsk[1], dsk[2], dsk[3]);
        const bf16x4 pk_b = pack4(pk[0], pk[1], pk[2], pk[3]);
        bf16_t* dst = dqkv + (size_t)row * C3 + h * P + gq * 4;
        const int troff = (gq * 4 + (li >> 2)) * (P * 2) + (li & 3) * 8;
#pragma unroll
        for (int dc = 0; dc < P / 16; ++dc) {
            const bf16x4 kt = tr_read(ldsK + troff + dc * 32);  // K[4gq+jj][16dc+li]
            const bf16x4 qt = tr_read(ldsQ + troff + dc * 32);  // Q[4gq+jj][16dc+li]
            const bf16x4 dt = tr_read(ldsD + troff + dc * 32);  // dO[4gq+jj][16dc+li]
            // dQ[li][d] = scale * sum_key dS[li][key] K[key][d]
            f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, dsq_b, z, 0, 0, 0);
            // dK[li][d] = scale * sum_q dS[q][li] Q[q][d]
            f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dsk_b, z, 0, 0, 0);
            // dV[li][d] = sum_q P[q][li] dO[q][d]
            f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dt, pk_b, z, 0, 0, 0);
            store4(dst + dc * 16, dq, g.scale);
            store4(dst + g.C + dc * 16, dk, g.scale);
            store4(dst + 2 * g.C + dc * 16, dv, 1.0f);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][li * 16 + gq * 4 + r] = dbacc[r];
    __syncthreads();
    dbias_part[(size_t)blockIdx.x * 256 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

bool make_geom(AttnGeom& g, int B, int H, int W, int C, int nh, int wh, int ww, int sh, int sw, int masked) {
    if (wh * ww != 16 || nh <= 0 || C % nh) return false;
    const int P = C / nh;
    if (P != 16 && P != 32) return false;
    if (H % wh || W % ww || sh >= H + (sh == 0) || sw >= W + (sw == 0)) return false;
    g.B = B; g.H = H; g.W = W; g.C = C; g.nh = nh; g.wh = wh; g.ww = ww; g.sh = sh; g.sw = sw;
    g.masked = masked & TULIP_ATTN_MASKED; g.fp8 = (masked & TULIP_ATTN_FP8) ? 1 : 0;
    g.nWy = H / wh; g.nWx = W / ww;
    g.scale = 1.0f / sqrtf((float)P);
    return true;
}

int pick_groups(const AttnGeom& g) {
    const int total = g.B * g.nWy * g.nWx;
    int ngrp = (256 * 16 + g.nh - 1) / g.nh;  // ~16 waves per CU
    if (ngrp > total) ngrp = total;
    if (ngrp < 1) ngrp = 1;
    return ngrp;
}

}  // namespace

extern "C" int tulip_window_attn_fwd(const uint16_t* qkv, const float* bias_table, const int32_t* rel_index,
                                     uint16_t* out, int B, int H, int W, int C, int nh, int wh, int ww, int sh, int sw,
                                     int masked, hipStream_t stream) {
    AttnGeom g;
    if (!make_geom(g, B, H, W, C, nh, wh, ww, sh, sw, masked)) return TULIP_ERR_ARG;
    if (B <= 0) return TULIP_OK;
    const int ngrp = pick_groups(g);
    const int blocks = (ngrp * nh + 3) / 4;
    if (C / nh == 32)
        hipLaunchKernelGGL(attn_fwd_kernel<32>, dim3(blocks), dim3(256), 0, stream, qkv, bias_table, rel_index, out, g,
                           ngrp);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<16>, dim3(blocks), dim3(256), 0, stream, qkv, bias_table, rel_index, out, g,
                           ngrp);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

// workgroups per head of the backward launch (each emits one 256-float d(bias) partial row)
static int bwd_blocks_per_head(int windows_total, int nh) {
    int ngrp = (256 * 8 + nh - 1) / nh;  // ~8 waves per CU
    if (ngrp > windows_total) ngrp = windows_total;
    if (ngrp < 1) ngrp = 1;
    return (ngrp + 3) / 4;
}

extern "C" int tulip_window_attn_bwd_partial_rows(int B, int H, int W, int nh, int wh, int ww) {
    if (B <= 0 || nh <= 0 || wh <= 0 || ww <= 0) return 0;
    return bwd_blocks_per_head(B * (H / wh) * (W / ww), nh);
}

extern "C" int tulip_window_attn_bwd(const uint16_t* qkv, const uint16_t* dout, const float* bias_table,
                                     const int32_t* rel_index, uint16_t* dqkv, float* dbias_partials, int B, int H,
                                     int W, int C, int nh, int wh, int ww, int sh, int sw, int masked,
                                     hipStream_t stream) {
    AttnGeom g;
    if (!make_geom(g, B, H, W, C, nh, wh, ww, sh, sw, masked)) return TULIP_ERR_ARG;
    if (B <= 0) return TULIP_OK;
    const int bph = bwd_blocks_per_head(g.B * g.nWy * g.nWx, nh);
    const int blocks = bph * nh, ngrp = bph * 4;
    if (C / nh == 32)
        hipLaunchKernelGGL(attn_bwd_kernel<32>, dim3(blocks), dim3(256), 0, stream, qkv, dout, bias_table, rel_index,
                           dqkv, dbias_partials, g, ngrp);
    else
        hipLaunchKernelGGL(attn_bwd_kernel<16>, dim3(blocks), dim3(256), 0, stream, qkv, dout, bias_table, rel_index,
                           dqkv, dbias_partials, g, ngrp);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
