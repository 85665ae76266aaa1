// The non-default decoder alternates of the reference: PatchExpanding (tulip.py:126-140, patch_unmerging=False) and
// FinalPatchExpanding (tulip.py:144-159, pixel_shuffle=False).  Both are   Linear(no bias) -> rearrange
// 'B H W (P1 P2 C) -> B (H P1) (W P2) C' -> LayerNorm(C).   The Linear runs on the MFMA GEMM (TULIP_EPI_F32, output
// y = [M][P*P*Cn] fp32 with M = B*H*W coarse tokens); a fine token (b, hP+p1, wP+p2) is then simply the contiguous
// Cn-slice p = p1*P + p2 of coarse row m, so the rearrange is address arithmetic on the OUTPUT side of the
// LayerNorm kernels below and never moves data on its own.
//
//   expand_norm_fwd : y slice -> LayerNorm -> bf16 rows in fine-token order (row pitch ld: e.g. the first half of a
//                     skip-concat buffer, tulip.py:715) and/or, for the final layer, the 1x1 decoder_pred conv
//                     (tulip.py:731, in_chans == 1) as a per-row dot product: the (B,4H,4W,E) tensor is never stored.
//   expand_norm_bwd : upstream gradient in fine-token order (bf16 rows, or d(pred) per fine token for the final layer)
//                     -> LayerNorm backward -> bf16 d(y) in the GEMM's natural [M][P*P*Cn] layout (the operand of the
//                     Linear's dgrad and wgrad GEMMs) + one partial row per workgroup of [dgamma | dbeta | d(dotw)].
//
// One wave per fine row, lanes own channels 4*lane + 256*v (+0..3); not a hot path (no reference launch script uses
// these flags), so the kernels favour clarity: HBM-bound, one pass over y each way.
#include "common.h"
#include "tulip_hip.h"

namespace {

constexpr int WAVES = 4;                  // waves (= rows in flight) per workgroup

struct ExpandArgs {
    const float* y;                       // [M][PP*Cn]
    const float *gamma, *beta;
    float *mean, *rstd;                   // [M*PP] in natural (m, p) order
    bf16_t* out_bf16; int ld;             // fwd: LayerNorm output, fine-token order;  bwd: upstream gradient (same layout)
    const float* dotw;                    // decoder_pred weight [Cn] or nullptr
    float* pred;                          // fwd out / bwd in: [fine tokens]  (pred, d(pred))
    bf16_t* dy_nat;                       // bwd out: [M][PP*Cn]
    float* partials;                      // bwd out: [gridDim.x][3*Cn]
    int B, H, W, P, Cn, rows;             // rows = B*H*W*P*P
    float eps;
};

__device__ __forceinline__ size_t fine_index(const ExpandArgs& a, int r) {
    const int PP = a.P * a.P;
    const int m = r / PP, p = r - m * PP;
    const int p1 = p / a.P, p2 = p - p1 * a.P;
    const int w = m % a.W, t = m / a.W, h = t % a.H, b = t / a.H;
    return ((size_t)b * a.H * a.P + (size_t)h * a.P + p1) * ((size_t)a.W * a.P) + (size_t)w * a.P + p2;
}

template <int NV>
__global__ __launch_bounds__(WAVES * 64) void expand_norm_fwd_kernel(const ExpandArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float inv = 1.0f / a.Cn;
    for (int r = blockIdx.x * WAVES + wid; r < a.rows; r += gridDim.x * WAVES) {
        const float* src = a.y + (size_t)r * a.Cn;
        float4 x[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * lane + 256 * v;
            x[v] = c < a.Cn ? *(const float4*)(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (x[v].x + x[v].y) + (x[v].z + x[v].w);
        }
        const float mu = group_sum<64>(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (4 * lane + 256 * v < a.Cn) {
                const float d0 = x[v].x - mu, d1 = x[v].y - mu, d2 = x[v].z - mu, d3 = x[v].w - mu;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        const float rs = rsqrtf(group_sum<64>(q) * inv + a.eps);
        if (lane == 0) { a.mean[r] = mu; a.rstd[r] = rs; }
        const size_t f = fine_index(a, r);
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * lane + 256 * v;
            if (c < a.Cn) {
                const float4 g = *(const float4*)(a.gamma + c), be = *(const float4*)(a.beta + c);
                const uint32_t lo = pack_bf16x2((x[v].x - mu) * rs * g.x + be.x, (x[v].y - mu) * rs * g.y + be.y);
                const uint32_t hi = pack_bf16x2((x[v].z - mu) * rs * g.z + be.z, (x[v].w - mu) * rs * g.w + be.w);
                if (a.out_bf16) *(uint2*)(a.out_bf16 + f * a.ld + c) = make_uint2(lo, hi);
                if (a.dotw) {                   // the conv operand is the bf16-rounded LayerNorm output
                    const float4 w = *(const float4*)(a.dotw + c);
                    dot += bf2f((bf16_t)(lo & 0xffff)) * w.x + bf2f((bf16_t)(lo >> 16)) * w.y +
                           bf2f((bf16_t)(hi & 0xffff)) * w.z + bf2f((bf16_t)(hi >> 16)) * w.w;
                }
            }
        }
        if (a.dotw) {
            dot = group_sum<64>(dot);
            if (lane == 0) a.pred[f] = dot;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(WAVES * 64) void expand_norm_bwd_kernel(const ExpandArgs a) {
    __shared__ float red[WAVES][3 * NV * 256];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float inv = 1.0f / a.Cn;
    float4 pg[NV], pb[NV], pw[NV];           // this lane's channels, summed over the rows of its wave
#pragma unroll
    for (int v = 0; v < NV; ++v) pg[v] = pb[v] = pw[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = blockIdx.x * WAVES + wid; r < a.rows; r += gridDim.x * WAVES) {
        const float* src = a.y + (size_t)r * a.Cn;
        const size_t f = fine_index(a, r);
        const float mu = a.mean[r], rs = a.rstd[r];
        const float dp = a.dotw ? a.pred[f] : 0.f;
        float4 xh[NV], d[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * lane + 256 * v;
            xh[v] = d[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < a.Cn) {
                const float4 x = *(const float4*)(src + c), g = *(const float4*)(a.gamma + c);
                xh[v] = make_float4((x.x - mu) * rs, (x.y - mu) * rs, (x.z - mu) * rs, (x.w - mu) * rs);
                float4 dy;
                if (a.dotw) {
                    const float4 w = *(const float4*)(a.dotw + c), be = *(const float4*)(a.beta + c);
                    dy = make_float4(dp * w.x, dp * w.y, dp * w.z, dp * w.w);
                    // d(decoder_pred.weight)[c] += d(pred) * bf16(LayerNorm output)[c]
                    const uint32_t lo = pack_bf16x2(xh[v].x * g.x + be.x, xh[v].y * g.y + be.y);
                    const uint32_t hi = pack_bf16x2(xh[v].z * g.z + be.z, xh[v].w * g.w + be.w);
                    pw[v].x += dp * bf2f((bf16_t)(lo & 0xffff)); pw[v].y += dp * bf2f((bf16_t)(lo >> 16));
                    pw[v].z += dp * bf2f((bf16_t)(hi & 0xffff)); pw[v].w += dp * bf2f((bf16_t)(hi >> 16));
                } else {
                    const uint2 u = *(const uint2*)(a.out_bf16 + f * a.ld + c);
                    dy = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)),
                                     bf2f((bf16_t)(u.y >> 16)));
                }
                pg[v].x += dy.x * xh[v].x; pg[v].y += dy.y * xh[v].y; pg[v].z += dy.z * xh[v].z; pg[v].w += dy.w * xh[v].w;
                pb[v].x += dy.x; pb[v].y += dy.y; pb[v].z += dy.z; pb[v].w += dy.w;
                d[v] = make_float4(dy.x * g.x, dy.y * g.y, dy.z * g.z, dy.w * g.w);
                s1 += (d[v].x + d[v].y) + (d[v].z + d[v].w);
                s2 += (d[v].x * xh[v].x + d[v].y * xh[v].y) + (d[v].z * xh[v].z + d[v].w * xh[v].w);
            }
        }
        const float m1 = group_sum<64>(s1) * inv, m2 = group_sum<64>(s2) * inv;
        bf16_t* dst = a.dy_nat + (size_t)r * a.Cn;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * lane + 256 * v;
            if (c < a.Cn)
                *(uint2*)(dst + c) = make_uint2(
                    pack_bf16x2(rs * (d[v].x - m1 - xh[v].x * m2), rs * (d[v].y - m1 - xh[v].y * m2)),
                    pack_bf16x2(rs * (d[v].z - m1 - xh[v].z * m2), rs * (d[v].w - m1 - xh[v].w * m2)));
        }
    }
    // one partial row per workgroup: waves summed in a fixed order (deterministic; folded by tulip_reduce_rows_multi)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float* w = red[wid] + 4 * lane + 256 * v;
        *(float4*)(w) = pg[v];
        *(float4*)(w + NV * 256) = pb[v];
        *(float4*)(w + 2 * NV * 256) = pw[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * a.Cn; i += WAVES * 64) {
        const int which = i / a.Cn, c = i - which * a.Cn;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][which * NV * 256 + c];
        a.partials[(size_t)blockIdx.x * 3 * a.Cn + i] = s;
    }
}

int grid_rows(int rows) { return rows < 4 * 1024 ? (rows + WAVES - 1) / WAVES : 1024; }

bool bad(int B, int H, int W, int P, int Cn) {
    return B <= 0 || H <= 0 || W <= 0 || (P != 2 && P != 4 && P != 8) || Cn <= 0 || (Cn & 3) || Cn > 768 ||
           (int64_t)B * H * W * P * P > (int64_t)1 << 30;
}

}  // namespace

extern "C" int tulip_expand_norm_bwd_partial_rows(int B, int H, int W, int P) {
    if (B <= 0 || H <= 0 || W <= 0 || P <= 0) return 0;
    return grid_rows(B * H * W * P * P);
}

extern "C" int tulip_expand_norm_fwd(const float* y, const float* gamma, const float* beta, uint16_t* out_bf16, int ld,
                                     const float* dotw, float* pred, float* mean, float* rstd, int B, int H, int W,
                                     int P, int Cn, float eps, hipStream_t stream) {
    if (bad(B, H, W, P, Cn) || !y || !gamma || !beta || !mean || !rstd || (!out_bf16 && !dotw) || (dotw && !pred) ||
        (out_bf16 && (ld < Cn || (ld & 3))))
        return TULIP_ERR_ARG;
    ExpandArgs a{};
    a.y = y; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.out_bf16 = out_bf16; a.ld = ld;
    a.dotw = dotw; a.pred = pred; a.B = B; a.H = H; a.W = W; a.P = P; a.Cn = Cn; a.rows = B * H * W * P * P; a.eps = eps;
    const dim3 grid(grid_rows(a.rows)), block(WAVES * 64);
    if (Cn <= 256) hipLaunchKernelGGL(expand_norm_fwd_kernel<1>, grid, block, 0, stream, a);
    else if (Cn <= 512) hipLaunchKernelGGL(expand_norm_fwd_kernel<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(expand_norm_fwd_kernel<3>, grid, block, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_expand_norm_bwd(const uint16_t* dy_fine, int ld, const float* dpred, const float* dotw, const float* y,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     uint16_t* dy_nat, float* partials, int B, int H, int W, int P, int Cn,
                                     hipStream_t stream) {
    if (bad(B, H, W, P, Cn) || !y || !gamma || !mean || !rstd || !dy_nat || !partials || (!dy_fine && !dotw) ||
        (dotw && (!dpred || !beta)) || (dy_fine && !dotw && (ld < Cn || (ld & 3))))
        return TULIP_ERR_ARG;
    ExpandArgs a{};
    a.y = y; a.gamma = gamma; a.beta = beta; a.mean = (float*)mean; a.rstd = (float*)rstd;
    a.out_bf16 = (bf16_t*)dy_fine; a.ld = ld; a.dotw = dotw; a.pred = (float*)dpred; a.dy_nat = dy_nat;
    a.partials = partials; a.B = B; a.H = H; a.W = W; a.P = P; a.Cn = Cn; a.rows = B * H * W * P * P; a.eps = 0.f;
    const dim3 grid(grid_rows(a.rows)), block(WAVES * 64);
    if (Cn <= 256) hipLaunchKernelGGL(expand_norm_bwd_kernel<1>, grid, block, 0, stream, a);
    else if (Cn <= 512) hipLaunchKernelGGL(expand_norm_bwd_kernel<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(expand_norm_bwd_kernel<3>, grid, block, 0, stream, a);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}
