// bf16 MFMA GEMM for the TULIP linears / 1x1 convs, forward + dgrad + wgrad.   gfx950 only.
//
//   C[M,N] = opA[M,K] . opB[N,K]^T        (fp32 accumulate, v_mfma_f32_16x16x32_bf16)
//
// Operand layouts (runtime independent, compile-time template):
//   A_T = false : A stored [M][lda], k contiguous   (activations in fwd / dgrad)
//   A_T = true  : A stored [K][lda], m contiguous   (wgrad: dY^T without materialising it)
//   B_T = false : B stored [N][ldb], k contiguous   (nn.Linear weight [out,in] in fwd)
//   B_T = true  : B stored [K][ldb], n contiguous   (dgrad: W as is; wgrad: X as is)
// k-contiguous tiles sit in LDS as [row][32 k] (64 B rows, 16-B chunk XOR swizzle, ds_read_b128
// fragments); k-slow tiles sit as [32 k][rows] (288 B pitch, 32-B chunk XOR swizzle) and are read
// with ds_read_b64_tr_b16, the CDNA4 LDS transpose read, so neither dgrad nor wgrad needs a
// transposed copy of weights or activations in HBM.
//
// Tile: BM x 96 x 32 per 256-thread workgroup (4 waves as 2x2, wave tile BM/2 x 48), double
// buffered LDS, register prefetch of the next k-tile.  The MFMA is issued "swapped"
// (a := B fragment, b := A fragment) so each lane ends up with 4 consecutive output columns of
// one row -> 8/16-byte epilogue loads and stores.
//
// Epilogues fuse: bias, exact-erf GELU (dual store), GELU backward, residual add with the
// per-sample DropPath multiplier, PixelShuffle(2) scatter (PatchUnmerging) and its inverse, fp32 accumulate, and
// split-K partial slabs (deterministic fold).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "tulip_hip.h"

#ifndef TULIP_GEMM_FAST_SHUF
#define TULIP_GEMM_FAST_SHUF 1
#endif
namespace {

constexpr int BN = 96;
constexpr int BK = 32;
constexpr int T_PITCH = 288;            // bytes per k-row of a k-slow tile (128 rows * 2 B + 32 B pad)
constexpr int TILE_BYTES = 32 * T_PITCH;  // 9216 >= 128 * 64 (k-contiguous tile)

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    int lda, ldb;
    int M, N, K;
    int kchunk;  // K range per blockIdx.z (multiple of 32)
    int epi;
    const float* bias;
    void* out;
    int ldo;
    void* out2;
    int ldo2;
    const void* aux;
    int ldaux;
    const float* rowscale;
    int rows_per_sample;
    int accumulate;
    int psH, psW;
    int touch;   // first touch of the cold weight panel split between the M-tile workgroups (off: TULIP_GEMM_NO_TOUCH)
    int checked; // the bounds-checked kernels even for whole-tile shapes (TULIP_GEMM_CHECKED: bit-compare tests)
    int mid;     // the 192 x 192 kernel for mid-size shapes: 2 = the caller asked for it (TULIP_GEMM_MID: wherever it fits), else never
    int vec_ok;  // the vectorised PixelShuffle / inverse-shuffle write-outs may be used: pitches and bases aligned for their 16-B / 4-B stores
};


// launch heuristics (compile-time; mirrored by bench.py's kernel-name bookkeeping)
#define TULIP_GEMM_MID_TILES 512    // taller tiles only while the launch still has this many of them (two rounds of the chip)
#define TULIP_GEMM_KSUB_GRID 400    // 128-deep k stages for grids up to this many workgroups
#ifndef TULIP_WGRAD_RING
#define TULIP_WGRAD_RING 3
#endif
#ifndef TULIP_GEMM_DEEP_RING
#define TULIP_GEMM_DEEP_RING 2      // register ring of the 128-deep k stages
#endif
// f(integral_constant<0>) ... f(integral_constant<N-1>), unrolled at compile time
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// ---- global -> registers (16 B chunks), zero fill out of range -------------------------------
// FULL: the launcher checked that every tile is whole (M, N multiples of the tile, the K range of a workgroup a multiple of the
// stage depth): no bounds test -- i.e. no exec-masked branch and no zero fill -- around the loads (the bounds-checked kernel carries
// 92 s_and_saveexec and 113 v_mov for its 64 static loads; 37 chain GEMMs of the batch-8 step isolated: 381 -> 355 us).  Walking the
// 96-row operand's 4 x 384 chunks as ONE list of 6 per thread, which also removes the half-empty second slot per sub-tile,
// measured much slower (522 us) and is not done.
template <int ROWS, bool T, int KSUB, bool FULL = false>
struct Stage {
    static constexpr int NCHUNK = ROWS * 4;                 // 16-B chunks per 32-deep k sub-tile
    static constexpr int PER_THREAD = (NCHUNK + 255) / 256;
    static constexpr int SUB_BYTES = T ? 32 * T_PITCH : ROWS * 64;   // LDS bytes of one sub-tile
    uint4 r[KSUB][PER_THREAD];

    // KSUB sub-tiles of 32 k each: all their 16-B loads are issued back to back (memory-level
    // parallelism is what the small-M / large-K launches of the deep stages are starved of)
    __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int k0, int kend,
                                         int tid) {
#pragma unroll
        for (int sidx = 0; sidx < KSUB; ++sidx) {
            const int ks = k0 + sidx * 32;
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                int c = tid + i * 256;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (NCHUNK % 256 == 0 || c < NCHUNK) {
                    if (!T) {
                        int row = c >> 2, kc = c & 3;
                        int gr = row0 + row, gk = ks + kc * 8;
                        if (FULL || (gr < nrows && gk < kend)) v = *(const uint4*)(base + (size_t)gr * ld + gk);
                    } else {
                        constexpr int CPR = ROWS / 8;  // chunks per k row
                        int k = c / CPR, mc = c % CPR;
                        int gk = ks + k, gr = row0 + mc * 8;
                        if (FULL || (gk < kend && gr < nrows)) v = *(const uint4*)(base + (size_t)gk * ld + gr);
                    }
                }
                r[sidx][i] = v;
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
        for (int sidx = 0; sidx < KSUB; ++sidx) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; ++i) {
                int c = tid + i * 256;
                if (NCHUNK % 256 == 0 || c < NCHUNK) {
                    int off;
                    if (!T) {
                        int row = c >> 2, kc = c & 3;
                        off = row * 64 + ((kc ^ swz4(row)) << 4);
                    } else {
                        constexpr int CPR = ROWS / 8;
                        int k = c / CPR, mc = c % CPR;
                        off = k * T_PITCH + ((((mc >> 1) ^ (((k >> 3) & 1) << 2))) << 5) + ((mc & 1) << 4);
                    }
                    *(uint4*)(lds + sidx * SUB_BYTES + off) = r[sidx][i];
                }
            }
        }
    }
};

// k-contiguous fragment: rows row0+(l&15), k slots (l>>4)*8 .. +7
__device__ __forceinline__ bf16x8 frag_n(const unsigned char* lds, int row, int g) {
    return *(const bf16x8*)(lds + row * 64 + ((g ^ swz4(row)) << 4));
}

// k-slow fragments through the LDS transpose read.  NF fragments (16 rows each, starting at 32-B
// chunk c32_0 + f); the builtin leaves the wait placement to the compiler.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
template <int NF>
__device__ __forceinline__ void frag_t(const unsigned char* lds, int c32_0, int lane, bf16x8* out) {
    const int g = lane >> 4, i = lane & 15;
    const int k = g * 8 + (i >> 2);
    const int sw = (g & 1) << 2;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const unsigned char* a = lds + k * T_PITCH + (((c32_0 + f) ^ sw) << 5) + ((i & 3) << 3);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)a);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(a + 4 * T_PITCH));
        out[f] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// ---- epilogue: 8 consecutive columns n..n+7 of row m (16-B bf16 / 32-B fp32 accesses) -----------
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void unpack8(uint4 u, float* h) {
    h[0] = bf2f((bf16_t)(u.x & 0xffff)); h[1] = bf2f((bf16_t)(u.x >> 16));
    h[2] = bf2f((bf16_t)(u.y & 0xffff)); h[3] = bf2f((bf16_t)(u.y >> 16));
    h[4] = bf2f((bf16_t)(u.z & 0xffff)); h[5] = bf2f((bf16_t)(u.z >> 16));
    h[6] = bf2f((bf16_t)(u.w & 0xffff)); h[7] = bf2f((bf16_t)(u.w >> 16));
}

__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, float4 lo, float4 hi, int zsplit) {
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (p.bias) {
        const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    switch (p.epi) {
        case TULIP_EPI_BF16: {
            *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = pack8(v);
        } break;
        case TULIP_EPI_GELU_DUAL: {
            const uint4 hb = pack8(v);
            float h[8], g[8];
            unpack8(hb, h);
#pragma unroll
            for (int r = 0; r < 8; r += 2) {                       // gelu of the *stored* (rounded) h
                const f32x2 y = gelu_exact2((f32x2){h[r], h[r + 1]});
                g[r] = y.x; g[r + 1] = y.y;
            }
            *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = hb;
            *(uint4*)((bf16_t*)p.out2 + (size_t)m * p.ldo2 + n) = pack8(g);
        } break;
        case TULIP_EPI_GELU_BWD: {
            float h[8];
            unpack8(*(const uint4*)((const bf16_t*)p.aux + (size_t)m * p.ldaux + n), h);
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const f32x2 d = gelu_exact_grad2((f32x2){h[r], h[r + 1]});
                v[r] *= d.x; v[r + 1] *= d.y;
            }
            *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = pack8(v);
        } break;
        case TULIP_EPI_F32: {
            float* o = (float*)p.out + (size_t)m * p.ldo + n;
            if (p.accumulate) {
                const float4 q0 = *(const float4*)o, q1 = *(const float4*)(o + 4);
                v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
                v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
            }
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            // ldo2 > 0: a bf16 copy for the next GEMM on the path (out2 with ldo2 == 0 is the wgrad row-sum output),
            // times the DropPath scale of the branch it enters
            if (p.out2 && p.ldo2 > 0) {
                const float s = p.rowscale ? p.rowscale[fast_div(m, p.rows_per_sample)] : 1.0f;
                const float r[8] = {v[0] * s, v[1] * s, v[2] * s, v[3] * s, v[4] * s, v[5] * s, v[6] * s, v[7] * s};
                *(uint4*)((bf16_t*)p.out2 + (size_t)m * p.ldo2 + n) = pack8(r);
            }
        } break;
        case TULIP_EPI_RESID_F32: {
            const float s = p.rowscale ? p.rowscale[fast_div(m, p.rows_per_sample)] : 1.0f;
            const float* a = (const float*)p.aux + (size_t)m * p.ldaux + n;
            const float4 q0 = *(const float4*)a, q1 = *(const float4*)(a + 4);
            float* o = (float*)p.out + (size_t)m * p.ldo + n;
            const float r[8] = {q0.x + s * v[0], q0.y + s * v[1], q0.z + s * v[2], q0.w + s * v[3],
                                q1.x + s * v[4], q1.y + s * v[5], q1.z + s * v[6], q1.w + s * v[7]};
            *(float4*)o = make_float4(r[0], r[1], r[2], r[3]);
            *(float4*)(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
            if (p.out2 && p.ldo2 > 0) *(uint4*)((bf16_t*)p.out2 + (size_t)m * p.ldo2 + n) = pack8(r);
        } break;
        case TULIP_EPI_PIXSHUF2_F32: {
            // token m=(b*H+h)*W+w, column n=4c+2i+j  ->  out[b, 2h+i, 2w+j, c], C_out = N/4
            const int t = fast_div(m, p.psW), w = m - t * p.psW;
            const int b = fast_div(t, p.psH), h = t - b * p.psH;
            const int co = p.N >> 2;
            float* o = (float*)p.out;               // fp32 (B,2H,2W,C_out), optional
            bf16_t* o2 = (bf16_t*)p.out2;           // bf16 with row pitch ldo2 (e.g. the first half of a concat buffer)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = (n + r) >> 2, i = (r >> 1) & 1, j = r & 1;
                const size_t tok = ((size_t)b * 2 * p.psH + 2 * h + i) * (2 * p.psW) + 2 * w + j;
                if (o) o[tok * co + c] = v[r];
                if (o2) o2[tok * p.ldo2 + c] = f2bf(v[r]);
            }
        } break;
        case TULIP_EPI_UNSHUF2_BF16: {
            // fine token m=(b*2H+2h+i)*2W+2w+j, column n=c  ->  out[(b*H+h)*W+w][4c+2i+j], row pitch ldo
            const int W2 = 2 * p.psW, H2 = 2 * p.psH;
            const int t = fast_div(m, W2), wf = m - t * W2;
            const int b = fast_div(t, H2), hf = t - b * H2;
            bf16_t* o = (bf16_t*)p.out + (((size_t)b * p.psH + (hf >> 1)) * p.psW + (wf >> 1)) * p.ldo + 2 * (hf & 1) + (wf & 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) o[4 * (n + r)] = f2bf(v[r]);
        } break;
        case TULIP_EPI_SPLIT_F32: {
            // split-K partial slab: out is [splits][M][ldo], folded by tulip_reduce_rows2 / splitk_epilogue
            float* o = (float*)p.out + ((size_t)zsplit * p.M + m) * p.ldo + n;
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } break;
        default: break;
    }
}

// The staged fp32 tile (ROWS x 96 at row pitch STG_PITCH, first row = global row mrow0) -> memory through the epilogues: contiguous
// 8-column chunks with consecutive lanes along the row (16-B bf16 / 32-B fp32 per lane) instead of 8-B pieces scattered over 16 rows
// per instruction -- the MFMA-layout stores were the bottleneck of every output-heavy GEMM here.
constexpr int STG_PITCH = BN * 4 + 16;                 // fp32 staging row pitch (bank-spread)
template <int ROWS>
__device__ __forceinline__ void write_out_staged(const GemmArgs& p, const unsigned char* smem, const int mrow0, const int n0, const int bz,
                                                 const int tid) {
    if (TULIP_GEMM_FAST_SHUF && p.epi == TULIP_EPI_PIXSHUF2_F32 && (p.N & 31) == 0 && p.vec_ok) {                        // (uniform)
        // PixelShuffle(2) write-out, vectorised: an item is (row, sub-position q = 2i + j, 8 output channels) -- its eight values
        // sit 16 B apart in the staged row (columns 4c + q) and leave as ONE 16-byte bf16 store (+ two fp32 ones) into the fine
        // token (2h + i, 2w + j); epilogue8's form of it is eight scattered 2-byte stores per thread
#pragma unroll
        for (int it = 0; it < (ROWS * (BN / 8) + 255) / 256; ++it) {
            const int c = tid + it * 256;
            const int rl = c / (BN / 8), c8 = c - rl * (BN / 8);
            const int q = c8 & 3, cg = c8 >> 2;
            const int m = mrow0 + rl, nb = n0 + cg * 32;
            if (rl < ROWS && m < p.M && nb < p.N) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *(const float*)(smem + rl * STG_PITCH + (cg * 32 + 4 * k + q) * 4);
                if (p.bias) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += p.bias[nb + 4 * k + q];
                }
                const int t = fast_div(m, p.psW), w = m - t * p.psW;
                const int b = fast_div(t, p.psH), h = t - b * p.psH;
                const size_t tok = ((size_t)b * 2 * p.psH + 2 * h + (q >> 1)) * (2 * p.psW) + 2 * w + (q & 1);
                const int cb = nb >> 2;
                if (p.out) {
                    float* o = (float*)p.out + tok * (p.N >> 2) + cb;
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (p.out2) *(uint4*)((bf16_t*)p.out2 + tok * p.ldo2 + cb) = pack8(v);
            }
        }
    } else if (TULIP_GEMM_FAST_SHUF && p.epi == TULIP_EPI_UNSHUF2_BF16 && !p.bias && (p.M & 1) == 0 && p.vec_ok) {     // (uniform)
        // inverse shuffle: fine tokens 2w, 2w + 1 of one row (staged rows rl, rl + 1) are neighbours in the coarse token's
        // channel quadruple -- an item is (row pair, 8 columns): eight 4-byte stores instead of sixteen 2-byte ones
#pragma unroll
        for (int it = 0; it < (ROWS / 2 * (BN / 8) + 255) / 256; ++it) {
            const int c = tid + it * 256;
            const int rp = c / (BN / 8), c8 = c - rp * (BN / 8);
            const int rl = 2 * rp, m = mrow0 + rl, n = n0 + c8 * 8;
            if (rp < ROWS / 2 && m < p.M && n < p.N) {
                const float4 a0 = *(const float4*)(smem + rl * STG_PITCH + c8 * 32), a1 = *(const float4*)(smem + rl * STG_PITCH + c8 * 32 + 16);
                const float4 b0 = *(const float4*)(smem + (rl + 1) * STG_PITCH + c8 * 32), b1 = *(const float4*)(smem + (rl + 1) * STG_PITCH + c8 * 32 + 16);
                const int W2 = 2 * p.psW, H2 = 2 * p.psH;
                const int t = fast_div(m, W2), wf = m - t * W2;
                const int b = fast_div(t, H2), hf = t - b * H2;
                bf16_t* o = (bf16_t*)p.out + (((size_t)b * p.psH + (hf >> 1)) * p.psW + (wf >> 1)) * p.ldo + 2 * (hf & 1);
                const float va[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 8; ++r) *(uint32_t*)(o + 4 * (n + r)) = pack_bf16x2(va[r], vb[r]);
            }
        }
    } else {
#pragma unroll
    for (int it = 0; it < (ROWS * (BN / 8) + 255) / 256; ++it) {
        const int c = tid + it * 256;
        const int rl = c / (BN / 8), c8 = c - rl * (BN / 8);
        const int m = mrow0 + rl, n = n0 + c8 * 8;
        if (rl < ROWS && m < p.M && n < p.N) {
            const float4 lo = *(const float4*)(smem + rl * STG_PITCH + c8 * 32);
            const float4 hi = *(const float4*)(smem + rl * STG_PITCH + c8 * 32 + 16);
            epilogue8(p, m, n, lo, hi, bz);
        }
    }
    }
}

// one output tile (bx, by) of K range bz: the body of both the plain and the grouped launch
template <int BM, bool A_T, bool B_T, int KSUB, bool FULL = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, const int bx, const int by, const int bz) {
    constexpr int FM = BM / 32;  // 16-row fragments per wave in M
    constexpr int FN = 3;
    using SA = Stage<BM, A_T, KSUB, FULL>;
    using SB = Stage<BN, B_T, KSUB, FULL>;
    constexpr int A_BYTES = KSUB * SA::SUB_BYTES, B_BYTES = KSUB * SB::SUB_BYTES;
    constexpr int BKS = BK * KSUB;  // k depth of one pipeline stage
    constexpr int STG_BYTES = 64 * STG_PITCH;              // 64 output rows per write-out pass
    constexpr int PIPE_BYTES = 2 * (A_BYTES + B_BYTES);
    __shared__ __attribute__((aligned(16))) unsigned char smem[PIPE_BYTES > STG_BYTES ? PIPE_BYTES : STG_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char warm_sink[1024];      // where the panel touch drops its data
    auto ldsA = [&](int buf) -> unsigned char* { return smem + buf * (A_BYTES + B_BYTES); };
    auto ldsB = [&](int buf) -> unsigned char* { return smem + buf * (A_BYTES + B_BYTES) + A_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + BKS - 1) / BKS;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Register prefetch ring: RING-1 stages of global loads are in flight while one stage is computed
    // (PMC: with a single prefetch stage the waves sat 62-73 % of their cycles in s_waitcnt / s_barrier).
    // Ring slots are selected with compile-time indices (the loop is unrolled by RING) so the staging
    // registers never spill to scratch; the compiler's counted vmcnt keeps the younger stage in flight
    // while the older one is written to LDS.
    // (a deeper ring for the weight-gradient form, K = thousands of tokens per workgroup, measured no faster: 4 stages
    // 31.7 us, 6 stages 32.3 us vs 31.3 us per grouped launch, and costs occupancy)
    constexpr int RING = (KSUB == 1) ? ((A_T && B_T) ? TULIP_WGRAD_RING : 3) : TULIP_GEMM_DEEP_RING;
    SA sa[RING];
    SB sb[RING];
    auto issue = [&](auto R, int t) {
        constexpr int r = decltype(R)::value;
        sa[r].load(p.A, p.lda, m0, p.M, kbeg + t * BKS, kend, tid);
        sb[r].load(p.B, p.ldb, n0, p.N, kbeg + t * BKS, kend, tid);
    };
    // Cold weights.  In the training step a Linear's weights were last read a step ago; the k loop below has two or three
    // stages in flight and so walks its [BN][k range] weight panel at miss latency.  The M-tile workgroups of one N panel
    // (blockIdx.y; on one XCD, i.e. one L2, when gridDim.x % 8 == 0) first split the panel between them and touch it with
    // loads whose data is dropped into a KiB of LDS nobody reads (warm_touch16, common.h: LDS-destination loads the
    // compiler tracks itself; see WeightWarm in swinw.hip): everything is in flight at once and the loop's own loads hit
    // L2.  The dropped loads are older than stage 0's, so they have retired when stage 0 is written to LDS.
    if constexpr (!A_T) {
        if (p.touch) {
            const int rowsN = min(BN, p.N - n0), kw = kend - kbeg;
            // 16-byte pieces of the panel: B_T: [k][n] rows of rowsN elements; else [n][k] rows of kw elements
            const int ppr = ((B_T ? rowsN : kw) * 2) >> 4, prow = B_T ? kw : rowsN;
            const int npieces = ppr * prow;
            const unsigned char* base = (const unsigned char*)(B_T ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg);
            for (int q = (by * 4 + wid) * 64 + lane; q < npieces; q += (int)gridDim.y * 256) {
                const int r = q / ppr, c16 = q - r * ppr;
                warm_touch16(base + ((size_t)r * p.ldb) * 2 + c16 * 16, warm_sink);
            }
        }
    }
    static_for<RING - 1>([&](auto R) { if (nt > decltype(R)::value) issue(R, decltype(R)::value); });
    if (nt > 0) {
        sa[0].store(ldsA(0), tid);
        sb[0].store(ldsB(0), tid);
    }
    __syncthreads();

    const int g = lane >> 4, li = lane & 15;
    // wgrad only: row sums of opA (= bias gradient, sum over tokens of dY) from one extra MFMA per
    // fragment against an all-ones operand, in the first column-tile's wn==0 waves.
    const bool do_rowsum = A_T && p.out2 != nullptr && bx == 0 && wn == 0 &&
                           (p.epi == TULIP_EPI_SPLIT_F32 || p.epi == TULIP_EPI_F32);
    f32x4 rsum[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) rsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const short one = (short)0x3F80;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};

    auto step = [&](auto R, int t) {
        constexpr int r = decltype(R)::value;
        const int cur = t & 1;
        if (t + RING - 1 < nt) issue(std::integral_constant<int, (r + RING - 1) % RING>{}, t + RING - 1);
        const int ksub_valid = min(KSUB, (kend - (kbeg + t * BKS) + BK - 1) / BK);  // block-uniform
#pragma unroll
        for (int sidx = 0; sidx < KSUB; ++sidx) {
            if (KSUB > 1 && sidx >= ksub_valid) break;
            const unsigned char* la = ldsA(cur) + sidx * SA::SUB_BYTES;
            const unsigned char* lb = ldsB(cur) + sidx * SB::SUB_BYTES;
            bf16x8 af[FM], bfr[FN];
            if (!A_T) {
#pragma unroll
                for (int f = 0; f < FM; ++f) af[f] = frag_n(la, wm * (BM / 2) + f * 16 + li, g);
            }
            if (!B_T) {
#pragma unroll
                for (int f = 0; f < FN; ++f) bfr[f] = frag_n(lb, wn * 48 + f * 16 + li, g);
            }
            if (A_T) frag_t<FM>(la, wm * (BM / 32), lane, af);
            if (B_T) frag_t<FN>(lb, wn * 3, lane, bfr);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (A_T && do_rowsum) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    rsum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], rsum[i], 0, 0, 0);
            }
        }
        if (t + 1 < nt) {
            sa[(r + 1) % RING].store(ldsA(cur ^ 1), tid);
            sb[(r + 1) % RING].store(ldsB(cur ^ 1), tid);
        }
        __syncthreads();
    };
    for (int t = 0; t < nt; t += RING)
        static_for<RING>([&](auto R) { if (t + decltype(R)::value < nt) step(R, t + decltype(R)::value); });

    // Write-out: the accumulator tile goes through LDS so that the epilogue runs on contiguous 8-column
    // chunks with consecutive lanes along the row (16-B bf16 / 32-B fp32 per lane, full 64-B+ segments per
    // row) instead of 8-B pieces scattered over 16 rows per instruction -- the MFMA-layout stores were the
    // bottleneck of every output-heavy GEMM here.  64 rows per pass (the fp32 tile is 25 KiB).
    constexpr int NPASS = BM / 64;
    constexpr int PPW = NPASS > 1 ? NPASS / 2 : 1;      // passes per wave row: a pass holds 64 rows = 4 fragments of ONE wave row
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (NPASS == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int rl = wm * (BM / 2) + i * 16 + li;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *(f32x4*)(smem + rl * STG_PITCH + (wn * 48 + j * 16 + g * 4) * 4) = acc[i][j];
            }
        } else if (wm == ps / PPW) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int rl = ii * 16 + li;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *(f32x4*)(smem + rl * STG_PITCH + (wn * 48 + j * 16 + g * 4) * 4) = acc[(ps % PPW) * 4 + ii][j];
            }
        }
        __syncthreads();
        write_out_staged<64>(p, smem, m0 + ps * 64, n0, bz, tid);
        if (ps + 1 < NPASS) __syncthreads();
    }
    if (A_T && do_rowsum && g == 0) {
        float* rs = (float*)p.out2;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + li;
            if (m >= p.M) continue;
            if (p.epi == TULIP_EPI_SPLIT_F32) rs[(size_t)bz * p.M + m] = rsum[i][0];
            else if (p.accumulate & 1) rs[m] += rsum[i][0];
            else rs[m] = rsum[i][0];
        }
    }
}

template <int BM, bool A_T, bool B_T, int KSUB>
// (two waves per SIMD promised where the LDS footprint allows it: the compiler then keeps the accumulators in VGPRs --
// MFMAs with AGPR accumulators issue at ~60 % of the rate, tools/probe_mfma.hip; the 128-deep k stages take 80-150 KB of LDS)
__global__ __launch_bounds__(256, (KSUB == 1 ? 2 : 1)) void gemm_kernel(const GemmArgs p) {
    gemm_tile<BM, A_T, B_T, KSUB>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}
// whole tiles only (Stage<..., FULL>): the forward / data-gradient form -- what the GEMMs of the bench configurations launch
template <int BM, bool B_T, int KSUB>
__global__ __launch_bounds__(256, (KSUB == 1 ? 2 : 1)) void gemm_kernel_full(const GemmArgs p) {
    gemm_tile<BM, false, B_T, KSUB, true>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent GEMMs in ONE launch (the weight gradients of a Swin block: each alone fills a fraction of the
// chip and costs a launch on the side queue).  Workgroups [first[i], first[i+1]) belong to problem i.
// ---------------------------------------------------------------------------------------------------------------------
// The small-K form (TULIP_GEMM_B_PACKED): a 32 x 96 output tile whose whole K range (KS 32-deep steps, K <= 1536 per split) is in
// flight at once.  B is the FRAGMENT-MAJOR copy of the [N][K] matrix (tulip_pack_bf16_multi; csrc/swin_stream.h): the operand of
// one MFMA is one contiguous 1-KiB wave load straight into registers, up to eight steps ahead and issued before anything else (the
// weights are the cold bytes); the 32 x K panel of A is fetched with every load issued before the first LDS write.  One latency
// instead of K / 128 dependent stages: the stage-boundary GEMMs of the batch-8 step (512 .. 2 048 rows, K <= 768) were 8-13 us of
// launch + dependent round trips on 128 workgroups.  Same MFMA order per output element as gemm_tile (k steps in sequence, weights
// as the first operand): the same bits.
template <int KS>
__global__ __launch_bounds__(256) void gemm_stream_kernel(const GemmArgs p) {
    constexpr int BMS = 32, NTW = 3, PF = KS < 8 ? KS : 8;
    constexpr int PANEL = KS * BMS * 64, STG = BMS * STG_PITCH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PANEL > STG ? PANEL : STG];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BMS, bz = blockIdx.z, kbeg = bz * p.kchunk;
    const bf16_t* wt[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
        wt[j] = p.B + ((size_t)((n0 >> 4) + wn * NTW + j) * (p.K >> 5) + (kbeg >> 5)) * 512 + lane * 8;
    // Cold weights: in the step the copies were written at the end of the previous one and come from HBM; a ring of PF steps would walk
    // them at miss latency (the fused boundary launches of csrc/glue.hip were worth 6 us of the step before their warm-up and 42 with
    // it).  The workgroup first touches its [96][K range] slice beyond the ring's first PF steps -- each KiB one LDS-destination wave
    // load whose data is dropped (warm_touch16) --, everything in flight at once; the stream below then hits L2.
    __shared__ __attribute__((aligned(16))) unsigned char warm_sink[1024];
    if (p.touch && KS > PF) {
        constexpr int KR = KS - PF, NB = 2 * NTW * KR;        // (the first PF steps are the ring's own first loads, issued right below)
#pragma unroll
        for (int i = 0; i < (NB + 3) / 4; ++i) {
            const int blk = wid + 4 * i;                       // (tile, k step) pairs of the slice, one KiB each
            if (NB % 4 == 0 || blk < NB) {
                const int tile = blk / KR, ks = PF + blk - tile * KR;
                warm_touch16(p.B + ((size_t)((n0 >> 4) + tile) * (p.K >> 5) + (kbeg >> 5) + ks) * 512 + lane * 8, warm_sink);
            }
        }
    }
    bf16x8 ring[PF][NTW];
    static_for<PF>([&](auto Q_) {
        constexpr int q = decltype(Q_)::value;
#pragma unroll
        for (int j = 0; j < NTW; ++j) ring[q][j] = *(const bf16x8*)(wt[j] + 512 * q);
    });
    // the A panel: [k tile][32 rows][64 B], 16-B chunks XOR-swizzled by row (frag_n's layout)
    constexpr int CPR = KS * 4, NCH = BMS * CPR, PER = (NCH + 255) / 256;
    uint4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * 256;
        v[i] = make_uint4(0, 0, 0, 0);
        if (NCH % 256 == 0 || c < NCH) {
            const int row = c / CPR, kc = c - row * CPR;
            v[i] = *(const uint4*)(p.A + (size_t)(m0 + row) * p.lda + kbeg + kc * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * 256;
        if (NCH % 256 == 0 || c < NCH) {
            const int row = c / CPR, kc = c - row * CPR;
            *(uint4*)(smem + (kc >> 2) * (BMS * 64) + row * 64 + (((kc & 3) ^ swz4(row)) << 4)) = v[i];
        }
    }
    __syncthreads();
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<KS>([&](auto K_) {
        constexpr int ks = decltype(K_)::value;
        bf16x8 a[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) a[j] = ring[ks % PF][j];
        if constexpr (ks + PF < KS) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) ring[ks % PF][j] = *(const bf16x8*)(wt[j] + 512 * (ks + PF));
        }
        const bf16x8 x = frag_n(smem + ks * (BMS * 64), wm * 16 + li, g);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], x, acc[j], 0, 0, 0);
    });
    __syncthreads();                                   // the panel is dead: its LDS becomes the staged output tile
#pragma unroll
    for (int j = 0; j < NTW; ++j) *(f32x4*)(smem + (wm * 16 + li) * STG_PITCH + (wn * 48 + j * 16 + g * 4) * 4) = acc[j];
    __syncthreads();
    write_out_staged<BMS>(p, smem, m0, n0, bz, tid);
}
static bool stream_shape_ok(int M, int N, int K, int kchunk) {
    return M % 32 == 0 && N % BN == 0 && K % kchunk == 0 && (kchunk == 96 || kchunk == 384 || kchunk == 768 || kchunk == 1536);
}

constexpr int GROUP_MAX = TULIP_WGRAD_GROUP_MAX;
struct GemmGroup {
    GemmArgs g[GROUP_MAX];
    int first[GROUP_MAX + 1];
    int gx[GROUP_MAX], gy[GROUP_MAX];
    int n;
};
template <int BM, bool A_T, bool B_T, int KSUB>
__global__ __launch_bounds__(256, (KSUB == 1 ? 2 : 1)) void gemm_group_kernel(const GemmGroup G) {
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.first[i + 1]) ++i;
    int b = blockIdx.x - G.first[i];
    const int bx = b % G.gx[i];
    b /= G.gx[i];
    const GemmArgs p = G.g[i];
    gemm_tile<BM, A_T, B_T, KSUB>(p, bx, b % G.gy[i], b / G.gy[i]);
}

// ---- weight gradients, large tiles ------------------------------------------------------------------------------
// dW[N,K] = dY^T . X over a chunk of tokens, both operands token-major (the A_T, B_T form).  The 64 x 96 tile of gemm_tile
// re-reads every operand element N/64 resp. K/96 times through the texture path and issues 10 transpose reads for 6 MFMAs
// per wave and k-step; here a workgroup owns (GM*96) x (GN*96) outputs, one 96 x 96 tile per wave: 36 accumulator
// fragments in registers (one wave per SIMD), 24 transpose reads for 36 MFMAs, and a third of the operand traffic.
// GM x GN = 2 x 2 (192 x 192: every linear of stages 1-3), 4 x 1 (384 x 96) and 1 x 4 (96 x 384) for the C = 96 stage.
// Each k-step's operands are [32 tokens][96 columns] sub-tiles in the k-slow LDS layout of Stage<96, true> (pitch 288 B).
// AdamW in the write-out of an un-split weight gradient (tulip_wgrad_group_adamw): the flat fp32 gradient / parameter / moment
// buffers share one layout, so an element's parameter, moments and bf16 shadow sit at the gradient's own offset from g0
constexpr int WG_SUB = 32 * T_PITCH;          // bytes of one sub-tile
constexpr int WG_STG_PITCH = 96 * 4 + 16;     // fp32 write-out staging row
constexpr int WG_LDS_BYTES = 2 * 5 * WG_SUB;  // double-buffered 4 x 1 stage (the 2 x 2 stage is 4 sub-tiles)

#ifndef TULIP_WGRAD_STORE_NT
#define TULIP_WGRAD_STORE_NT 0
#endif
#ifndef TULIP_WGRAD_LOAD_NT
#define TULIP_WGRAD_LOAD_NT 1      // operand stream non-temporal: same-box A/B of the step 1.996 vs 2.014 ms (batch 8), batch 64 flat -- it ran the fused wide blocks' weights out of L2
#endif
template <int GM, int GN, int RING>
__device__ __forceinline__ void wgrad_tile(const GemmArgs& p, const int bx, const int by, const int bz,
                                           unsigned char* __restrict__ smem, const AdamRef& ad) {
    static_assert(GM * GN == 4, "four compute waves");
    constexpr int SUBS = GM + GN;
    constexpr int STAGE = SUBS * WG_SUB;
    constexpr bool A_FIRST = GM >= GN;                 // the wider operand first: its chunk count is a multiple of 256
    constexpr int G1 = A_FIRST ? GM : GN, G2 = A_FIRST ? GN : GM;
    constexpr int NCH = SUBS * 384;                    // 16-B chunks of one k-step
    constexpr int PT = (NCH + 255) / 256;
    constexpr int I1 = G1 * 384 / 256;                 // chunk slots (per thread) of the first operand
    // eight waves, two per SIMD: waves 0-3 own the 96 x 96 tiles (LDS fragment reads + MFMAs), waves 4-7 move the
    // operands (global -> registers -> LDS).  A wave issues in order, and a global load waits at issue while the CU's
    // address unit works through the ~24 KB a k-step fetches (~850 cycles per step measured with the MFMAs removed,
    // more than the 36 MFMAs take): in one wave the two add up (1500 cycles per step); in two waves of one SIMD the
    // matrix pipe runs while the loader is held.
    const int wid8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wid8 >= 4;
    const int tid = threadIdx.x & 255, lane = tid & 63, wid = wid8 & 3;
    const int wm = wid / GN, wn = wid % GN;
    const int m0 = by * (GM * 96), n0 = bx * (GN * 96);
    const int kbeg = bz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + 31) >> 5;
    // p.aux (unused by this form): optional profile buffer, 4 shader-clock stamps per workgroup (tools/wgrad_phases.py)
    unsigned long long* prof = (unsigned long long*)p.aux;
#if TULIP_DEV_VARIANTS
#define TULIP_WG_STAMP(k) do { if (prof && threadIdx.x == 0) prof[(size_t)blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TULIP_WG_STAMP(k) ((void)0)
#endif
    TULIP_WG_STAMP(0);
    if (nt <= 0) return;                               // (uniform; the launcher never creates an empty chunk)

    constexpr int NSTEP_UNROLL = RING;
    if (loader) {
        // per-thread chunk slots: byte offset inside a k-step's operand rows (32-bit, added to a wave-uniform base that
        // advances by 32 token rows per k-step) and the LDS offset.  Everything in the k-loop is unconditional: the token
        // count is a multiple of 32 (the launcher's condition for this kernel); sub-tiles beyond M / N are never read (M,
        // N are multiples of 96, only inactive waves own them), so their slots load the k-step's first chunk; the slots
        // past the end of the chunk list (4 x 1 / 1 x 4: 7.5 per thread) write into the unused last 32 B of a sub-tile row.
        unsigned goff[PT];
        int loff[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const bool first = i < I1;
            const bool isA = first == A_FIRST;
            const int G = first ? G1 : G2;
            const int cc = tid + i * 256 - (first ? 0 : G1 * 384);
            const int k = cc / (12 * G), mc = cc - k * (12 * G);
            const int sub = mc / 12, mcs = mc - sub * 12;
            const int col = (isA ? m0 : n0) + mc * 8;
            const bool real = first || cc < G2 * 384;
            const bool ok = real && col < (isA ? p.M : p.N);
            goff[i] = ok ? (unsigned)(k * (isA ? p.lda : p.ldb) + col) * 2u : 0u;
            loff[i] = real ? ((isA ? sub : GM + sub) * WG_SUB) + k * T_PITCH + ((((mcs >> 1) ^ (((k >> 3) & 1) << 2))) << 5) +
                                 ((mcs & 1) << 4)
                           : (tid & 31) * T_PITCH + 256 + ((tid >> 5) & 1) * 16;
        }
        const unsigned char* baseA = (const unsigned char*)(p.A + (size_t)kbeg * p.lda);
        const unsigned char* baseB = (const unsigned char*)(p.B + (size_t)kbeg * p.ldb);
        // (a native vector type: a struct uint4 assigned straight from memory becomes a memcpy into the array, which
        // keeps the whole ring in scratch)
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t ring[RING][PT];
        auto issue = [&](auto R, int t) {
            constexpr int r = decltype(R)::value;
            const unsigned char* a = baseA + (size_t)t * 64 * p.lda;
            const unsigned char* b = baseB + (size_t)t * 64 * p.ldb;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
#if TULIP_WGRAD_LOAD_NT
                ring[r][i] = __builtin_nontemporal_load((const u32x4_t*)(((i < I1) == A_FIRST ? a : b) + goff[i]));
#else
                ring[r][i] = *(const u32x4_t*)(((i < I1) == A_FIRST ? a : b) + goff[i]);
#endif
            }
        };
        auto stash = [&](auto R, unsigned char* dst) {
            constexpr int r = decltype(R)::value;
#pragma unroll
            for (int i = 0; i < PT; ++i) *(u32x4_t*)(dst + loff[i]) = ring[r][i];
        };
        // (the global loads are issued unconditionally -- past the end of the chunk they re-read its last k-step into a
        // slot nobody stashes: a branch around them makes the compiler's counted vmcnt waits collapse to vmcnt(0))
        static_for<RING>([&](auto R) { issue(R, min((int)decltype(R)::value, nt - 1)); });
        stash(std::integral_constant<int, 0>{}, smem);
        __syncthreads();
        if (nt > 1) stash(std::integral_constant<int, 1>{}, smem + STAGE);
        __syncthreads();
        // step t: load k-step t+RING into the slot whose k-step t went to LDS two steps ago; k-step t+2 goes into the
        // buffer whose fragments the compute waves read during step t-1
        auto step = [&](auto R, int t) {
            constexpr int r = decltype(R)::value;
            issue(R, min(t + RING, nt - 1));
            if (t + 2 < nt) stash(std::integral_constant<int, (r + 2) % RING>{}, smem + (t & 1) * STAGE);
            __syncthreads();
        };
        for (int t = 0; t < nt; t += NSTEP_UNROLL)
            static_for<NSTEP_UNROLL>([&](auto R) { if (t + decltype(R)::value < nt) step(R, t + decltype(R)::value); });
        return;
    }

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 rsum = (f32x4){0.f, 0.f, 0.f, 0.f};          // bias gradient: row sums of the six A fragments, one per MFMA column

    const int g = lane >> 4, li = lane & 15;
    const int fbase = (g * 8 + (li >> 2)) * T_PITCH + ((li & 3) << 3);
    const int fsw = (g & 1) << 7;                      // the 32-B chunk swizzle of the odd 8-row groups
    const bool active = m0 + wm * 96 < p.M && n0 + wn * 96 < p.N;      // wave-uniform
    const bool do_rowsum = active && p.out2 != nullptr && bx == 0 && wn == 0;
    const short one = (short)0x3F80;
    const bf16x8 ones = {one, one, one, one, one, one, one, one}, zeros = {0, 0, 0, 0, 0, 0, 0, 0};

    // compute waves, step t: issue the transpose reads of k-step t+1's A fragments into the second set, then the 36
    // MFMAs of k-step t, which run while the reads are in flight; each B fragment of k-step t+1 is read right behind the
    // last MFMA that uses its predecessor (two full fragment sets + 36 accumulators do not fit 256 registers, the
    // budget of a wave when two share a SIMD).  One barrier per step, shared with the loaders.
    bf16x8 fa[2][6], fb[6];
    auto tr2 = [&](const unsigned char* a) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)a);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(a + 4 * T_PITCH));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto readA = [&](auto S, const unsigned char* stage) {
        const unsigned char* la = stage + wm * WG_SUB + fbase;
#pragma unroll
        for (int f = 0; f < 6; ++f) fa[decltype(S)::value][f] = tr2(la + ((f << 5) ^ fsw));
    };
    __syncthreads();
    if (active) {
        readA(std::integral_constant<int, 0>{}, smem);
#pragma unroll
        for (int f = 0; f < 6; ++f) fb[f] = tr2(smem + (GM + wn) * WG_SUB + fbase + ((f << 5) ^ fsw));
    }
    __syncthreads();
    TULIP_WG_STAMP(1);

    auto step = [&](auto R, int t) {
        constexpr int r = decltype(R)::value;
        const unsigned char* nxt = smem + ((t & 1) ^ 1) * STAGE;
        if (active) {
            const bool more = t + 1 < nt;
            if (more) readA(std::integral_constant<int, (r + 1) & 1>{}, nxt);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[r & 1][i], acc[i][j], 0, 0, 0);
            // second half column by column: B fragment j is dead after its three MFMAs and is re-read for the next k-step
            // while the remaining columns still compute
            const unsigned char* lb = nxt + (GM + wn) * WG_SUB + fbase;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int i = 3; i < 6; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[r & 1][i], acc[i][j], 0, 0, 0);
                if (more) fb[j] = tr2(lb + ((j << 5) ^ fsw));
            }
            if (do_rowsum) {        // out[n][m] += sel_i[n][k] a_i[m][k], sel_i = 1 on row n == i: column i collects fragment i
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    rsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r & 1][i], li == i ? ones : zeros, rsum, 0, 0, 0);
            }
        }
        __syncthreads();
    };
    for (int t = 0; t < nt; t += 2)
        static_for<2>([&](auto R) { if (t + decltype(R)::value < nt) step(R, t + decltype(R)::value); });

    TULIP_WG_STAMP(2);
    // write-out: each wave stages 32 rows of its own tile at a time and stores them as full 384-B rows
    if (!active) return;
    unsigned char* wst = smem + wid * (32 * WG_STG_PITCH);
    const bool split = p.epi == TULIP_EPI_SPLIT_F32;
    float* obase = (float*)p.out + (split ? (size_t)bz * p.M * p.ldo : 0);
    const bool step_here = !split && (p.accumulate & 2);               // (uniform) AdamW in the write-out, see AdamRef
    const AdamwCoef cf = step_here ? adamw_coef(ad.hyper, true) : AdamwCoef{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *(f32x4*)(wst + (ii * 16 + li) * WG_STG_PITCH + (j * 16 + g * 4) * 4) = acc[ps * 2 + ii][j];
        if (step_here) {
            // the optimizer step of this tile right here (adamw_step4, common.h: the same operations as adamw_kernel; weights
            // always decay): the gradient is never stored, the end-of-step AdamW skips the tensor.  Parameter and moments of FOUR
            // pieces are fetched before the first is stepped -- twelve loads in flight per lane instead of three: one piece at a
            // time the write-out was a chain of 36 memory latencies per wave (the deep stage's group 243 us in the step).
#pragma unroll
            for (int it0 = 0; it0 < 12; it0 += 4) {
                float4 pp[4], mm[4], vv[4];
                size_t idx[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = lane + (it0 + u) * 64;
                    const int rl = c / 24, c4 = c - rl * 24;
                    const int m = m0 + wm * 96 + ps * 32 + rl, n = n0 + wn * 96 + c4 * 4;
                    ok[u] = m < p.M && n < p.N;
                    idx[u] = ok[u] ? (size_t)((obase + (size_t)m * p.ldo + n) - ad.g0) : 0;
                    pp[u] = ld_state(ad.p0 + idx[u]); mm[u] = ld_state(ad.m0 + idx[u]); vv[u] = ld_state(ad.v0 + idx[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = lane + (it0 + u) * 64;
                    const int rl = c / 24, c4 = c - rl * 24;
                    if (ok[u]) {
                        const float4 v = *(const float4*)(wst + rl * WG_STG_PITCH + c4 * 16);
                        adamw_step4(pp[u], mm[u], vv[u], v, cf);
                        st_state(ad.p0 + idx[u], pp[u]); st_state(ad.m0 + idx[u], mm[u]); st_state(ad.v0 + idx[u], vv[u]);
                        st_state_bf16x4(ad.pb0 + idx[u], make_uint2(pack_bf16x2(pp[u].x, pp[u].y), pack_bf16x2(pp[u].z, pp[u].w)));
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int c = lane + it * 64;
            const int rl = c / 24, c4 = c - rl * 24;
            const int m = m0 + wm * 96 + ps * 32 + rl, n = n0 + wn * 96 + c4 * 4;
            if (m < p.M && n < p.N) {
                float4 v = *(const float4*)(wst + rl * WG_STG_PITCH + c4 * 16);
                float* o = obase + (size_t)m * p.ldo + n;
                if (!split && (p.accumulate & 1)) {
                    const float4 q = *(const float4*)o;
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
#if TULIP_WGRAD_STORE_NT
                __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4*)o);
#else
                *(float4*)o = v;
#endif
            }
        }
    }
    TULIP_WG_STAMP(3);
    if (do_rowsum && li < 6) {
        // rsum lane (g, li), element e = sum over k of fragment li's row 4 g + e
        float* rs = (float*)p.out2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 96 + li * 16 + g * 4 + e;
            if (m >= p.M) continue;
            if (split) rs[(size_t)bz * p.M + m] = rsum[e];
            else if (p.accumulate & 1) rs[m] += rsum[e];
            else rs[m] = rsum[e];
        }
    }
}

struct WgradGroup {
    GemmArgs g[GROUP_MAX];
    int first[GROUP_MAX + 1];
    int gx[GROUP_MAX], gy[GROUP_MAX], shape[GROUP_MAX];
    int n;
    AdamRef adam;
};
__global__ __launch_bounds__(512) void wgrad_group_kernel(const WgradGroup G) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_LDS_BYTES];
    // workgroup b runs on XCD b % 8: neighbours in the (token chunk, tile) order -- tiles of one chunk share its
    // operand rows -- are given to the same XCD's L2, 8 launch slots apart
    int b = blockIdx.x;
    const int nb8 = (int)gridDim.x & ~7;
    if (b < nb8) b = (b & 7) * (nb8 >> 3) + (b >> 3);
    int i = 0;
    while (i + 1 < G.n && b >= G.first[i + 1]) ++i;
    b -= G.first[i];
    const int tiles = G.gx[i] * G.gy[i];
    const int bz = b / tiles;
    b -= bz * tiles;
    const int by = b / G.gx[i], bx = b - by * G.gx[i];
    const GemmArgs p = G.g[i];
    switch (G.shape[i]) {
        case 0: wgrad_tile<2, 2, 4>(p, bx, by, bz, smem, G.adam); break;
        case 1: wgrad_tile<4, 1, 4>(p, bx, by, bz, smem, G.adam); break;
        default: wgrad_tile<1, 4, 4>(p, bx, by, bz, smem, G.adam); break;
    }
}

// ---- forward / data-gradient GEMM, mid-size shapes (round 5) -----------------------------------------------------------
// C[M][N] = A[M][K] . opB^T for M = 1 024 .. 16 384, N, K = 384 .. 6 144 (the deep stage at batch 64, stages 3-4 of tulip_large at
// the 2048-wide grids): the 64 / 128 x 96 tiles of gemm_tile fetch every operand element N / 96 resp. M / BM times through a CU
// whose address unit moves ~20 B/clk of 64-byte row pieces and read 7 fragments from LDS for 12 MFMAs.  Here the weight-gradient
// kernel's shape: 192 x 192 outputs per workgroup, eight waves -- waves 0-3 own one 96 x 96 tile each (36 accumulator
// fragments, 12 fragment reads for 36 MFMAs per 32 k), waves 4-7 only move operands: global -> a register ring of three
// 64-deep stages (128-byte row pieces, one cache line per 8 lanes) -> double-buffered 32-deep LDS tiles.  A is k-fast
// ([M][K]); B is k-fast ([N][K]: forward, y = x W^T) or k-slow ([K][N]: data gradient, dx = dy W; the k-slow LDS layout and
// transpose reads of wgrad_tile).  Rows beyond M / N are clamped in the loads and dropped in the write-out; the K range of a
// workgroup is a multiple of 64.  Write-out: every wave stages 32 rows of its tile at a time and runs epilogue8 on 8-column
// chunks (all the epilogues of gemm_tile, split-K slabs included).
constexpr int MID_A_BYTES = 192 * 64;                     // one 32-deep k-fast operand tile: [192 rows][64 B], 16-B chunks swizzled
template <bool B_T>
struct MidGeo {
    static constexpr int B_BYTES = B_T ? 2 * WG_SUB : MID_A_BYTES;
    static constexpr int STAGE = MID_A_BYTES + B_BYTES;   // one 32-deep LDS stage
    static constexpr int STG = 4 * 32 * WG_STG_PITCH;     // write-out staging (after the k loop, same memory)
    static constexpr int LDS = 2 * STAGE > STG ? 2 * STAGE : STG;
};
#ifndef TULIP_GEMM_MID_RING
#define TULIP_GEMM_MID_RING 3
#endif
template <bool B_T>
__device__ __forceinline__ void gemm_mid_tile(const GemmArgs& p, const int bx, const int by, const int bz,
                                              unsigned char* __restrict__ smem) {
    using Z = MidGeo<B_T>;
    constexpr int STAGE = Z::STAGE;
    constexpr int NR = TULIP_GEMM_MID_RING;
    const int wid8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wid8 >= 4;
    const int tid = threadIdx.x & 255, lane = tid & 63, wid = wid8 & 3;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = by * 192, n0 = bx * 192;
    const int kbeg = bz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg) >> 5;                  // 32-deep steps: even (the launcher's condition)
    const int ns = nt >> 1;                             // 64-deep register stages
    if (ns <= 0) return;

    if (loader) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        // A (and a k-fast B): a stage is [192 rows][128 B]; slot i of a thread = chunk (tid + 256 i): row = c >> 3, 16-B piece
        // kc = c & 7 for slots 0-2 and (c & 7) ^ 4 for slots 3-5 -- 8 lanes still cover one 128-byte line, and every thread holds
        // three chunks of EACH 32-deep half: half h is slots 0-2 for the lanes whose piece index has bit 2 == h, slots 3-5 for
        // the others (a select per register, full-lane LDS writes).  k-slow B: a stage is [64 k][384 B], chunk c: k = c / 24 --
        // slots 0-2 are k < 32, slots 3-5 the second half for every lane.
        const bool lowhalf = ((tid & 7) < 4);
        unsigned goffA[6], goffB[6];
        int loffA[6], loffB[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, kc = (c & 7) ^ (i >= 3 ? 4 : 0);
            const int ra = min(m0 + row, p.M - 1);
            goffA[i] = (unsigned)ra * (unsigned)p.lda * 2u + (unsigned)kc * 16u;
            loffA[i] = row * 64 + (((kc & 3) ^ swz4(row)) << 4);
            if constexpr (!B_T) {
                const int rb = min(n0 + row, p.N - 1);
                goffB[i] = (unsigned)rb * (unsigned)p.ldb * 2u + (unsigned)kc * 16u;
                loffB[i] = MID_A_BYTES + loffA[i];
            } else {
                const int k = c / 24, mc = c - k * 24;
                const int sub = mc / 12, mcs = mc - sub * 12;
                const int col = min(n0 + mc * 8, p.N - 8);
                goffB[i] = (unsigned)k * (unsigned)p.ldb * 2u + (unsigned)col * 2u;
                const int kk = k & 31;
                loffB[i] = MID_A_BYTES + sub * WG_SUB + kk * T_PITCH + ((((mcs >> 1) ^ (((kk >> 3) & 1) << 2))) << 5) + ((mcs & 1) << 4);
            }
        }
        const unsigned char* baseA = (const unsigned char*)(p.A + kbeg);
        const unsigned char* baseB = B_T ? (const unsigned char*)(p.B + (size_t)kbeg * p.ldb) : (const unsigned char*)(p.B + kbeg);
        u32x4_t ra[NR][6], rb[NR][6];
        auto issue = [&](auto R, int s) {
            constexpr int r = decltype(R)::value;
            const unsigned char* a = baseA + (size_t)s * 128;
            const unsigned char* b = B_T ? baseB + (size_t)s * 128 * p.ldb : baseB + (size_t)s * 128;
#pragma unroll
            for (int i = 0; i < 6; ++i) ra[r][i] = *(const u32x4_t*)(a + goffA[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) rb[r][i] = *(const u32x4_t*)(b + goffB[i]);
        };
        // half h of register stage r -> the 32-deep LDS stage at dst
        auto stash = [&](auto R, int h, unsigned char* dst) {
            constexpr int r = decltype(R)::value;
            const bool first = lowhalf == (h == 0);           // this lane's chunks of half h sit in slots 0-2 (else 3-5)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const u32x4_t v = first ? ra[r][i] : ra[r][i + 3];
                *(u32x4_t*)(dst + (first ? loffA[i] : loffA[i + 3])) = v;
            }
            if constexpr (!B_T) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4_t v = first ? rb[r][i] : rb[r][i + 3];
                    *(u32x4_t*)(dst + (first ? loffB[i] : loffB[i + 3])) = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4_t v = h == 0 ? rb[r][i] : rb[r][i + 3];
                    *(u32x4_t*)(dst + (h == 0 ? loffB[i] : loffB[i + 3])) = v;
                }
            }
        };
        // (loads are issued unconditionally: past the end they re-read the last stage into a slot nobody stashes)
        static_for<NR>([&](auto R) { issue(R, min((int)decltype(R)::value, ns - 1)); });
        stash(std::integral_constant<int, 0>{}, 0, smem);
        __syncthreads();
        stash(std::integral_constant<int, 0>{}, 1, smem + STAGE);
        __syncthreads();
        // stage s (steps 2s, 2s+1): its ring slot is free (both halves went to LDS during stage s-1): refill it with stage
        // s + NR; the halves of stage s + 1 go into the LDS buffers whose fragments the compute waves read one step earlier
        auto stage = [&](auto R, int s) {
            constexpr int r = decltype(R)::value;
            issue(R, min(s + NR, ns - 1));
            if (s + 1 < ns) stash(std::integral_constant<int, (r + 1) % NR>{}, 0, smem);
            __syncthreads();
            if (s + 1 < ns) stash(std::integral_constant<int, (r + 1) % NR>{}, 1, smem + STAGE);
            __syncthreads();
        };
        for (int s = 0; s < ns; s += NR)
            static_for<NR>([&](auto R) { if (s + decltype(R)::value < ns) stage(R, s + decltype(R)::value); });
        return;
    }

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, li = lane & 15;
    const bool active = m0 + wm * 96 < p.M && n0 + wn * 96 < p.N;      // wave-uniform
    bf16x8 fa[2][6], fb[6];
    // k-fast fragment f of a 96-row operand block starting at row r0: rows r0 + 16 f + li, k slots 8 g .. 8 g + 7
    auto readN = [&](const unsigned char* tile, int r0, int f) {
        const int row = r0 + f * 16 + li;
        return *(const bf16x8*)(tile + row * 64 + ((g ^ swz4(row)) << 4));
    };
    const int fbase = (g * 8 + (li >> 2)) * T_PITCH + ((li & 3) << 3), fsw = (g & 1) << 7;
    auto readT = [&](const unsigned char* sub, int f) {
        const unsigned char* a = sub + fbase + ((f << 5) ^ fsw);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)a);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(a + 4 * T_PITCH));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto readB = [&](const unsigned char* stage, int f) {
        if constexpr (B_T) return readT(stage + MID_A_BYTES + wn * WG_SUB, f);
        else return readN(stage + MID_A_BYTES, wn * 96, f);
    };
    auto readA = [&](auto S, const unsigned char* stage) {
#pragma unroll
        for (int f = 0; f < 6; ++f) fa[decltype(S)::value][f] = readN(stage, wm * 96, f);
    };
    __syncthreads();
    if (active) {
        readA(std::integral_constant<int, 0>{}, smem);
#pragma unroll
        for (int f = 0; f < 6; ++f) fb[f] = readB(smem, f);
    }
    __syncthreads();
    // step t: the A fragments of step t + 1 are requested first, then the 36 MFMAs of step t run; each B fragment of step t + 1
    // is read right behind the last MFMA that uses its predecessor (wgrad_tile's schedule: one barrier per step)
    auto step = [&](auto R, int t) {
        constexpr int r = decltype(R)::value;
        const unsigned char* nxt = smem + ((t & 1) ^ 1) * STAGE;
        if (active) {
            const bool more = t + 1 < nt;
            if (more) readA(std::integral_constant<int, (r + 1) & 1>{}, nxt);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[r & 1][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int i = 3; i < 6; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[r & 1][i], acc[i][j], 0, 0, 0);
                if (more) fb[j] = readB(nxt, j);
            }
        }
        __syncthreads();
    };
    for (int t = 0; t < nt; t += 2)
        static_for<2>([&](auto R) { step(R, t + decltype(R)::value); });

    if (!active) return;
    unsigned char* wst = smem + wid * (32 * WG_STG_PITCH);
    // (ps through static_for: a loop the compiler declines to unroll would index the accumulators dynamically, i.e. keep all 36 in
    // scratch; the item loop inside stays rolled -- it only reads the staged rows -- so the epilogue switch exists three times, not 18)
    static_for<3>([&](auto PS) {
        constexpr int ps = decltype(PS)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *(f32x4*)(wst + (ii * 16 + li) * WG_STG_PITCH + (j * 16 + g * 4) * 4) = acc[ps * 2 + ii][j];
        // (a wave reads back only what it wrote: no barrier)
#pragma unroll 1
        for (int it = 0; it < 6; ++it) {
            const int c = lane + it * 64;
            const int rl = c / 12, c8 = c - rl * 12;
            const int m = m0 + wm * 96 + ps * 32 + rl, n = n0 + wn * 96 + c8 * 8;
            if (m < p.M && n < p.N) {
                const float4 lo = *(const float4*)(wst + rl * WG_STG_PITCH + c8 * 32);
                const float4 hi = *(const float4*)(wst + rl * WG_STG_PITCH + c8 * 32 + 16);
                epilogue8(p, m, n, lo, hi, bz);
            }
        }
    });
}
template <bool B_T>
__global__ __launch_bounds__(512) void gemm_mid_kernel(const GemmArgs p, const int gx, const int gy) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[MidGeo<B_T>::LDS];
    // workgroup b runs on XCD b % 8: each XCD gets a contiguous run of tiles in (K split, row block, column block) order, so the
    // tiles that share a block of A rows read it through one L2
    int b = blockIdx.x;
    const int nb8 = (int)gridDim.x & ~7;
    if (b < nb8) b = (b & 7) * (nb8 >> 3) + (b >> 3);
    const int tiles = gx * gy;
    const int bz = b / tiles;
    b -= bz * tiles;
    const int by = b / gx;
    gemm_mid_tile<B_T>(p, b - by * gx, by, bz, smem);
}

// split-K for the ordinary epilogues: the GEMM wrote raw fp32 partial slabs [splits][M][N]; fold them
// and apply the fused epilogue (bias / GELU / residual / ...) once.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs p, const float* __restrict__ slabs,
                                                              int splits) {
    const int n8 = p.N >> 3;
    const int total = p.M * n8;                 // (the launcher only folds outputs of < 2^31 8-column chunks)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int m = fast_div(i, n8), n = (i - m * n8) * 8;
        const float* src = slabs + (size_t)m * p.N + n;
        const float4 lo = fold_slabs4(src, (size_t)p.M * p.N, splits), hi = fold_slabs4(src + 4, (size_t)p.M * p.N, splits);
        epilogue8(p, m, n, lo, hi, 0);
    }
}

template <bool A_T, bool B_T>
int launch(const GemmArgs& p, int splits, hipStream_t stream) {
    if constexpr (!A_T) {
        // the 192 x 192 loader-wave kernel (whole 64-deep stages only): where the CALLER asks for it (TULIP_GEMM_MID) -- it wins on
        // narrow outputs over a deep K with the K split that fills the chip and loses elsewhere (profiles/r5_gemm_big.txt; a rule
        // of the launcher's own by tile count put the 32 768 x 96 x 192 skip Linear of the batch-8 step on it: 11 -> 17.6 us), so
        // the choice is the engine's (TulipEngine._gemm), not a heuristic down here
        const int gx = (p.N + 191) / 192, gy = (p.M + 191) / 192;
        const bool fits = p.kchunk % 64 == 0 && p.K % p.kchunk == 0 && p.N >= 96 && p.M >= 96 &&
                          p.epi != TULIP_EPI_PIXSHUF2_F32 && p.epi != TULIP_EPI_UNSHUF2_BF16;
        if (fits && p.mid == 2) {
            hipLaunchKernelGGL((gemm_mid_kernel<B_T>), dim3(gx * gy * splits), dim3(512), 0, stream, p, gx, gy);
            TULIP_CHECK_LAUNCH();
            return TULIP_OK;
        }
    }
    // BM=64 (with 128-deep k stages: 4 sub-tiles of loads in flight per thread) when BM=128 would leave
    // most of the 256 CUs idle -- the small-M / large-K GEMMs of the deep stages are load-latency bound
    const int gn = (p.N + BN - 1) / BN;
    // 64-row tiles unless the launch already has thousands of 128-row tiles: at B=8 every GEMM of this model
    // is latency-bound per workgroup, and twice as many half-size workgroups in flight measured 4 % faster
    // end to end
    // Tile height by the tiles the launch would have (tools/gemm_big.py, isolated, TFLOP/s at 64 / 128 / 256 rows): 4096x2304x768
    // 355 / 492 / 453, 4096x3072x768 441 / 461 / 529, 2048x6144x1536 519 / 555 / 704, 8192x2304x768 461 / 580 / 536, the N = 768
    // data gradients (K = 2304..6144) 287-312 / 261-297 / same: taller tiles only while at least two rounds of the chip remain,
    // 256 rows only for wide outputs (>= 32 column tiles).  Every GEMM of the batch-8 step stays on 64 rows.
    const int t128 = ((p.M + 127) / 128) * gn * splits, t256 = ((p.M + 255) / 256) * gn * splits;
    int bm = (t128 < TULIP_GEMM_MID_TILES || p.M <= 64) ? 64 : 128;
    if (t256 >= TULIP_GEMM_MID_TILES && gn >= 32) bm = 256;
    if (bm == 64) {
        dim3 grid(gn, (p.M + 63) / 64, splits);
        // 128-deep k stages (80-150 KB LDS, 1-2 workgroups/CU) pay while the grid is at most ~1.5 waves of the chip
        const bool deep = (int)(grid.x * grid.y * grid.z) <= TULIP_GEMM_KSUB_GRID && p.kchunk >= 256;
        const int bks = deep ? 128 : 32;
        const bool full = !p.checked && !A_T && p.M % 64 == 0 && p.N % BN == 0 && p.kchunk % bks == 0 && p.K % p.kchunk == 0;
        if constexpr (!A_T) {
            if (full) {
                if (deep) hipLaunchKernelGGL((gemm_kernel_full<64, B_T, 4>), grid, dim3(256), 0, stream, p);
                else hipLaunchKernelGGL((gemm_kernel_full<64, B_T, 1>), grid, dim3(256), 0, stream, p);
                TULIP_CHECK_LAUNCH();
                return TULIP_OK;
            }
        }
        if (deep)
            hipLaunchKernelGGL((gemm_kernel<64, A_T, B_T, 4>), grid, dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((gemm_kernel<64, A_T, B_T, 1>), grid, dim3(256), 0, stream, p);
    } else {
        const bool full = !p.checked && !A_T && p.M % bm == 0 && p.N % BN == 0 && p.kchunk % 32 == 0 && p.K % p.kchunk == 0;
        dim3 grid(gn, (p.M + bm - 1) / bm, splits);
        if constexpr (!A_T) {
            if (full) {
                if (bm == 256) hipLaunchKernelGGL((gemm_kernel_full<256, B_T, 1>), grid, dim3(256), 0, stream, p);
                else hipLaunchKernelGGL((gemm_kernel_full<128, B_T, 1>), grid, dim3(256), 0, stream, p);
                TULIP_CHECK_LAUNCH();
                return TULIP_OK;
            }
        }
        if (bm == 256) hipLaunchKernelGGL((gemm_kernel<256, A_T, B_T, 1>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_kernel<128, A_T, B_T, 1>), grid, dim3(256), 0, stream, p);
    }
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

}  // namespace

static int effective_splits(int K, int splits) {
    if (splits < 1) splits = 1;
    const int kchunk = (((K + splits - 1) / splits) + BK - 1) / BK * BK;
    return (K + kchunk - 1) / kchunk;
}

extern "C" int tulip_gemm_effective_splits(int K, int splits) { return K > 0 ? effective_splits(K, splits) : 1; }
extern "C" int tulip_gemm_packed_supported(int M, int N, int K, int splits) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 31)) return 0;
    if (splits < 1) splits = 1;
    const int kchunk = (((K + splits - 1) / splits) + BK - 1) / BK * BK;
    return stream_shape_ok(M, N, K, kchunk) ? 1 : 0;
}

extern "C" int tulip_gemm_bf16(const void* A, int lda, int a_trans, const void* B, int ldb, int b_trans, int M, int N,
                               int K, int epi, const float* bias, void* out, int ldo, void* out2, int ldo2,
                               const void* aux, int ldaux, const float* rowscale, int rows_per_sample, int accumulate,
                               int psH, int psW, int splits, void* workspace, int64_t workspace_bytes,
                               hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TULIP_OK;
    if ((K & 7) || (N & 7) || (lda & 7) || (ldb & 7)) return TULIP_ERR_ARG;
    if (a_trans && (M & 7)) return TULIP_ERR_ARG;
    if (b_trans && (N & 7)) return TULIP_ERR_ARG;
    if (splits < 1) splits = 1;
    const bool raw_split = epi == TULIP_EPI_SPLIT_F32;
    if (splits > 1 && !raw_split) {
        splits = effective_splits(K, splits);
        if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * N * 4)) return TULIP_ERR_ARG;
    }
    if ((epi == TULIP_EPI_RESID_F32 || epi == TULIP_EPI_GELU_BWD) && !aux) return TULIP_ERR_ARG;
    if (epi == TULIP_EPI_GELU_DUAL && !out2) return TULIP_ERR_ARG;
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K;
    int kchunk = (((K + splits - 1) / splits) + BK - 1) / BK * BK;
    p.kchunk = kchunk;
    splits = (K + kchunk - 1) / kchunk;
    p.epi = epi; p.bias = bias; p.out = out; p.ldo = ldo; p.out2 = out2; p.ldo2 = ldo2;
    p.aux = aux; p.ldaux = ldaux; p.rowscale = rowscale; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    p.accumulate = accumulate & TULIP_GEMM_ACCUMULATE; p.psH = psH; p.psW = psW;
    p.touch = (accumulate & TULIP_GEMM_NO_TOUCH) ? 0 : 1; p.checked = (accumulate & TULIP_GEMM_CHECKED) ? 1 : 0;
    p.mid = (accumulate & TULIP_GEMM_MID) ? 2 : (accumulate & TULIP_GEMM_NO_MID) ? 0 : 1;
    // (any pitch is accepted for these epilogues: the scattered 2-byte form takes what the vector stores cannot)
    p.vec_ok = epi == TULIP_EPI_PIXSHUF2_F32 ? ((ldo2 & 7) == 0 && ((uintptr_t)out2 & 15) == 0 && ((uintptr_t)out & 15) == 0)
             : epi == TULIP_EPI_UNSHUF2_BF16 ? ((ldo & 1) == 0 && ((uintptr_t)out & 3) == 0) : 0;
    GemmArgs q = p;  // what the GEMM kernel itself does
    const bool fold = splits > 1 && !raw_split;
    if (fold) {
        q.epi = TULIP_EPI_SPLIT_F32; q.bias = nullptr; q.out = workspace; q.ldo = N; q.out2 = nullptr;
    }
    int rc;
    if (accumulate & TULIP_GEMM_B_PACKED) {           // B: the fragment-major copy of the [N][K] matrix (gemm_stream_kernel)
        if (a_trans || b_trans || !stream_shape_ok(M, N, K, kchunk)) return TULIP_ERR_ARG;
        const dim3 grid(N / BN, M / 32, splits);
        if (kchunk == 1536) hipLaunchKernelGGL((gemm_stream_kernel<48>), grid, dim3(256), 0, stream, q);
        else if (kchunk == 768) hipLaunchKernelGGL((gemm_stream_kernel<24>), grid, dim3(256), 0, stream, q);
        else if (kchunk == 384) hipLaunchKernelGGL((gemm_stream_kernel<12>), grid, dim3(256), 0, stream, q);
        else hipLaunchKernelGGL((gemm_stream_kernel<3>), grid, dim3(256), 0, stream, q);
        hipError_t e__ = hipGetLastError();
        rc = e__ != hipSuccess ? -(1000 + (int)e__) : TULIP_OK;
    } else
    if (!a_trans && !b_trans) rc = launch<false, false>(q, splits, stream);
    else if (!a_trans && b_trans) rc = launch<false, true>(q, splits, stream);
    else if (a_trans && b_trans) rc = launch<true, true>(q, splits, stream);
    else rc = launch<true, false>(q, splits, stream);
    if (rc != TULIP_OK || !fold) return rc;
    const int64_t work = (int64_t)M * (N >> 3);
    if (work >= (int64_t)1 << 31) return TULIP_ERR_ARG;
    const int grid = (int)std::min<int64_t>((work + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, stream, p, (const float*)workspace, splits);
    TULIP_CHECK_LAUNCH();
    return TULIP_OK;
}

extern "C" int tulip_reduce_rows_multi_adamw(const tulip_reduce_region* regions, int n, const tulip_adamw_ref* adam,
                                             hipStream_t stream);

// tile shape of the large-tile weight-gradient kernel for a [Nw][Kw] gradient: 0 = 192 x 192, 1 = 384 x 96, 2 = 96 x 384,
// -1 = none (the 64 x 96 tile of gemm_group_kernel)
static int wgrad_shape(int Nw, int Kw, int flags) {
    if (flags & TULIP_WGRAD_SMALL_TILES) return -1;
    if (Nw % 192 == 0 && Kw % 192 == 0) return 0;
    if (Kw == 96 && Nw % 96 == 0) return 1;
    if (Nw == 96 && Kw % 96 == 0) return 2;
    return -1;
}
static void wgrad_tile_grid(int shape, int Nw, int Kw, int* gx, int* gy) {
    const int tm = shape == 0 ? 192 : shape == 1 ? 384 : shape == 2 ? 96 : 64;
    const int tn = shape == 0 ? 192 : shape == 1 ? 96 : shape == 2 ? 384 : BN;
    *gx = (Kw + tn - 1) / tn;
    *gy = (Nw + tm - 1) / tm;
}
extern "C" int tulip_wgrad_tiles(int Nw, int Kw, int flags) {
    int gx, gy;
    wgrad_tile_grid(wgrad_shape(Nw, Kw, flags), Nw, Kw, &gx, &gy);
    return gx * gy;
}

static int wgrad_group_impl(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                            void* workspace, int64_t workspace_bytes, int fold, const tulip_adamw_ref* adam, void* prof,
                            hipStream_t stream);

extern "C" int tulip_wgrad_group(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                                 void* workspace, int64_t workspace_bytes, int fold, hipStream_t stream) {
    return wgrad_group_impl(items, n, extra, n_extra, workspace, workspace_bytes, fold, nullptr, nullptr, stream);
}

extern "C" int tulip_wgrad_group_adamw(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                                       void* workspace, int64_t workspace_bytes, int fold, const tulip_adamw_ref* adam,
                                       hipStream_t stream) {
    if (adam && (!adam->hyper || !adam->grad || !adam->param || !adam->exp_avg || !adam->exp_avg_sq || !adam->param_bf16))
        return TULIP_ERR_ARG;
    return wgrad_group_impl(items, n, extra, n_extra, workspace, workspace_bytes, fold, adam, nullptr, stream);
}

// dev (tools/wgrad_phases.py): the grouped launch alone (no fold, no optimizer step) with 4 shader-clock stamps per workgroup
extern "C" int tulip_wgrad_group_profiled(const tulip_wgrad_item* items, int n, void* workspace, int64_t workspace_bytes, int flags,
                                          void* stamps, hipStream_t stream) {
#if !TULIP_DEV_VARIANTS
    return TULIP_ERR_NOT_BUILT;
#endif
    return wgrad_group_impl(items, n, nullptr, 0, workspace, workspace_bytes, flags & ~TULIP_WGRAD_FOLD, nullptr, stamps, stream);
}

static int wgrad_group_impl(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                            void* workspace, int64_t workspace_bytes, int fold, const tulip_adamw_ref* adam, void* prof,
                            hipStream_t stream) {
    if (n < 0 || n > GROUP_MAX || n_extra < 0 || n + n + n_extra > TULIP_REDUCE_REGIONS_MAX || (n && !items) ||
        (n_extra && !extra))
        return TULIP_ERR_ARG;
    bool big = true;
    for (int i = 0; i < n; ++i)
        if (items[i].Nw > 0 && items[i].Kw > 0 && items[i].Mtok > 0 &&
            (wgrad_shape(items[i].Nw, items[i].Kw, fold) < 0 || (items[i].Mtok & 31)))
            big = false;
    WgradGroup G;
    tulip_reduce_region folds[TULIP_REDUCE_REGIONS_MAX];
    int nf = 0;
    int64_t ws_used = 0;                       // floats
    float* ws = (float*)workspace;
    G.n = 0;
    G.first[0] = 0;
    bool deep = true;
    for (int i = 0; i < n; ++i) {
        const tulip_wgrad_item& it = items[i];
        if (it.Nw <= 0 || it.Kw <= 0 || it.Mtok <= 0) continue;
        if ((it.Mtok & 7) || (it.Nw & 7) || (it.Kw & 7) || (it.ldy & 7) || (it.ldx & 7) || !it.dY || !it.X || !it.dW)
            return TULIP_ERR_ARG;
        int splits = it.splits < 1 ? 1 : it.splits;
        const int kchunk = (((it.Mtok + splits - 1) / splits) + BK - 1) / BK * BK;
        splits = (it.Mtok + kchunk - 1) / kchunk;
        GemmArgs& p = G.g[G.n];
        p.A = (const bf16_t*)it.dY; p.B = (const bf16_t*)it.X; p.lda = it.ldy; p.ldb = it.ldx;
        p.M = it.Nw; p.N = it.Kw; p.K = it.Mtok; p.kchunk = kchunk;
        p.bias = nullptr; p.ldo = it.Kw; p.ldo2 = 0; p.aux = nullptr; p.ldaux = 0; p.rowscale = nullptr;
        p.rows_per_sample = 1; p.psH = 0; p.psW = 0; p.touch = 0; p.checked = 0; p.mid = 0; p.vec_ok = 0;
        if (splits > 1) {
            const int64_t nw = (int64_t)it.Nw * it.Kw, need = (nw + (it.db ? it.Nw : 0)) * splits;
            if (!ws || (ws_used + need) * 4 > workspace_bytes) return TULIP_ERR_ARG;
            p.epi = TULIP_EPI_SPLIT_F32; p.accumulate = 0;
            p.out = ws + ws_used;
            p.out2 = it.db ? (void*)(ws + ws_used + nw * splits) : nullptr;
            // reserved_ = 1 on a token-split item: the fold of its slabs takes the optimizer step (weight and bias)
            const int st = (it.reserved_ == 1 && adam && it.overwrite) ? 1 : 0;
            folds[nf++] = tulip_reduce_region{ws + ws_used, it.dW, nw, nw, splits, it.overwrite, nullptr, 0, 0, st};
            if (it.db) folds[nf++] = tulip_reduce_region{ws + ws_used + nw * splits, it.db, it.Nw, it.Nw, splits, it.overwrite, nullptr, 0, 0, st};
            ws_used += need;
        } else {
            p.epi = TULIP_EPI_F32; p.accumulate = it.overwrite ? 0 : 1; p.out = it.dW; p.out2 = it.db;
            // reserved_ = 1: apply AdamW to this (un-split, large-tile, written-not-accumulated) weight gradient in the write-out
            if (it.reserved_ == 1 && adam && big && it.overwrite) p.accumulate |= 2;
        }
        G.shape[G.n] = big ? wgrad_shape(it.Nw, it.Kw, fold) : -1;
        wgrad_tile_grid(G.shape[G.n], it.Nw, it.Kw, &G.gx[G.n], &G.gy[G.n]);
        G.first[G.n + 1] = G.first[G.n] + G.gx[G.n] * G.gy[G.n] * splits;
        deep = deep && kchunk >= 256;
        ++G.n;
    }
    if (G.n > 0) {
        const int blocks = G.first[G.n];
        if (big) {
            G.adam = adam ? AdamRef{adam->hyper, adam->grad, adam->param, adam->exp_avg, adam->exp_avg_sq, (bf16_t*)adam->param_bf16,
                                    adam->decay_mask64}
                          : AdamRef{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            for (int i = 0; i < G.n; ++i) G.g[i].aux = prof;
            hipLaunchKernelGGL(wgrad_group_kernel, dim3(blocks), dim3(512), 0, stream, G);
        } else {
            GemmGroup S;
            S.n = G.n;
            for (int i = 0; i < G.n; ++i) { S.g[i] = G.g[i]; S.gx[i] = G.gx[i]; S.gy[i] = G.gy[i]; }
            for (int i = 0; i <= G.n; ++i) S.first[i] = G.first[i];
            if (blocks <= TULIP_GEMM_KSUB_GRID && deep)
                hipLaunchKernelGGL((gemm_group_kernel<64, true, true, 4>), dim3(blocks), dim3(256), 0, stream, S);
            else
                hipLaunchKernelGGL((gemm_group_kernel<64, true, true, 1>), dim3(blocks), dim3(256), 0, stream, S);
        }
        TULIP_CHECK_LAUNCH();
    }
    if (!(fold & TULIP_WGRAD_FOLD)) return TULIP_OK;
    for (int i = 0; i < n_extra; ++i) folds[nf++] = extra[i];
    return nf ? tulip_reduce_rows_multi_adamw(folds, nf, adam, stream) : TULIP_OK;
}

// The fold regions tulip_wgrad_group(..., fold = 1) would pass to tulip_reduce_rows_multi for these items and this workspace
// (pure host code, same slab layout): for a caller that launches with fold = 0 and folds later in a launch of its own,
// together with regions that only become ready in between.  Returns the number of regions written (<= max), or < 0.
extern "C" int tulip_wgrad_group_regions(const tulip_wgrad_item* items, int n, void* workspace, tulip_reduce_region* out, int max) {
    if (n < 0 || n > GROUP_MAX || (n && !items) || !out) return TULIP_ERR_ARG;
    int nf = 0;
    int64_t ws_used = 0;
    float* ws = (float*)workspace;
    for (int i = 0; i < n; ++i) {
        const tulip_wgrad_item& it = items[i];
        if (it.Nw <= 0 || it.Kw <= 0 || it.Mtok <= 0) continue;
        int splits = it.splits < 1 ? 1 : it.splits;
        const int kchunk = (((it.Mtok + splits - 1) / splits) + BK - 1) / BK * BK;
        splits = (it.Mtok + kchunk - 1) / kchunk;
        if (splits <= 1) continue;
        const int64_t nw = (int64_t)it.Nw * it.Kw, need = (nw + (it.db ? it.Nw : 0)) * splits;
        if (nf + 2 > max) return TULIP_ERR_ARG;
        const int st = (it.reserved_ == 1 && it.overwrite) ? 1 : 0;     // (folded with tulip_reduce_rows_multi_adamw)
        out[nf++] = tulip_reduce_region{ws + ws_used, it.dW, nw, nw, splits, it.overwrite, nullptr, 0, 0, st};
        if (it.db) out[nf++] = tulip_reduce_region{ws + ws_used + nw * splits, it.db, it.Nw, it.Nw, splits, it.overwrite, nullptr, 0, 0, st};
        ws_used += need;
    }
    return nf;
}


