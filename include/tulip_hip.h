/* tulip_hip.h -- C ABI of libtulip_hip.so: the MI355X (gfx950) kernels behind the TULIP Swin hot path.
 *
 * The reference (ethz-asl/TULIP) is pure Python and defines no FFI; the boundary it exposes for
 * this path is the nn.Module in tulip/model/tulip.py.  Each entry point below replaces the ATen
 * op sequence of the cited reference lines (paths relative to the reference repo).  The Python
 * host (tulip_amd/model/tulip.py) binds them with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes; every buffer is owned by the caller and lives in device memory.
 *   - bf16 tensors are uint16_t (raw bits), row-major, innermost dimension contiguous.
 *   - every call is asynchronous on `stream`, allocates nothing, keeps no global state (the library has no mutable
 *     globals, no setters and reads no environment variable: measurement switches are per-call flag bits), performs
 *     no synchronisation, is re-entrant and is legal under HIP-graph capture.
 *   - return value: 0 on success, TULIP_ERR_ARG (-1) for an unsupported argument combination, TULIP_ERR_NOT_BUILT (-3)
 *     for a form that exists only in the development build (below), -(1000 + hipError_t) if the launch failed.
 *     Nothing throws, nothing exits.
 *   - TWO builds of one symbol set: libtulip_hip.so carries what a training / inference step launches; libtulip_hip_dev.so
 *     (-DTULIP_DEV_VARIANTS=1; tulip_dev_variants() == 1) additionally the forms that were built, bit-tested and measured but
 *     are not on the path -- the *_profiled twins (in-kernel clock stamps), the recomputing C = 96 backward (qkv = fc1_pre = NULL
 *     with the other saved tensors given), tulip_swinw_block_bwd_split, and tulip_swinw_block_fwd / _bwd in their training
 *     forms WITHOUT TULIP_BLOCK_FC1_GRAD.  The product library answers those calls with TULIP_ERR_NOT_BUILT.
 *   - "stream" tensors (the residual stream) are fp32 (B,H,W,C); GEMM operands are bf16 with
 *     fp32 accumulation (the autocast contract of engine_upsampling.py:77, bf16 instead of fp16).
 */
#ifndef TULIP_HIP_H_
#define TULIP_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

/* GEMM epilogues */
#define TULIP_EPI_BF16 0         /* out_bf16 = acc + bias                                             */
#define TULIP_EPI_GELU_DUAL 1    /* out = bf16(acc+bias) ; out2 = bf16(gelu_erf(out))  (tulip.py:195-196) */
#define TULIP_EPI_GELU_BWD 2     /* out_bf16 = acc * gelu'(aux_bf16)                                    */
#define TULIP_EPI_F32 3          /* out_f32 (+)= acc + bias                                             */
#define TULIP_EPI_RESID_F32 4    /* out_f32 = aux_f32 + rowscale[m/rows_per_sample]*(acc+bias) (tulip.py:343-344,350-351) */
#define TULIP_EPI_PIXSHUF2_F32 5 /* PatchUnmerging scatter: PixelShuffle(2) + BCHW->BHWC (tulip.py:120-122) */
/* (6: retired) */
#define TULIP_EPI_SPLIT_F32 7    /* out_f32[split][M][ldo] = acc : split-K partial slabs (deterministic) */
#define TULIP_EPI_UNSHUF2_BF16 8 /* inverse of 5, bf16: row m = fine token (b,2h+i,2w+j), column c -> out[(b,h,w)][4c+2i+j]
                                    with psH, psW = the COARSE grid, N = fine channels (backward of PixelShuffle(2)) */
/* TULIP_EPI_F32 / TULIP_EPI_RESID_F32 with out2 != NULL and ldo2 > 0 additionally store bf16(result * rowscale)
 * at out2[m*ldo2 + n] (the operand of the next GEMM on the path); TULIP_EPI_PIXSHUF2_F32 stores fp32 to `out`
 * and/or bf16 to `out2` (row pitch ldo2), whichever is non-NULL. */

/* C[M,N] = opA[M,K] . opB[N,K]^T, bf16 in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.
 * a_trans=0: A is [M][lda]; a_trans=1: A is [K][lda] (A^T stored).  Same for B ([N][ldb] / [K][ldb]).
 * Replaces nn.Linear / 1x1 nn.Conv2d forward (tulip.py:298,318,195,198,105,119,716,175) and their
 * autograd dgrad/wgrad.  Requirements: K%8==0, N%8==0, lda%8==0, ldb%8==0 (and M%8 / N%8 for the
 * transposed operands).  splits>1: the K range is cut across workgroups (for launches too small to fill 256
 * CUs).  With TULIP_EPI_SPLIT_F32 the raw partials go to `out`; with any other epilogue
 * the partial slabs go to `workspace` (>= effective_splits*M*N*4 bytes) and a second kernel folds them and
 * applies the epilogue.  workspace may be NULL when splits == 1.
 * Weight-gradient form (a_trans=1, epi SPLIT_F32 or F32): if out2 != NULL it additionally receives
 * the row sums of opA, i.e. sum over tokens of dY = the bias gradient, as fp32 [splits][M] (SPLIT) or
 * [M] (F32, += when accumulate) -- computed by one extra MFMA per fragment against an all-ones operand.
 * `accumulate` is a flag word: TULIP_GEMM_ACCUMULATE (bit 0, the 0 / 1 of earlier versions); TULIP_GEMM_NO_TOUCH: skip the
 * split first touch of the cold [96][k range] weight panel that the M-tile workgroups of an N panel perform before their k
 * loops (forward / data-gradient form); TULIP_GEMM_CHECKED: the bounds-checked kernels even where whole tiles allow the
 * unchecked ones.  The last two are measurement / bit-compare switches: results are identical either way. */
#define TULIP_GEMM_ACCUMULATE 1
#define TULIP_GEMM_NO_TOUCH 0x100
#define TULIP_GEMM_CHECKED 0x200
/* TULIP_GEMM_MID: run the 192 x 192 loader-wave kernel (csrc/gemm.hip, gemm_mid_tile) wherever it fits (a_trans = 0, K range per
 * split a multiple of 64, M, N >= 96, not the PixelShuffle epilogues) -- the caller's choice, for narrow outputs over a deep K at
 * M >= 4096 with the K split that fills the chip; bit-identical results (both kernels sum k in the same order per output element).
 * TULIP_GEMM_NO_MID is accepted and means the default (never, unless asked). */
#define TULIP_GEMM_NO_MID 0x400
#define TULIP_GEMM_MID 0x800
/* TULIP_GEMM_B_PACKED: B is the FRAGMENT-MAJOR copy (tulip_pack_bf16_multi) of the [N][K] matrix -- for a data gradient, of the
 * transposed weight -- and the launch is the small-K form (csrc/gemm.hip, gemm_stream_kernel): 32 x 96 output tiles, the whole K
 * range of a split in flight at once.  a_trans = b_trans = 0, ldb ignored; tulip_gemm_packed_supported(M, N, K, splits) says where
 * it exists (M % 32 == 0, N % 96 == 0, K per split in {96, 384, 768, 1536}), TULIP_ERR_ARG elsewhere.  Same epilogues, same bits as the
 * plain call on the row-major matrix. */
#define TULIP_GEMM_B_PACKED 0x1000
int tulip_gemm_bf16(const void* A, int lda, int a_trans, const void* B, int ldb, int b_trans, int M, int N, int K,
                    int epi, const float* bias, void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux,
                    const float* rowscale, int rows_per_sample, int accumulate, int psH, int psW, int splits,
                    void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* Up to TULIP_REDUCE_REGIONS_MAX row reductions in one launch:  out[i] (+)= sum_{s<rows} partials[s*stride + i],
 * i < n (n, stride multiples of 4).  With scatter_index != NULL the region is the dense [scatter_nh][scatter_len]
 * relative-position-bias gradient and its sums are added to out[scatter_index[ij]*scatter_nh + h] instead
 * (tulip.py:304-308 backwards; one launch, deterministic: no atomics). */
#define TULIP_REDUCE_REGIONS_MAX 48
typedef struct tulip_reduce_region {
    const float* partials; float* out;
    int64_t stride; int64_t n;
    int rows; int overwrite;
    const int32_t* scatter_index; int scatter_nh; int scatter_len;
    int adamw;      /* 1: tulip_reduce_rows_multi_adamw takes the optimizer step of out[0..n) instead of storing the sum */
} tulip_reduce_region;
int tulip_reduce_rows_multi(const tulip_reduce_region* regions, int n, hipStream_t stream);

/* The weight (and bias) gradients of up to TULIP_WGRAD_GROUP_MAX Linear layers in TWO launches (autograd of
 * nn.Linear, tulip.py:298,318,195,198):  dW[Nw][Kw] += dY[Mtok][Nw]^T . X[Mtok][Kw],  db[Nw] += sum_tokens dY.
 * One grouped GEMM launch covers all items (token dimension cut `splits` ways into fp32 slabs in `workspace`, or
 * accumulated in place when splits == 1), one tulip_reduce_rows_multi launch folds the slabs -- and the `extra`
 * regions (LayerNorm / bias-table partial rows of the same block) ride along in that launch.  `fold` is a flag word:
 * without TULIP_WGRAD_FOLD (bit 0, the 0 / 1 of earlier versions) the second launch is skipped and the slabs stay in the
 * workspace; TULIP_WGRAD_SMALL_TILES forces the 64 x 96 tile everywhere (A/B measurements; tulip_wgrad_tiles takes the
 * same word).  A launch is sized by its caller to
 * about one workgroup per CU in total (`splits`), so the slab traffic is per LAUNCH, not per linear: the engine puts a
 * whole stage (two Swin blocks + boundary linears, up to 12 items) into one. */
#define TULIP_WGRAD_GROUP_MAX 16
#define TULIP_WGRAD_FOLD 1
#define TULIP_WGRAD_SMALL_TILES 0x100
typedef struct tulip_wgrad_item {
    const void* dY; const void* X; float* dW; float* db;
    int ldy; int ldx; int Nw; int Kw; int Mtok; int splits;
    int overwrite;      /* 0: dW += ..., db += ... (autograd's accumulation); 1: dW = ..., db = ... (the caller guarantees this is the
                           only contribution of the step: the gradient buffer then needs no clearing) */
    int reserved_;      /* tulip_wgrad_group_adamw: 1 = apply the optimizer step in the write-out (see there); else 0 */
} tulip_wgrad_item;
int tulip_wgrad_group(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                      void* workspace, int64_t workspace_bytes, int fold, hipStream_t stream);
/* tulip_wgrad_group that also takes the optimizer step of the items marked reserved_ = 1 (and overwrite = 1, large tiles, no
 * token split: the workgroup holds the complete gradient tile) in its write-out -- torch.optim.AdamW.step (main_lidar_upsampling.py:283)
 * on fp32 master weights + bf16 shadow, same arithmetic as tulip_adamw, decoupled decay on -- instead of storing the gradient:
 * param / exp_avg / exp_avg_sq / param_bf16 are flat buffers laid out like `grad` (an element is addressed by its gradient's
 * offset from `grad`), hyper as in tulip_adamw.  The caller's end-of-step tulip_adamw must skip those tensors (mask bit 1).
 * adam == NULL: exactly tulip_wgrad_group. */
typedef struct tulip_adamw_ref {
    const float* hyper; const float* grad; float* param; float* exp_avg; float* exp_avg_sq; uint16_t* param_bf16;
    const uint8_t* decay_mask64;    /* tulip_adamw's mask (bit 0 = weight decay applies to these 64 elements); NULL: everywhere */
} tulip_adamw_ref;
int tulip_wgrad_group_adamw(const tulip_wgrad_item* items, int n, const tulip_reduce_region* extra, int n_extra,
                            void* workspace, int64_t workspace_bytes, int fold, const tulip_adamw_ref* adam, hipStream_t stream);
/* tulip_reduce_rows_multi whose regions marked adamw = 1 (overwrite = 1, no scatter: the sum is the complete gradient of
 * out[0..n), a range of `adam->grad`) take the optimizer step in place of storing the sum -- same arithmetic as tulip_adamw,
 * decay per 64 elements from adam->decay_mask64; the caller's end-of-step tulip_adamw must skip those ranges (mask bit 1).
 * tulip_wgrad_group_adamw marks the fold regions of its token-split items with reserved_ = 1 this way (weight and bias).
 * adam == NULL: exactly tulip_reduce_rows_multi (a marked region is then an argument error). */
int tulip_reduce_rows_multi_adamw(const tulip_reduce_region* regions, int n, const tulip_adamw_ref* adam, hipStream_t stream);
/* The fold regions tulip_wgrad_group(fold = 1) would hand to tulip_reduce_rows_multi for these items and this workspace (host
 * code only): for a caller that launches with fold = 0 and folds later, in one launch with regions that become ready in
 * between (the engine: the patch-embedding partial rows ride in the fold of the backward's last weight-gradient group).
 * Returns the number of regions written to `out` (<= max), or a negative error. */
int tulip_wgrad_group_regions(const tulip_wgrad_item* items, int n, void* workspace, tulip_reduce_region* out, int max);

/* Workgroup tiles per token split the grouped launch uses for a [Nw][Kw] weight gradient (192 x 192 per workgroup where
 * both dimensions are multiples of 192, 384 x 96 / 96 x 384 for the 96-wide stage, else 64 x 96): what a caller sizes
 * `splits` with.  flags: the `fold` word of the launch it is sized for (TULIP_WGRAD_SMALL_TILES matters). */
int tulip_wgrad_tiles(int Nw, int Kw, int flags);
/* Profiling twin (tools/wgrad_phases.py): the grouped launch alone -- no fold, no optimizer step -- writing 4 x uint64 per
 * workgroup of a large-tile launch into `stamps` (shader clock: start, end of the pipeline prologue, end of the k loop, end
 * of the write-out). */
int tulip_wgrad_group_profiled(const tulip_wgrad_item* items, int n, void* workspace, int64_t workspace_bytes, int flags,
                               void* stamps, hipStream_t stream);

/* number of K-splits tulip_gemm_bf16 actually launches for (K, splits): K is cut in multiples of 32 */
int tulip_gemm_effective_splits(int K, int splits);
int tulip_gemm_packed_supported(int M, int N, int K, int splits);

/* REDUCTIONS.  No kernel in this library funnels many workgroups into same-address atomics (on gfx950 a
 * chain of same-address device-scope fp32 atomics costs ~0.1-0.5 us per link).  Every cross-workgroup sum
 * (split-K weight gradients, bias / LayerNorm / relative-position-bias / patch-embed / decoder_pred
 * gradients) is written as per-workgroup PARTIAL ROWS with plain stores and folded, deterministically, by:
 *   out_r[i] += sum_{s<nrows} part_r[s*stride_r + i],  i < n_r,   for two regions r = 0,1 in one launch.
 * n, stride multiples of 4; a region with n<=0 is skipped. */
int tulip_reduce_rows2(const float* part0, int64_t stride0, float* out0, int64_t n0, const float* part1,
                       int64_t stride1, float* out1, int64_t n1, int nrows, hipStream_t stream);
/* out[i] += sum_s slabs[s*n + i]  (= tulip_reduce_rows2 with one region of stride n) */
int tulip_reduce_splits(const float* slabs, float* out, int64_t n, int splits, hipStream_t stream);

/* LayerNorm over the last dim of an fp32 stream tensor -> bf16 (the next op is always a GEMM).
 * merge=0: x is [rows][C].  merge=1 (PatchMerging, tulip.py:92-105): x is (B,H,W,C/4) and row
 * (b,h',w') is the concat [x(2h',2w'), x(2h'+1,2w'), x(2h',2w'+1), x(2h'+1,2w'+1)]; rows=B*H/2*W/2.
 * Saves mean/rstd per row for the backward.  Replaces tulip.py:340,347,104,720. */
int tulip_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean, float* rstd,
                        int rows, int C, float eps, int merge, int B, int H, int W, hipStream_t stream);

/* dx = dres + LayerNorm-backward(dy) in the layout of x (scatter for merge=1).  dres may be NULL
 * (treated as zero) or alias dx.  param_partials (may be NULL): receives
 * tulip_layernorm_bwd_partial_rows(rows, C) partial rows of [dgamma[C] | dbeta[C]] (stride 2C), to be
 * folded with tulip_reduce_rows2 -- the parameter gradients cost no second pass over dy and x.
 * dx_bf16 (may be NULL): additionally receives bf16(dx * cast_rowscale[token / cast_rows_per_sample]) in the
 * layout of dx -- the operand of the next dgrad/wgrad GEMM on the backward chain (DropPath scale of the branch
 * it enters, tulip.py:25-29), so no separate cast pass is needed. */
int tulip_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, int rows, int C, int merge, int B, int H, int W,
                        float* param_partials, uint16_t* dx_bf16, const float* cast_rowscale,
                        int cast_rows_per_sample, hipStream_t stream);
/* tulip_layernorm_bwd whose incoming gradient is still the `nslab` raw split-K slabs [nslab][rows][C] of the data-gradient
 * GEMM in front (tulip_gemm_bf16 with TULIP_EPI_SPLIT_F32): folded in slab order and rounded to bf16 exactly as that
 * GEMM's own fold launch would have stored them -- one launch less on the backward chain. */
int tulip_layernorm_bwd_splitk(const float* slabs, int nslab, const float* x, const float* mean, const float* rstd,
                               const float* gamma, const float* dres, float* dx, int rows, int C, int merge, int B, int H,
                               int W, float* param_partials, uint16_t* dx_bf16, const float* cast_rowscale,
                               int cast_rows_per_sample, hipStream_t stream);
/* Split-K fold + residual epilogue + the LayerNorm that follows (proj -> norm2, fc2 -> next block's norm1, tulip.py:344-347)
 * in one launch, for a GEMM that left raw slabs [nslab][M][N]: out = aux + rowscale[row / rows_per_sample] * (sum slabs +
 * bias) (aux NULL: no residual), optional bf16 copy, then ln_out = bf16(LayerNorm(out row)), mean, rstd.  N % 256 == 0,
 * N <= 2048 (tulip_splitk_resid_ln_supported). */
int tulip_splitk_resid_ln_supported(int N);
int tulip_splitk_resid_ln(const float* slabs, int nslab, int M, int N, const float* bias, const float* aux, int ldaux,
                          const float* rowscale, int rows_per_sample, float* out, int ldo, uint16_t* out_bf16, int ldo2,
                          const float* gamma, const float* beta, uint16_t* ln_out, float* mean, float* rstd, float eps,
                          hipStream_t stream);
/* partial rows tulip_layernorm_bwd writes for (rows, C); 0 if C is too wide for the fused form (C > 2048) */
int tulip_layernorm_bwd_partial_rows(int rows, int C);

/* stand-alone dgamma[c] += sum_rows dy*xhat ; dbeta[c] += sum_rows dy (atomic; only for C > 2048) */
int tulip_layernorm_bwd_params(const uint16_t* dy, const float* x, const float* mean, const float* rstd, float* dgamma,
                               float* dbeta, int rows, int C, int merge, int B, int H, int W, hipStream_t stream);

/* PatchEmbedding (tulip.py:63-73): optional circular W padding by (2,2), Conv2d(Cin->E,(p0,kw),
 * stride (p0,p1)), BCHW->BHWC, LayerNorm(E).  img (B,Cin,Hin,Win) fp32 -> out (B,Hin/p0,Win/p1,E).
 * out_bf16 (optional): the same rows in bf16 with row pitch ld_bf16 (x_save half of the first skip concat, tulip.py:715). */
int tulip_patch_embed_fwd(const float* img, const float* w, const float* b, const float* gamma, const float* beta,
                          float* out, int B, int Cin, int Hin, int Win, int E, int p0, int p1, int kw, int circular,
                          float eps, uint16_t* out_bf16, int ld_bf16, hipStream_t stream);
/* The same launch also drawing the step's DropPath multipliers (tulip_drop_path_scales below: same generator, same counter
 * update) in its first workgroup -- nothing in this launch reads them, every later kernel of the step is ordered behind it -- so a
 * training step does not begin with a launch of its own for a few hundred numbers.  draw == NULL: exactly tulip_patch_embed_fwd.
 * Shapes the row kernel does not cover issue the draw as its own launch first. */
typedef struct tulip_drop_draw {
    const float* keep; float* scale; float* u_out; int nslots; int B; uint64_t seed; uint64_t* counter;
} tulip_drop_draw;
int tulip_patch_embed_fwd_draw(const float* img, const float* w, const float* b, const float* gamma, const float* beta,
                               float* out, int B, int Cin, int Hin, int Win, int E, int p0, int p1, int kw, int circular,
                               float eps, uint16_t* out_bf16, int ld_bf16, const tulip_drop_draw* draw, hipStream_t stream);
/* parameter gradients of the above (the input image needs no gradient).  partial_stride > 0: dw/db/dgamma/
 * dbeta point into row 0 of a [tulip_patch_embed_bwd_blocks(ntok)][partial_stride] partial buffer (plain
 * stores; fold with tulip_reduce_rows2); partial_stride == 0: accumulate atomically into the gradients. */
int tulip_patch_embed_bwd(const float* img, const float* w, const float* b, const float* gamma, const float* dout,
                          float* dw, float* db, float* dgamma, float* dbeta, int B, int Cin, int Hin, int Win, int E,
                          int p0, int p1, int kw, int circular, float eps, int partial_stride, hipStream_t stream);
int tulip_patch_embed_bwd_blocks(int ntok);

/* Shifted-window attention core (tulip.py:289-323 minus the two Linears): cyclic shift, window
 * partition, q*scale, QK^T, + relative-position bias gathered through rel_index, + shift mask
 * (0/-100 from region labels, tulip.py:254-280), softmax, PV, window reverse, reverse shift -- all
 * as address arithmetic on natural-order tokens.  qkv is [B*H*W][3C] with channel order (T,Nh,P)
 * (tulip.py:298); out is [B*H*W][C] with channel order (Nh,P).  L = wh*ww must be 16; head dim P
 * in {16,32}.  rel_index is the module's (L,L) relative_position_index buffer as int32.
 * `masked` (here and in the block descriptors below) is a bit set: bit 0 (TULIP_ATTN_MASKED) = shifted block, apply the
 * region mask; bit 1 (TULIP_ATTN_FP8) = BASELINE configs[4] "fp8 MFMA attention": the scores Q.K^T are computed by
 * v_mfma_f32_16x16x32_fp8_fp8 from q, k rounded bf16 -> OCP e4m3 (round to nearest even); softmax, P.V and all
 * stored tensors stay as they are, and the backward multiplies dS with the same rounded q, k (it differentiates the
 * function the forward ran).  Default 0/1: the reference's bf16 / fp16 scores. */
#define TULIP_ATTN_MASKED 1
#define TULIP_ATTN_FP8 2
/* bit 2, the fused block kernels (tulip_swin96_block_fwd / _bwd, tulip_swinw_block_fwd / _bwd) only: the fc1_pre buffer carries bf16(gelu'(h)) from the forward to the backward instead
 * of the fc1 pre-activation h itself -- the derivative is all the backward wants from h (autograd of tulip.py:196), and the
 * forward has erf and the Gaussian at hand.  Forward and backward of a block must agree on it. */
#define TULIP_BLOCK_FC1_GRAD 4
/* bit 3, tulip_swinw_block_fwd / _bwd (+ split forms) only: skip the L2 warm-up at the head of the launch (below; measurement
 * and the bit-compare test only, results are identical either way) */
#define TULIP_BLOCK_NO_WARM 8
int tulip_window_attn_fwd(const uint16_t* qkv, const float* bias_table, const int32_t* rel_index, uint16_t* out, int B,
                          int H, int W, int C, int nh, int wh, int ww, int sh, int sw, int masked, hipStream_t stream);
/* dqkv from dout.  d(bias) leaves as R = tulip_window_attn_bwd_partial_rows(...) partial rows per head:
 * dbias_partials[(j*nh + h)*256 + i*16 + k], j < R  ==  a [R][nh*256] matrix whose column sums are the dense
 * [nh][16][16] gradient (fold + scatter into the table: tulip_reduce_rows_multi with scatter_index). */
int tulip_window_attn_bwd(const uint16_t* qkv, const uint16_t* dout, const float* bias_table, const int32_t* rel_index,
                          uint16_t* dqkv, float* dbias_partials, int B, int H, int W, int C, int nh, int wh, int ww,
                          int sh, int sw, int masked, hipStream_t stream);
int tulip_window_attn_bwd_partial_rows(int B, int H, int W, int nh, int wh, int ww);

/* y_bf16 = bf16(x * rowscale[row/rows_per_sample]) ; rowscale may be NULL. x is [rows][cols]. */
int tulip_cast_f32_bf16(const float* x, uint16_t* y, int rows, int cols, const float* rowscale, int rows_per_sample,
                        hipStream_t stream);
/* flat fp32 -> bf16 copy (weight shadow refresh) */
int tulip_cast_flat(const float* x, uint16_t* y, int64_t n, hipStream_t stream);
/* flat bf16 -> fp32 copy (gradients all-reduced in bf16 come back into the fp32 gradient buffer of the optimizer) */
int tulip_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, hipStream_t stream);

/* Fused head (tulip.py:724-731): conv1x1 E->16E (+bias), LeakyReLU(0.01), PixelShuffle(4),
 * conv1x1 E->1 (no bias); xn is norm_up's bf16 output [B*H*W][E]; pred is (B,1,4H,4W) fp32.  The
 * (B,16E,H,W) intermediate (100 MB at B=8) is never materialised.  upscale factor 4, in_chans 1. */
int tulip_tail_fwd(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, float* pred, int B, int H,
                   int W, int E, hipStream_t stream);
/* backward of the head w.r.t. the expand pre-activation: dz[B*H*W][16E] (bf16), and decoder_pred's weight
 * gradient as ceil(B*H*W/32) partial rows dwd_partials[row][128] (first E valid; fold with
 * tulip_reduce_rows2).  target == NULL: dpred is the upstream gradient of pred.  target != NULL: dpred is the
 * forward's pred and the L1 gradient gscale*sign(pred-target)/numel (forward_loss backward, tulip.py:692-693;
 * gscale read from gscale_dev if non-NULL) is formed inside the kernel. */
int tulip_tail_bwd(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                   uint16_t* dz, float* dwd_partials, int B, int H, int W, int E, const float* target,
                   const float* gscale_dev, float gscale, hipStream_t stream);
/* The same backward WITHOUT the (B,16E,H,W) gradient tensor (100 MB at batch 8, written once and read twice): both
 * consumers recompute it.  tulip_tail_bwd_dgrad (the chain): dxn[B*H*W][E] (bf16) = dz . We, the input of norm_up's
 * backward (autograd of tulip.py:724-731), and the decoder_pred partial rows exactly as tulip_tail_bwd writes them.
 * tulip_tail_wgrad (beside the chain): the expand conv's weight / bias gradient as tulip_tail_wgrad_splits(B,H,W,E) plain
 * slabs slabs_w[split][16E*E], slabs_b[split][16E] (fold with tulip_reduce_rows_multi, stride 16E*E resp. 16E).
 * dpred / target / gscale as in tulip_tail_bwd.  E % 16 == 0, E <= 128 (tulip_tail_fused_bwd_supported). */
int tulip_tail_fused_bwd_supported(int E);
int tulip_tail_bwd_dgrad(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                         uint16_t* dxn, float* dwd_partials, int B, int H, int W, int E, const float* target,
                         const float* gscale_dev, float gscale, hipStream_t stream);
/* tulip_tail_bwd_dgrad with norm_up's backward (autograd of tulip.py:720) in its epilogue: instead of dxn it writes what
 * tulip_layernorm_bwd would -- dx[B*H*W][E] (fp32, overwritten), optionally dx_bf16 = bf16(dx * cast_rowscale[token /
 * cast_rows_per_sample]) and ceil(B*H*W/32) partial rows ln_partials[row][2E] = [dgamma | dbeta] (fold with
 * tulip_reduce_rows_multi).  x / mean / rstd / gamma: the LayerNorm's input rows, saved statistics and weight. */
int tulip_tail_bwd_dgrad_ln(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                            float* dwd_partials, int B, int H, int W, int E, const float* target, const float* gscale_dev,
                            float gscale, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, uint16_t* dx_bf16, const float* cast_rowscale, int cast_rows_per_sample,
                            float* ln_partials, hipStream_t stream);
/* tulip_tail_fwd with norm_up (tulip.py:720: LayerNorm(E) of the fp32 rows x, eps) in front -- xn / mean / rstd are written
 * for the backward -- and, when loss_partials != NULL, the partial sums of forward_loss (tulip.py:690-700) behind it:
 * loss_partials[2*wg] = sum |pred - target|, [2*wg+1] = sum |expm1(pred) - expm1(target)| (log_transform) over the
 * workgroup's 32 tokens, wg < ceil(B*H*W/32); tulip_l1_loss_final folds them into losses[2] (mean over n elements). */
int tulip_tail_fwd_ln(const float* x, const float* gamma, const float* beta, float eps, uint16_t* xn, float* mean,
                      float* rstd, const uint16_t* We, const float* be, const float* wd, float* pred, const float* target,
                      float* loss_partials, int log_transform, int B, int H, int W, int E, hipStream_t stream);
int tulip_l1_loss_final(const float* partials, float* losses, int nblocks, int64_t n, int log_transform,
                        hipStream_t stream);
int tulip_tail_wgrad_splits(int B, int H, int W, int E);
int tulip_tail_wgrad(const uint16_t* xn, const uint16_t* We, const float* be, const float* wd, const float* dpred,
                     float* slabs_w, float* slabs_b, int B, int H, int W, int E, const float* target,
                     const float* gscale_dev, float gscale, hipStream_t stream);


/* The non-default decoder alternates PatchExpanding (tulip.py:126-140, patch_unmerging=False; P = 2, Cn = C/2) and
 * FinalPatchExpanding (tulip.py:144-159, pixel_shuffle=False; P = upscale_factor, Cn = embed_dim):
 * Linear(no bias) -> 'B H W (P1 P2 C) -> B (H P1) (W P2) C' -> LayerNorm(Cn).  The Linear is tulip_gemm_bf16 with
 * TULIP_EPI_F32 into y [B*H*W][P*P*Cn] fp32; a fine token is a contiguous Cn-slice of a row of y, so the rearrange is
 * the OUTPUT addressing of these kernels.  fwd: LayerNorm of every slice -> out_bf16 (optional; fine-token order,
 * row pitch ld, e.g. the first half of a skip-concat buffer, tulip.py:715) and/or pred[fine token] =
 * sum_c dotw[c] * bf16(LayerNorm out)[c] (optional: decoder_pred, tulip.py:731, in_chans == 1, the (B,PH,PW,Cn)
 * tensor is never stored).  mean / rstd: [B*H*W*P*P] in the (coarse token, p) order of y.
 * bwd: upstream gradient either as bf16 rows dy_fine (fine-token order, row pitch ld) or, when dotw != NULL, as
 * dpred[fine token] (then d(LayerNorm out)[c] = dpred * dotw[c]) -> dy_nat = bf16 d(y) in y's layout (operand of the
 * Linear's dgrad / wgrad GEMMs) and tulip_expand_norm_bwd_partial_rows(...) partial rows of
 * [dgamma[Cn] | dbeta[Cn] | d(dotw)[Cn]] (stride 3*Cn; fold with tulip_reduce_rows_multi).  Cn % 4 == 0, Cn <= 768. */
int tulip_expand_norm_fwd(const float* y, const float* gamma, const float* beta, uint16_t* out_bf16, int ld,
                          const float* dotw, float* pred, float* mean, float* rstd, int B, int H, int W, int P, int Cn,
                          float eps, hipStream_t stream);
int tulip_expand_norm_bwd(const uint16_t* dy_fine, int ld, const float* dpred, const float* dotw, const float* y,
                          const float* mean, const float* rstd, const float* gamma, const float* beta, uint16_t* dy_nat,
                          float* partials, int B, int H, int W, int P, int Cn, hipStream_t stream);
int tulip_expand_norm_bwd_partial_rows(int B, int H, int W, int P);

/* forward_loss (tulip.py:690-700): losses[0]=mean|pred-target|, losses[1]=mean|expm1(pred)-expm1(target)|
 * (or a copy of losses[0] when log_transform==0).  partials: scratch of 2*1024 floats. Deterministic. */
int tulip_l1_loss_fwd(const float* pred, const float* target, float* partials, float* losses, int64_t n,
                      int log_transform, hipStream_t stream);
/* dpred = gscale * sign(pred-target)/n ; gscale read from device memory if gscale_dev!=NULL */
int tulip_l1_loss_bwd(const float* pred, const float* target, const float* gscale_dev, float gscale, float* dpred,
                      int64_t n, hipStream_t stream);

/* Fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics, main_lidar_upsampling.py:283):
 * hyper (device, 8 floats) = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale}.
 * decay_mask64[i/64] bit 0 selects weight decay for elements of 64-float block i/64 (timm's grouping,
 * main:282: decay only for ndim>1 parameters), bit 1 SKIPS the block (its step was taken elsewhere:
 * tulip_wgrad_group_adamw); NULL = decay everywhere, skip nothing.  Also refreshes the bf16 shadow.
 * zero_grad != 0: g is cleared after it has been consumed (optimizer.zero_grad(), engine_upsampling.py:97). */
int tulip_adamw(float* p, float* g, float* m, float* v, uint16_t* p_bf16, int64_t n, const float* hyper,
                const uint8_t* decay_mask64, int zero_grad, hipStream_t stream);
/* The same step for the 64-float blocks blocks[0..nblocks) only (device int32 block indices; decay_mask64 is indexed by block,
 * bit 1 is ignored): the end-of-step launch when most tensors were stepped where their gradient was completed
 * (tulip_wgrad_group_adamw, tulip_reduce_rows_multi_adamw) -- nothing is scanned. */
int tulip_adamw_blocks(float* p, float* g, float* m, float* v, uint16_t* p_bf16, const int32_t* blocks, int nblocks,
                       const float* hyper, const uint8_t* decay_mask64, int zero_grad, hipStream_t stream);

/* DropPath multipliers of one step (tulip.py:25-29; timm drop_path: keep a sample's residual branch with
 * probability keep, scale kept branches by 1/keep): scale[slot*B+b] = floor(keep[slot] + u)/keep[slot] with
 * u ~ U[0,1) from a counter-based generator keyed by (seed, *counter, index).  *counter is advanced by one:
 * replaying the launch from a HIP graph draws fresh numbers every step.  u_out (optional) receives the draws. */
int tulip_drop_path_scales(const float* keep, float* scale, float* u_out, int nslots, int B, uint64_t seed,
                           uint64_t* counter, hipStream_t stream);

/* Gradient L2 norm read-out (misc.py:317-329 get_grad_norm_, taken before the optimizer step, misc.py:303):
 * out[0] = sqrt(sum g[i]^2) * scale * (scale_dev ? scale_dev[0] : 1).  partials: scratch of 1024 doubles.
 * Fixed partition and fold order: deterministic. */
int tulip_grad_norm(const float* g, int64_t n, double* partials, const float* scale_dev, float scale, float* out,
                    hipStream_t stream);

/* Range-image input transforms in one pass (util/datasets.py): ScaleTensor :138-142, FilterInvalidPixels
 * :144-151, DownsampleTensor :117-125, DownsampleTensorWidth :127-135, LogTransform :68-70,
 * RandomRollRangeMap :96-107, composed as build_{durlar,kitti,carla}_upsampling_dataset do (:244-340).
 * raw: B sensor images in metres, element type float32 (raw_dtype 0) or float16 (raw_dtype 1); pixel (i,j) of
 * image b is raw[b*batch_stride + base_offset + i*row_stride + j*col_stride] (strides in elements, may be
 * negative), so the (H,W,2) .npy payload is read in place with col_stride 2 (npy_loader :175-179 takes channel 0)
 * and the .rimg payload with row_stride -1, col_stride -s0, base_offset s0*s1-1 (rimg_loader :181-193:
 * transpose + flip).  hi (B,1,H,W) and/or lo (B,1,H/row_factor,W/col_factor); either may be NULL.
 *   v = raw*scale; if gate: v = (min_range<=v<=max_range) ? v : 0; if log_transform: v = log1p(v);
 *   hi[i][(j+roll)%W] = v;  lo[(i-row_phase)/row_factor][((j-col_phase)/col_factor+roll)%(W/col_factor)] = v
 *   for rows/columns with (i-row_phase)%row_factor==0, (j-col_phase)%col_factor==0. */
int tulip_range_prep(const void* raw, int raw_dtype, int64_t batch_stride, int64_t row_stride, int64_t col_stride,
                     int64_t base_offset, float* hi, float* lo, int B, int H, int W, int row_factor, int col_factor,
                     int row_phase, int col_phase, float scale, int gate, float min_range, float max_range,
                     int log_transform, int roll_shift, hipStream_t stream);

/* ---- evaluation post-processing and 3-D metrics (engine_upsampling.py:126-355,361-608; util/evaluation.py) ---- */

/* MC-dropout aggregate (engine_upsampling.py:421-426): preds (passes, n) -> out (n):
 * mean over the passes, zeroed where the unbiased std exceeds noise_threshold*mean. */
int tulip_mc_aggregate(const float* preds, int passes, int64_t n, float noise_threshold, float* out,
                       hipStream_t stream);

/* One image (engine_upsampling.py:176-251): pred/hi (H,W), lo (h,w) in the model's value space.
 * expm1 of all three when log_transform (:178-181); pred = (gate_min<=pred<=gate_max) ? pred : 0 (:183-190);
 * mae_out[0] = mean|pred-hi| (:192-193); when w==W: mae_out[1] = mean over rows 0::H/h of |pred-lo| (:226-228) and
 * those rows of pred are replaced by lo (:230), else mae_out[1] = 0 (:208-209); keep_close>0 zeroes pixels of both
 * images above it (:247-249).  pred_img/hi_img (H,W): the images the point clouds are made from.
 * partials: scratch of 2*1024 doubles. */
int tulip_eval_postprocess(const float* pred, const float* hi, const float* lo, float* pred_img, float* hi_img,
                           double* partials, float* mae_out, int H, int W, int h, int w, int log_transform,
                           float gate_min, float gate_max, float keep_close, hipStream_t stream);

/* Range image -> (H*W,3) float32 points, row-major pixel order (evaluation.py img_to_pcd_kitti :52-87 and
 * img_to_pcd_carla :90-116): r = img*max_range; x = (sin_h[j]*cos_v[i])*r; y = (cos_h[j]*cos_v[i])*r;
 * z = sin_v[i]*r, float32 products in that order (bit exact against numpy given the same tables). */
int tulip_range_to_xyz(const float* img, const float* sin_h, const float* cos_h, const float* sin_v,
                       const float* cos_v, float max_range, int H, int W, float* xyz, hipStream_t stream);

/* DurLAR / OS1-128 (evaluation.py img_to_pcd_durlar :21-50) -> (H*W,3) float64 points:
 * t = float32(img*max_range) - origin_offset (float32); col_tables = [cos(enc+az) | sin(enc+az) | o*cos(enc) |
 * o*sin(enc)] (4*W), row_tables = [cos(el) | sin(el)] (2*H); x = -((t*c0[u])*cos_el[v] + c2[u]),
 * y = -((t*c1[u])*cos_el[v] + c3[u]), z = t*sin_el[v] + z_offset, unfused float64; pixel (v,u) is written at
 * index v*W + (u + W - row_offset[v]) % W (idx_from_px :21-24). */
int tulip_range_to_xyz_durlar(const float* img, const double* col_tables, const double* row_tables,
                              const int32_t* row_offset, float max_range, float origin_offset, double z_offset, int H,
                              int W, double* xyz, hipStream_t stream);

/* Voxel IoU / precision / recall / F1 of two clouds (engine_upsampling.py:254-271; evaluation.py
 * voxelize_point_cloud :148-160, calculate_metrics :162-175).  Points (n,3) float32 (is_f64=0) or float64; voxel
 * of p = ((p - min)/grid_size).astype(int) with min/max over BOTH clouds, evaluated in the clouds' own type.
 * The two occupancy grids are bitmaps of bitmap_words 32-bit words each, all zero on entry and on exit (cleared
 * again point by point); out = {iou, precision, recall, f1, dims0, dims1, dims2, status}; status 1 (metrics NaN)
 * when dims0*dims1*dims2 exceeds 32*bitmap_words.  scratch: 6*1024+16 doubles, zero before first use. */
int tulip_voxel_metrics(const void* pcd_pred, int64_t n_pred, const void* pcd_gt, int64_t n_gt, int is_f64,
                        double grid_size, uint32_t* bitmap_pred, uint32_t* bitmap_gt, int64_t bitmap_words,
                        double* scratch, double* out, hipStream_t stream);

/* Chamfer distance as the reference reduces it (evaluation.py:125-135): out[0] = mean(dist_a) + mean(dist_b),
 * dist_a[i] = min_j |a_i - b_j|^2 and dist_b[j] = min_i |b_j - a_i|^2 in float32 (float64 clouds are cast), the
 * semantics of the ChamferDistance CUDA extension the reference imports (not vendored there).
 * scratch: 2*1024 doubles. */
int tulip_chamfer_sq(const void* a, int64_t na, const void* b, int64_t nb, int is_f64, float* dist_a, float* dist_b,
                     double* scratch, double* out, hipStream_t stream);

/* KITTI point cloud -> range image, the producer of the .npy files the loaders read
 * (kitti_utils/sample_kitti_dataset.py create_range_map :24-66, parameters :139-145): points (n,4) float32
 * [x,y,z,intensity]; row = rint((atan2(z,sqrt(x^2+y^2))*180/pi + ang_start_y)/ang_res_y),
 * col = -trunc((atan2(x,y)*180/pi - 90)/ang_res_x) + cols/2 (wrapped once), all float32 in numpy's op order; a pixel
 * keeps the LAST point (in point order) that falls into it; range outside [min_range,max_range] -> 0, the
 * intensity of such a point is kept (as in the reference); out (rows,cols,2) float32 [range, intensity].
 * winner: scratch of rows*cols int32, all -1 on entry and on exit. */
int tulip_kitti_range_map(const float* points, int64_t n, int rows, int cols, float ang_start_y, float ang_res_y,
                          float ang_res_x, float max_range, float min_range, int32_t* winner, float* out,
                          hipStream_t stream);

/* Forward of one whole Swin block at embed width 96 (3 heads x 32, window 2x8, MLP 96->384->96) in ONE launch:
 * SwinTransformerBlock.forward (tulip.py:338-352) = norm1 -> WindowAttention.forward (:282-324, shifted when
 * shift_h/shift_w != 0, mask when `masked`) -> +DropPath residual -> norm2 -> Mlp.forward (:194-200) -> +residual.
 * Tokens (B,H,W,96) fp32, H even, W % 64 == 0.  Every tensor the backward reads is written exactly as the separate
 * kernels write it: xn1 [M][96] bf16, qkv [M][288] bf16, attn_out [M][96] bf16, x1 [M][96] fp32, xn2 [M][96] bf16,
 * fc1_pre / fc1_act [M][384] bf16, mean/rstd [M] fp32.  drop_scale_* : per-sample DropPath multipliers or NULL.
 * Inference form: ALL eleven saved-activation pointers (x1, xn1, qkv, attn_out, xn2, fc1_pre, fc1_act, mean1, rstd1, mean2,
 * rstd2) NULL -- only x_out (and the wide kernels' out_bf16) is written.  tulip_swin96_block_fwd only: qkv and fc1_pre
 * (both) NULL with the other nine given -- the lean training form for the recomputing backward (tulip_swin96_bwd_desc).
 * Any other mix of NULL and non-NULL is an argument error. */
typedef struct tulip_swin96_desc {
    const float* x_in; float* x1; float* x_out;
    void* xn1; void* qkv; void* attn_out; void* xn2; void* fc1_pre; void* fc1_act;
    float* mean1; float* rstd1; float* mean2; float* rstd2;
    const void* w_qkv; const void* w_proj; const void* w_fc1; const void* w_fc2;
    const float* b_qkv; const float* b_proj; const float* b_fc1; const float* b_fc2;
    const float* norm1_weight; const float* norm1_bias; const float* norm2_weight; const float* norm2_bias;
    const float* bias_table; const int32_t* rel_index; const float* drop_scale_attn; const float* drop_scale_mlp;
    int B; int H; int W; int shift_h; int shift_w; int masked; float eps;
} tulip_swin96_desc;
int tulip_swin96_block_fwd(const tulip_swin96_desc* d, hipStream_t stream);
/* diagnostic twins (tools/swin96_phases.py): stamps[(workgroup * 8 + wave) * 16 + k] = s_memtime (shader clock) at phase
 * boundary k of every wave; workgroups = tulip_swin96_bwd_partial_rows(B, H, W) */
int tulip_swin96_block_fwd_profiled(const tulip_swin96_desc* d, uint64_t* stamps, hipStream_t stream);
/* Two consecutive blocks of the stage (tulip.py:399-436: the un-shifted block d0, then the shifted block d1 on its output) in ONE
 * launch: a workgroup runs its tile of d0, publishes the 49-KB output tile (write-through stores + one flag per tile), waits for
 * the <= 4 tiles of d0 that its tile of d1 reads, and runs d1 -- no grid barrier; first-block tiles never wait.  d0->x_out must be
 * d1->x_in; B (H/2) (W/64) <= the number of CUs (one workgroup per tile, all resident; TULIP_ERR_ARG beyond); both descriptors in the same form (inference, or training with TULIP_BLOCK_FC1_GRAD); same tensors and bits as the
 * two launches.  sync: tulip_swin96_pair_sync_bytes(B, H, W) bytes, 16-byte aligned, ZERO before the first launch and owned by
 * these launches from then on (epoch-stamped flags: nothing to clear between launches or graph replays; words 2 / 3 are a test
 * hook that holds the tiles of one parity back, see csrc/swin96.hip; word 4 counts polls that gave up after ~0.5 s -- it stays 0 unless
 * a producer never ran, and the results of such a launch are undefined). */
int tulip_swin96_pair_sync_bytes(int B, int H, int W);
int tulip_swin96_pair_fwd(const tulip_swin96_desc* d0, const tulip_swin96_desc* d1, void* sync, size_t sync_bytes,
                          hipStream_t stream);

/* Backward of the same block in ONE launch (replaces, for C = 96, the chain tulip_gemm_bf16(EPI_GELU_BWD) ->
 * tulip_gemm_bf16 -> tulip_layernorm_bwd -> tulip_gemm_bf16 -> tulip_window_attn_bwd -> tulip_gemm_bf16 ->
 * tulip_layernorm_bwd; autograd of tulip.py:338-352).  dx holds d(block output) on entry and d(block input) on
 * return.  The kernel also writes the bf16 operands of the four weight-gradient GEMMs -- d_out_mlp = bf16(dx*s_mlp)
 * [M][96], d_fc1_pre [M][384], d_out_attn = bf16(d(x1)*s_attn) [M][96], d_qkv [M][288] -- and ONE partial row per
 * workgroup (tulip_swin96_bwd_partial_rows of them) for each of: norm1 / norm2 affine gradients ([rows][192] =
 * d(weight) | d(bias), folded by tulip_reduce_rows2) and the dense relative-position-bias gradient ([rows][3*256],
 * folded and scattered by tulip_reduce_rows_multi).  dx_bf16 (optional): bf16(dx * dx_bf16_scale[sample]). */
typedef struct tulip_swin96_bwd_desc {
    float* dx; const float* x_in; const float* x1;
    const void* qkv; const void* fc1_pre;
    const float* mean1; const float* rstd1; const float* mean2; const float* rstd2;
    const void* w_qkv; const void* w_proj; const void* w_fc1; const void* w_fc2;
    const float* norm1_weight; const float* norm2_weight;
    const float* bias_table; const int32_t* rel_index; const float* drop_scale_attn; const float* drop_scale_mlp;
    void* d_out_mlp; void* d_fc1_pre; void* d_out_attn; void* d_qkv;
    void* dx_bf16; const float* dx_bf16_scale;
    float* norm1_partials; float* norm2_partials; float* bias_partials;
    int B; int H; int W; int shift_h; int shift_w; int masked;
    /* Recomputation form (C = 96 only, round 4): qkv == NULL and fc1_pre == NULL -- the forward was launched without them
     * (tulip_swin96_block_fwd with qkv = fc1_pre = NULL writes 2.1 instead of 3.5 KB per token) and the backward recomputes
     * both from x_in / x1, the saved statistics and the weights it holds in LDS anyway, with the forward's own operand
     * fragments and accumulation order (the forward's bits).  Needs the four bias vectors the forward added; ignored (may be
     * NULL) when qkv / fc1_pre are given.  tulip_swinw_block_bwd does not read them. */
    const float* b_qkv; const float* b_fc1; const float* norm1_bias; const float* norm2_bias;
} tulip_swin96_bwd_desc;
int tulip_swin96_bwd_partial_rows(int B, int H, int W);
int tulip_swin96_block_bwd(const tulip_swin96_bwd_desc* d, hipStream_t stream);
int tulip_swin96_block_bwd_profiled(const tulip_swin96_bwd_desc* d, uint64_t* stamps, hipStream_t stream);

/* The same two launches for the wider stages, C = 192 and C = 384 (heads of 32, window 2x8, MLP C -> 4C -> C; H even,
 * W % 16 == 0; tulip_swinw_supported): a workgroup owns 2 or 4 neighbouring windows with one wave per head, every GEMM
 * of the block is split along its output channels across the waves, weights stream from L2 straight into MFMA
 * operand registers, activations pass through LDS (csrc/swinw.hip).  Descriptors, saved tensors and partial rows are
 * those of the C = 96 entry points with 96 -> C, 288 -> 3C, 384 -> 4C, 3*256 -> (C/32)*256 and partial rows of
 * [d(weight)[C] | d(bias)[C]].  out_bf16 (optional): bf16 copy of the block output [M][C].
 * WEIGHTS are read in fragment-major ("packed") order, written by tulip_pack_bf16_multi: the 16 x 32 block (rows 16 nt..,
 * columns 32 ks..) of a [N][K] matrix is the 1-KiB block nt*K/32 + ks, and inside it element (n, k) sits at
 * ((n%16) + 16*((k%32)/8))*8 + k%8 -- one MFMA A operand = one contiguous 1-KiB wave load (4x the L2 -> register rate
 * of row-major fragments on this chip).  Forward: w_qkv / w_proj / w_fc1 / w_fc2 = packed copies of the weights as
 * they are ([3C][C], [C][C], [4C][C], [C][4C]); backward: packed copies of their TRANSPOSES (item.transpose = 1).
 * rows % 16 == 0 and cols % 32 == 0 of the matrix being packed (after the transpose, if any); up to TULIP_PACK_MAX
 * matrices per launch. */
#define TULIP_PACK_MAX 64
typedef struct tulip_pack_item { const void* src; void* dst; int rows; int cols; int transpose; } tulip_pack_item;
int tulip_swinw_supported(int C, int H, int W);
int tulip_swinw_block_fwd(const tulip_swin96_desc* d, int C, void* out_bf16, hipStream_t stream);
/* diagnostic twin of tulip_swinw_block_fwd: stamps[(workgroup * waves + wave) * 16 + k] = s_memtime (shader clock) at
 * phase boundary k of every wave (waves = C/32; workgroups = tulip_swinw_bwd_partial_rows); tools/swinw_phases.py */
int tulip_swinw_block_fwd_profiled(const tulip_swin96_desc* d, int C, void* out_bf16, uint64_t* stamps,
                                   hipStream_t stream);
/* Split form of the forward (round 4; C = 384 where a workgroup owns ONE window, i.e. few windows: batch 8 at stage 2 is 128
 * workgroups on 256 CUs, each streaming the block's 3.5 MB of weights): two workgroups per window.  Both run the attention
 * half, each takes half of the MLP's hidden channels; the two fc2 partial sums meet in `exchange` (tulip_swinw_split_bytes
 * bytes, 16-byte aligned, zero before the first launch, left zeroed; its last 4 bytes per window are arrival tickets): the
 * last-arriving workgroup of a pair adds its partner's partial and writes the block output, nobody waits.  Same arguments,
 * tensors and results (the fc2 sum is formed as two halves: last-bit differences) as tulip_swinw_block_fwd in its training
 * form with TULIP_BLOCK_FC1_GRAD or in its inference form; stamps optional (the profiled twin).  tulip_swinw_split_bytes returns 0 where the form
 * does not exist (then call tulip_swinw_block_fwd). */
int tulip_swinw_split_bytes(int C, int B, int H, int W);
int tulip_swinw_block_fwd_split(const tulip_swin96_desc* d, int C, void* out_bf16, void* exchange, size_t exchange_bytes,
                                uint64_t* stamps, hipStream_t stream);
int tulip_swinw_bwd_partial_rows(int C, int B, int H, int W);
/* Launches of at most 256 workgroups (the whole grid resident at once) start by spreading the block's weights over the
 * L2 of each XCD (every wave touches a few KiB nobody else touches): in a training step the weights are cold, and the
 * per-wave streams would otherwise run at miss latency (tools/cold_probe.py).  TULIP_BLOCK_NO_WARM in d->masked switches
 * that off for one launch.  (tulip_gemm_bf16 has the same first touch of its weight panel: TULIP_GEMM_NO_TOUCH.) */
int tulip_swinw_block_bwd(const tulip_swin96_bwd_desc* d, int C, hipStream_t stream);
/* split form of the backward (see tulip_swinw_block_fwd_split; the same exchange buffer may serve both directions of a block):
 * the MLP half's hidden channels are halved between the two workgroups of a window, the partial sums of d(norm2 output) meet in
 * `exchange`, the last arriver runs norm2' .. norm1' alone.  Needs TULIP_BLOCK_FC1_GRAD in d->masked.  Partial rows: one per
 * WINDOW, tulip_swinw_bwd_partial_rows as for tulip_swinw_block_bwd. */
int tulip_swinw_block_bwd_split(const tulip_swin96_bwd_desc* d, int C, void* exchange, size_t exchange_bytes, hipStream_t stream);
int tulip_pack_bf16_multi(const tulip_pack_item* items, int n, hipStream_t stream);

/* The same block at the DEEP widths, C = 768 (stage 3) and C = 1536 (stage 4 of tulip_large) -- replaces, per direction, the 15 / 16
 * launch sequence tulip_layernorm_fwd -> tulip_gemm_bf16 -> tulip_window_attn_fwd -> ... of tulip.py:338-352 -- as a chain of SLICED
 * launches (csrc/swind.hip): eight workgroups per group of 1-2 windows, workgroup (group, slice) on XCD `slice`, every CU streaming
 * one eighth of a fragment-major weight matrix; the slicing alternates between heads / hidden channels and output channels, so no
 * launch holds a partial sum.  Window wh x ww = 2 x 8 or the 1 x 16 backup window (tulip.py:284-287).  Descriptors, saved tensors and
 * weight copies (fragment-major; backward: of the transposes) are those of tulip_swinw_block_fwd / _bwd.
 * Forward, `phases` bits: 1 norm1 + qkv + attention (by heads) -> attn_out; 2 proj + residual (by output channels) -> x1;
 * 4 norm2 + fc1 + GELU (by hidden channels) -> fc1_act; 8 fc2 + residual -> x_out (+ out_bf16).  The training form needs
 * TULIP_BLOCK_FC1_GRAD in d->masked (the fc1_pre buffer carries gelu'(h)); the inference form is xn1, qkv, xn2, fc1_pre and the
 * four statistics NULL -- attn_out, x1 and fc1_act pass from launch to launch and must always be given.
 * Backward, `phases` bits: 1 fc2' + GELU' (by hidden channels): d->dx (read only) -> d_out_mlp, d_fc1_pre; 2 fc1' -> d_norm_out =
 * fp32 d(norm2 output) [M][C]; the caller then runs tulip_layernorm_bwd_splitk(d_norm_out, 1, x1, mean2, rstd2, ...) (norm2',
 * residual, d_out_attn); 4 proj' + attention' (by heads): d_out_attn -> d_qkv, bias_partials [tulip_swind_groups][heads * 256];
 * 8 qkv' -> d_norm_out = fp32 d(norm1 output); the caller runs tulip_layernorm_bwd_splitk again (norm1', residual).  The
 * descriptor's norm partials, dx_bf16, x_in, x1 and statistics fields are not read by these launches.
 * stamps (optional): s_memtime per wave at the phase boundaries, [launch 0..3][workgroup][wave][16]. */
int tulip_swind_supported(int C, int H, int W, int wh, int ww);
int tulip_swind_groups(int C, int B, int H, int W, int wh, int ww);
int tulip_swind_block_fwd(const tulip_swin96_desc* d, int C, int wh, int ww, void* out_bf16, int phases, uint64_t* stamps,
                          hipStream_t stream);
int tulip_swind_block_bwd(const tulip_swin96_bwd_desc* d, int C, int wh, int ww, float* d_norm_out, int phases, uint64_t* stamps,
                          hipStream_t stream);

/* ---- the stage boundaries of the U-Net as ONE launch each (csrc/glue.hip, round 6) -------------------------------------
 * A workgroup owns a block of 16 / 32 rows for the whole boundary: the row block's activations sit in LDS as the MFMA B operand,
 * the weights stream from their FRAGMENT-MAJOR copies (tulip_pack_bf16_multi: `*_packed` = the matrix as it is, `*_t_packed` =
 * its transpose), and what the launch sequences below pass through HBM passes through LDS.  bf16 rounding points are those of
 * the sequences they replace (every GEMM operand bf16, every accumulator fp32, LayerNorm in fp32): results agree with them to
 * fp32 summation order.  Every *_supported() is 0 where the form does not exist; the caller then issues the sequence. */

/* PatchMerging.forward (tulip.py:101-106) = tulip_layernorm_fwd(merge = 1) + tulip_gemm_bf16(TULIP_EPI_F32): x (B,H,W,Cin) fp32
 * -> 2x2 gather -> LayerNorm(4 Cin) -> xm bf16 [rows][4 Cin], mean, rstd [rows] (saved for the backward) -> y fp32 [rows][2 Cin]
 * = xm . W^T (no bias); y_bf16 (optional, row pitch ld_bf16 elements): bf16 copy of y (x_save half of the next level's skip
 * concat, tulip.py:715).  rows = B H/2 W/2.  Cin in {96, 192, 384}, rows % 32 == 0. */
typedef struct tulip_merge_fwd_desc {
    const float* x; const float* gamma; const float* beta;
    const void* w_packed;          /* reduction.weight [2 Cin][4 Cin], fragment-major */
    void* xm; float* mean; float* rstd; float* y;
    void* y_bf16; int ld_bf16;
    int B, H, W, Cin; float eps;
} tulip_merge_fwd_desc;
int tulip_merge_fwd_supported(int Cin, int B, int H, int W);
int tulip_merge_fwd(const tulip_merge_fwd_desc* d, hipStream_t stream);

/* Backward of the same boundary (autograd of tulip.py:101-106, and of the x_save half of tulip.py:715 in front of it) =
 * [tulip_gemm_bf16(dy_skip . W_skip[:, Cs:], TULIP_EPI_F32 accumulate + bf16 copy)] + tulip_gemm_bf16(dyb . W_red, TULIP_EPI_BF16)
 * + tulip_layernorm_bwd(merge = 1).  (B,H,W,Cp) is the FINER stage (the LayerNorm's input x_prev), Cs = 2 Cp, rows = B H/2 W/2.
 * dy_skip != NULL: dyb = bf16(dx_in + dy_skip . W_skip[:, Cs:2Cs]) is formed and WRITTEN (operand of the reduction's weight
 * gradient; w_skip_t_packed = packed W_skip^T, a [2 Cs][Cs] matrix); dy_skip == NULL: dyb is READ (the bottleneck stage: no skip).
 * dx_prev (B,H,W,Cp) fp32 is overwritten with the LayerNorm backward's result scattered through the gather; dx_bf16 (optional) =
 * bf16(dx_prev * cast_rowscale[token / cast_rows_per_sample]); param_partials: tulip_merge_bwd_partial_rows() rows of
 * [dgamma[4 Cp] | dbeta[4 Cp]] (fold with tulip_reduce_rows_multi).  Cp in {96, 192}, rows % 32 == 0. */
typedef struct tulip_merge_bwd_desc {
    const float* dx_in; const void* dy_skip; const void* w_skip_t_packed;
    void* dyb;
    const void* w_red_t_packed;    /* packed reduction.weight^T: a [4 Cp][2 Cp] matrix */
    const float* x_prev; const float* mean; const float* rstd; const float* gamma;
    float* dx_prev; float* param_partials;
    void* dx_bf16; const float* cast_rowscale; int cast_rows_per_sample;
    int B, H, W, Cp;
} tulip_merge_bwd_desc;
int tulip_merge_bwd_supported(int Cp, int B, int H, int W);
int tulip_merge_bwd_partial_rows(int Cp, int B, int H, int W);
int tulip_merge_bwd(const tulip_merge_bwd_desc* d, hipStream_t stream);

/* PatchUnmerging.forward -> skip Linear(cat[...]) (tulip.py:117-123, :713-716) = tulip_gemm_bf16(TULIP_EPI_PIXSHUF2_F32) +
 * tulip_gemm_bf16(TULIP_EPI_F32).  x_bf16 [M][C], M = B H W coarse tokens; cat bf16 [4M][C] (fine-token order): its first half
 * [:, :C/2] = bf16(PixelShuffle(2)(x . We^T + be)) is WRITTEN (operand of the skip weight gradient), its second half (x_save)
 * is READ; out fp32 [4M][C/2] = cat . Ws^T + bs.  C in {192, 384}, M % 16 == 0. */
typedef struct tulip_unmerge_skip_desc {
    const void* x_bf16;
    const void* w_expand_packed; const float* b_expand;     /* expand.weight [2C][C] fragment-major, bias [2C] */
    void* cat;
    const void* w_skip_packed; const float* b_skip;         /* skip weight [C/2][C] fragment-major, bias [C/2] */
    float* out;
    int B, H, W, C;
} tulip_unmerge_skip_desc;
int tulip_unmerge_skip_supported(int C, int B, int H, int W);
int tulip_unmerge_skip_fwd(const tulip_unmerge_skip_desc* d, hipStream_t stream);

/* Its backward along the unmerged stream = tulip_gemm_bf16(dy_skip . W_skip[:, :C/2], TULIP_EPI_UNSHUF2_BF16) +
 * tulip_gemm_bf16(dz . We, TULIP_EPI_F32 + bf16 copy): dy_skip bf16 [4M][C/2]; dz bf16 [M][2C] is WRITTEN (operand of the expand
 * weight / bias gradient); dx fp32 [M][C] overwritten; dx_bf16 (optional) = bf16(dx * cast_rowscale[m / cast_rows_per_sample]).
 * w_skip_t_packed = packed W_skip^T ([C][C/2], rows 0 .. C/2-1 used), w_expand_t_packed = packed We^T ([C][2C]). */
typedef struct tulip_skip_unmerge_bwd_desc {
    const void* dy_skip; const void* w_skip_t_packed;
    void* dz;
    const void* w_expand_t_packed;
    float* dx; void* dx_bf16; const float* cast_rowscale; int cast_rows_per_sample;
    int B, H, W, C;
} tulip_skip_unmerge_bwd_desc;
int tulip_skip_unmerge_bwd(const tulip_skip_unmerge_bwd_desc* d, hipStream_t stream);

/* library self-description */
/* diagnostics: *dst = the 100 MHz constant device clock (s_memrealtime) when the stream reaches this point; capturable
 * (tools/step_stamps.py time-lines a captured training step with it, no tracer attached) */
int tulip_stamp_realtime(uint64_t* dst, hipStream_t stream);
/* Layout version of the structs and signatures in this header (round 3: tulip_wgrad_item, tulip_reduce_region and tulip_adamw_ref grew
 * fields, entry points were added): a caller built against another version must not bind. */
#define TULIP_ABI_VERSION 6
int tulip_abi_version(void);
const char* tulip_build_arch(void);
int tulip_dev_variants(void);      /* 1: the development build (see the conventions at the top) */

#ifdef __cplusplus
}
#endif
#endif /* TULIP_HIP_H_ */
