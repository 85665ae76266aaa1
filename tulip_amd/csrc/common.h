// Shared device helpers for the TULIP gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TULIP_OK 0
#define TULIP_ERR_ARG (-1)
#define TULIP_ERR_LAUNCH (-2)
// -DTULIP_DEV_VARIANTS=1 (libtulip_hip_dev.so): the forms a training / inference step never launches -- the profiled twins (in-kernel
// clock stamps), the recomputing C = 96 backward, the backward's split form at C = 384, the wide blocks' training forms that save h
// instead of gelu'(h).  The product library keeps their entry points (one symbol set, one header) and answers TULIP_ERR_NOT_BUILT.
#ifndef TULIP_DEV_VARIANTS
#define TULIP_DEV_VARIANTS 0
#endif
#define TULIP_ERR_NOT_BUILT (-3)

#define TULIP_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return -(1000 + (int)e__);      \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bfloat16, round-to-nearest-even (same as torch): gfx950 has the conversion in hardware, two values per
// instruction (v_cvt_pk_bf16_f32); the integer sequence it replaces was ~7 VALU instructions per element, and the fused
// block kernels convert 1344 elements per token.
typedef float tulip_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 tulip_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((tulip_f32x2_t){lo, hi}, tulip_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// L2 warm-up touch: 16 bytes per lane at `gptr` are pulled through the caches and DROPPED into `lds_scratch` (a KiB of LDS
// per wave instruction, lane i lands at scratch + 16 i; nothing reads it, so every wave may share one KiB).
// global_load_lds_dwordx4 via the compiler's builtin: no VGPR destination for the register allocator to move underneath an
// in-flight load, and the compiler's own vmcnt accounting covers it.
// The 4-byte form: every lane names its own address, lane i's word lands at scratch + 4 i (a 256-byte sink) -- one instruction
// pulls up to 64 cache lines towards the XCD's L2 (the backward kernels' prefetch of the activations their forward saved).
__device__ __forceinline__ void warm_touch4(const void* gptr, void* lds_scratch) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_scratch, 4, 0, 0);
}
__device__ __forceinline__ void warm_touch16(const void* gptr, void* lds_scratch) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_scratch, 16, 0, 0);
}

// One AdamW step (torch.optim.AdamW semantics, decoupled decay) of four elements, shared by adamw_kernel (csrc/elementwise.hip)
// and the weight-gradient write-out that takes the step in place (csrc/gemm.hip): floating-point contraction is off inside,
// so both call sites execute the same operations and a tensor ends up with the same bits whichever of them stepped it.
struct AdamwCoef { float b1, b2, eps, decay, step, rbc2, gs; };
__device__ __forceinline__ AdamwCoef adamw_coef(const float* __restrict__ hyper, bool decay_on) {
    // hyper = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale}
    const float lr = hyper[0], wd = hyper[4];
    return AdamwCoef{hyper[1], hyper[2], hyper[3], decay_on ? 1.0f - lr * wd : 1.0f, lr / hyper[5], rsqrtf(hyper[6]), hyper[7]};
}
__device__ __forceinline__ void adamw_step4(float4& pp, float4& mm, float4& vv, const float4 gg, const AdamwCoef c) {
#pragma clang fp contract(off)
    float* P = (float*)&pp; float* M = (float*)&mm; float* V = (float*)&vv; const float* G = (const float*)&gg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gr = G[k] * c.gs;
        M[k] = c.b1 * M[k] + (1.0f - c.b1) * gr;
        V[k] = c.b2 * V[k] + ((1.0f - c.b2) * gr) * gr;
        const float denom = sqrtf(V[k]) * c.rbc2 + c.eps;
        P[k] = P[k] * c.decay - c.step * (M[k] / denom);
    }
}

// DropPath draws of one step by ONE 256-thread workgroup (tulip_drop_path_scales; also the first workgroup of the patch-embedding
// forward, tulip_patch_embed_fwd_draw, so that a training step does not start with a launch of its own for 224 numbers):
// scale[i] = floor(keep[slot] + u) / keep[slot], u ~ U[0,1) from a counter-based generator keyed by (seed, *counter, i).
struct DropDraw { const float* keep; float* scale; float* u_out; int nslots, B; unsigned long long seed; unsigned long long* counter; };
__device__ __forceinline__ uint64_t drop_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void drop_draw_block(const DropDraw& d) {
    const unsigned long long c = *d.counter;
    const int n = d.nslots * d.B;
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t r = drop_mix64(drop_mix64(d.seed + 0x9E3779B97F4A7C15ull * (c + 1)) ^ (0xD1B54A32D192ED03ull * (uint64_t)(i + 1)));
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
        const float k = d.keep[i / d.B];
        d.scale[i] = floorf(k + u) / k;
        if (d.u_out) d.u_out[i] = u;
    }
    __syncthreads();
    if (threadIdx.x == 0) *d.counter = c + 1;
}

// Sum of the `nslab` split-K partial slabs of one float4, in slab order (the order every fold of these slabs uses: same bits).
// Four loads are issued before the first add -- a loop of load / add pairs with a run-time trip count is a chain of nslab
// memory latencies, and the kernels that fold (3-4 slabs, a few rows per workgroup) are nothing but that chain.
__device__ __forceinline__ float4 fold_slabs4(const float* __restrict__ p, size_t stride, int nslab) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    auto group = [&](int s0) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(p + (size_t)min(s0 + u, nslab - 1) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + u < nslab) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    };
    group(0);
    for (int s0 = 4; s0 < nslab; s0 += 4) group(s0);
    return a;
}

// The flat optimizer buffers of a step taken outside adamw_kernel (tulip_adamw_ref): an element is addressed by the offset of its
// gradient from g0; mask64 (optional): one byte per 64 elements, bit 0 = decoupled weight decay applies (NULL: everywhere).
struct AdamRef { const float* hyper; const float* g0; float* p0; float* m0; float* v0; bf16_t* pb0; const uint8_t* mask64; };

// A bf16 MFMA operand that was JUST packed by vector-ALU instructions (v_cvt_pk_bf16_f32 behind an fma): pin 8 wait states
// between the pack and the MFMAs that read it.  Measured on gfx950 / ROCm 7.2 (tools/det_tail_instep.py): without them the
// head's weight-gradient kernel returned, about once in 200 launches INSIDE the training step (never in isolation), one
// 16-channel block of a slab that differed in the 5th digit.  The cause is not pinned down further (other kernels have the
// same pack -> MFMA distance and are reproducible); this fence, or a compare/select form of the arithmetic in front of the
// pack, each made 800 repeats bit-identical.  The in/out operand makes the nops a data dependence between the pack and its
// consumer, so they cannot be scheduled away.
__device__ __forceinline__ bf16x8 mfma_operand_fence(bf16x8 v) {
    typedef uint32_t u32x4_f __attribute__((ext_vector_type(4)));
    u32x4_f t = __builtin_bit_cast(u32x4_f, v);
    asm volatile("s_nop 7" : "+v"(t));
    return __builtin_bit_cast(bf16x8, t);
}

// A store whose data nobody reads before this launch ends (the activations a fused block saves for its backward, the
// operands it leaves for the weight-gradient launches): TULIP_STORE_LATE selects the cache policy of those stores --
// 0 plain, 1 non-temporal (global_store ... nt).  Plain stores stay dirty in the XCD's L2 until the end-of-kernel
// write-back (guide row `boundary`: + dirty bytes / 6 TB/s behind the launch).
#ifndef TULIP_STORE_LATE
#define TULIP_STORE_LATE 0
#endif
template <class V>
__device__ __forceinline__ void store_late(V* p, V v) {
#if TULIP_STORE_LATE == 1
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void store_late(float4* p, float4 v) {
    store_late((f32x4*)p, (f32x4){v.x, v.y, v.z, v.w});
}

// Optimizer state (parameter, moments, bf16 shadow) where the step is taken inside a weight-gradient write-out or a fold: every
// element is read once and written once per step, 22 B per parameter.  TULIP_ADAM_NT = 1: non-temporal loads and stores (bit 0)
// for it; bit 1: the split-K slabs / partial rows a fold reads (read once) as non-temporal loads.
#ifndef TULIP_ADAM_NT
#define TULIP_ADAM_NT 3          // same-box A/B of the step (profiles/r4_ab_nt_stores.txt): 0 -> 1: -14 us, 0 -> 3: -12 us at batch 8; batch 64 -0.2 % with 3
#endif
__device__ __forceinline__ float4 ld_state(const float* p) {
#if TULIP_ADAM_NT & 1
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *(const float4*)p;
#endif
}
__device__ __forceinline__ void st_state(float* p, float4 v) {
#if TULIP_ADAM_NT & 1
    __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4*)p);
#else
    *(float4*)p = v;
#endif
}
__device__ __forceinline__ void st_state_bf16x4(bf16_t* p, uint2 v) {
    typedef uint32_t u32x2_s __attribute__((ext_vector_type(2)));
#if TULIP_ADAM_NT & 1
    __builtin_nontemporal_store((u32x2_s){v.x, v.y}, (u32x2_s*)p);
#else
    *(uint2*)p = v;
#endif
}
__device__ __forceinline__ float4 ld_partial(const float* p) {
#if TULIP_ADAM_NT & 2
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *(const float4*)p;
#endif
}

// A load of an activation the forward saved for this backward launch (read once, by one workgroup): TULIP_LOAD_SAVED_NT = 1
// makes it non-temporal (the block's weights, which every workgroup streams from L2, are what should stay there).
#ifndef TULIP_LOAD_SAVED_NT
#define TULIP_LOAD_SAVED_NT 0
#endif
template <class V>
__device__ __forceinline__ V ld_saved(const V* p) {
#if TULIP_LOAD_SAVED_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// Two adjacent 16-column tiles of one output row in the MFMA accumulator layout -- lane (t, gq) holds 4 bf16 of tile A at
// columns 4 gq and 4 bf16 of tile B at columns 16 + 4 gq -- leave as ONE 16-byte store per lane instead of two 8-byte
// ones: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the other, after which an
// even-gq lane holds columns [4 gq, 4 gq + 8) of tile A and an odd-gq lane columns [4 (gq - 1), + 8) of tile B.  The
// fused block kernels are store-ISSUE bound in their write-outs: same bytes, same addresses, half the instructions.
template <bool LATE = false>
__device__ __forceinline__ void store_bf16_tile_pair(bf16_t* rowp, bf16x4 A, bf16x4 B, int gq) {
    typedef uint32_t u32x2_c __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_c2 __attribute__((ext_vector_type(4)));
    const u32x2_c a = __builtin_bit_cast(u32x2_c, A), b = __builtin_bit_cast(u32x2_c, B);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    const int col = (gq & 1) ? 16 + 4 * (gq - 1) : 4 * gq;
    if constexpr (LATE) store_late((u32x4_c2*)(rowp + col), (u32x4_c2){r0[0], r1[0], r0[1], r1[1]});
    else *(u32x4_c2*)(rowp + col) = (u32x4_c2){r0[0], r1[0], r0[1], r1[1]};
}

// ---- fp8 (OCP e4m3, the gfx950 format) for the optional fp8 attention scores (BASELINE configs[4]) ------------------
// eight bf16 values -> the 64-bit fp8 operand of v_mfma_f32_16x16x32_fp8_fp8 (v_cvt_pk_fp8_f32: round to nearest even)
__device__ __forceinline__ long bf16x8_to_fp8(bf16x8 v) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)v[0]), bf2f((bf16_t)v[1]), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)v[2]), bf2f((bf16_t)v[3]), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)v[4]), bf2f((bf16_t)v[5]), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)v[6]), bf2f((bf16_t)v[7]), hi, true);
    return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// the same values as bf16 again (every e4m3 value is a bf16 value): what the backward multiplies dS with, so that the
// gradient is the one of the function the forward computed (straight-through across the rounding)
__device__ __forceinline__ bf16x8 round_through_fp8(bf16x8 v) {
    const long p = bf16x8_to_fp8(v);
    const int lo = (int)(p & 0xffffffff), hi = (int)((unsigned long long)p >> 32);
    const tulip_f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const tulip_f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const uint32_t w0 = pack_bf16x2(a.x, a.y), w1 = pack_bf16x2(b.x, b.y), w2 = pack_bf16x2(c.x, c.y), w3 = pack_bf16x2(d.x, d.y);
    typedef uint32_t u32x4_c __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(bf16x8, (u32x4_c){w0, w1, w2, w3});
}
// (the `masked` argument of the attention / block kernels is a bit set: TULIP_ATTN_MASKED | TULIP_ATTN_FP8, tulip_hip.h)
#ifndef TULIP_ATTN_MASKED
#define TULIP_ATTN_MASKED 1
#define TULIP_ATTN_FP8 2
#endif
#ifndef TULIP_BLOCK_FC1_GRAD
#define TULIP_BLOCK_FC1_GRAD 4
#endif

// a / d for a >= 0, d > 0 with d uniform over the launch (a kernel argument): the token grids, tokens per sample etc.
// are powers of two in every configuration of the reference, and a 32-bit integer divide is ~40 VALU instructions
// (a 64-bit one > 100) -- the scalar test picks a shift when it can.
__device__ __forceinline__ int fast_div(int a, int d) {
    return (d & (d - 1)) == 0 ? a >> (31 - __builtin_clz(d)) : a / d;
}

// sum / max over a power-of-two group of WIDTH lanes (WIDTH <= 64)
// The exchange with lane ^ 1, ^ 2, and the two mirror steps that complete a 16-lane butterfly are DPP modifiers of a
// VALU move (no LDS round trip; __shfl_xor is a ds_bpermute_b32, ~100 cycles of latency per step in a dependent
// chain).  Only the steps across 16-lane rows (^ 16, ^ 32) still go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// v (op) v[lane ^ 16] and v (op) v[lane ^ 32] without the LDS round trip of __shfl_xor (ds_bpermute_b32, ~100 cycles in a dependent
// chain -- the softmax of one attention head is four of them in a row): gfx950's v_permlane16_swap / v_permlane32_swap exchange
// the odd 16-lane rows (the upper 32 lanes) of one register with the even rows (the lower 32 lanes) of another; fed the same
// value twice they return [even-row value, odd-row value] ([lower-half value, upper-half value]) in every lane.  op is
// commutative: the same bits as the shuffle form.
template <class OP>
__device__ __forceinline__ float xor16_reduce(float v, OP op) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
template <class OP>
__device__ __forceinline__ float xor32_reduce(float v, OP op) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
// sum / max over the four 16-lane rows of a wave (lanes t, t + 16, t + 32, t + 48): every lane gets the result
__device__ __forceinline__ float rows_sum(float v) {
    auto add = [](float a, float b) { return a + b; };
    return xor32_reduce(xor16_reduce(v, add), add);
}
__device__ __forceinline__ float rows_max(float v) {
    auto mx = [](float a, float b) { return fmaxf(a, b); };
    return xor32_reduce(xor16_reduce(v, mx), mx);
}
template <int WIDTH, class OP>
__device__ __forceinline__ float group_reduce(float v, OP op) {
    static_assert(WIDTH == 2 || WIDTH == 4 || WIDTH == 8 || WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "WIDTH");
    v = op(v, dpp_move<0xB1>(v));                          // quad_perm [1,0,3,2]: lane ^ 1
    if (WIDTH >= 4) v = op(v, dpp_move<0x4E>(v));          // quad_perm [2,3,0,1]: lane ^ 2
    if (WIDTH >= 8) v = op(v, dpp_move<0x141>(v));         // row_half_mirror: the other quad of the 8-lane half
    if (WIDTH >= 16) v = op(v, dpp_move<0x140>(v));        // row_mirror: the other half of the 16-lane row
    // (the last two steps stay on __shfl_xor here: the permlane form is the same arithmetic -- tools/probe_permlane.hip -- but it
    // moves the compiler's fma contraction around the call in the LayerNorm / patch-embedding kernels, and the tiny L1-loss
    // fixtures, whose gradient is a sign function of the prediction, are sensitive to single-ulp changes there)
    if (WIDTH >= 32) v = op(v, __shfl_xor(v, 16, 64));
    if (WIDTH >= 64) v = op(v, __shfl_xor(v, 32, 64));
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
    return group_reduce<WIDTH>(v, [](float a, float b) { return a + b; });
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
    return group_reduce<WIDTH>(v, [](float a, float b) { return fmaxf(a, b); });
}

// erf-GELU (nn.GELU default, tulip.py:183,196) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7 (+ 1 ulp of the hardware reciprocal), two orders below the bf16 rounding of the stored result):
// one exp, one rcp, six FMAs -- libm's erff was ~40 % of the fc1-forward / fc2-dgrad GEMM time.  exp(-z^2), z = x/sqrt(2), is
// shared between erf and the Gaussian term of the derivative.
__device__ __forceinline__ void gelu_terms(float x, float& erf_z, float& gauss) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));   // v_rcp_f32 (1 ulp); the IEEE divide is 10 more instructions
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    gauss = __expf(-az * az);                       // = exp(-x^2/2)
    erf_z = copysignf(1.0f - poly * gauss, z);
}
// two elements per lane: the polynomial / product chain maps onto v_pk_fma_f32 / v_pk_mul_f32 (rcp and exp stay scalar)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_terms2(f32x2 x, f32x2& erf_z, f32x2& gauss) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    const f32x2 den = __builtin_elementwise_fma(az, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 poly = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){0.254829592f, 0.254829592f});
    poly = poly * t;
    const f32x2 m = -az * az;
    gauss = (f32x2){__expf(m.x), __expf(m.y)};
    const f32x2 e = __builtin_elementwise_fma(-poly, gauss, (f32x2){1.0f, 1.0f});
    erf_z = (f32x2){copysignf(e.x, z.x), copysignf(e.y, z.y)};
}
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x) {
    f32x2 e, g;
    gelu_terms2(x, e, g);
    return x * 0.5f * (e + 1.0f);
}
__device__ __forceinline__ f32x2 gelu_exact_grad2(f32x2 x) {
    f32x2 e, g;
    gelu_terms2(x, e, g);
    return __builtin_elementwise_fma(x * 0.39894228040143267794f, g, (e + 1.0f) * 0.5f);
}
// both at once: the forward of a fused block that hands gelu'(h) to its backward (TULIP_BLOCK_FC1_GRAD)
__device__ __forceinline__ void gelu_exact_and_grad2(f32x2 x, f32x2& y, f32x2& dy) {
    f32x2 e, g;
    gelu_terms2(x, e, g);
    const f32x2 e1 = e + 1.0f;
    y = x * 0.5f * e1;                                                            // gelu_exact2's operation order: same bits
    dy = __builtin_elementwise_fma(x * 0.39894228040143267794f, g, e1 * 0.5f);    // gelu_exact_grad2's
}
__device__ __forceinline__ float gelu_exact(float x) {
    float e, g;
    gelu_terms(x, e, g);
    return 0.5f * x * (1.0f + e);
}
__device__ __forceinline__ float gelu_exact_grad(float x) {
    float e, g;
    gelu_terms(x, e, g);
    return 0.5f * (1.0f + e) + x * 0.39894228040143267794f * g;
}
