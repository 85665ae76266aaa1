// Fragment helpers shared by the fused Swin-block kernels that stream their weights from L2 (csrc/swinw.hip: C = 192 / 384,
// csrc/swind.hip: C = 768 / 1536): LDS activation layout, fragment-major weight stream, small packing helpers.
#pragma once
#include <type_traits>
#include "common.h"

namespace {

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
__device__ __forceinline__ int region(int x, int X, int wsz, int ssz) {       // create_mask slices, tulip.py:261-266
    return (ssz == 0 || x >= X - ssz) ? 2 : (x >= X - wsz ? 1 : 0);
}
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    return __builtin_bit_cast(bf16x4, (u32x2_t){pack_bf16x2(a, b), pack_bf16x2(c, d)});
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 trr(const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)p); }

// activations in LDS: [k tile of 32 channels][T tokens][64 B], 16-B chunks XOR-swizzled by token
template <int T>
__device__ __forceinline__ void put4(unsigned char* base, int tok, int c, bf16x4 v) {
    *(bf16x4*)(base + (c >> 5) * (T * 64) + tok * 64 + ((((c >> 3) & 3) ^ swz4(tok)) << 4) + (c & 7) * 2) = v;
}
template <int T>
__device__ __forceinline__ bf16x8 frag(const unsigned char* base, int kt, int tok, int gq) {
    return *(const bf16x8*)(base + kt * (T * 64) + tok * 64 + ((gq ^ swz4(tok)) << 4));
}

// Weights are read in FRAGMENT-MAJOR ("packed") order: the 16 x 32 block of rows 16 nt.., k 32 ks.. of a [N][K] matrix is
// the 1-KiB block (nt * K/32 + ks) and lane (t, gq) owns its bytes [16 (t + 16 gq), +16) = W[16 nt + t][32 ks + 8 gq ..
// +7] -- so the A operand of one MFMA is ONE fully contiguous 1-KiB wave load.  Measured on this chip (tools/
// probe_stream.hip): a global_load_dwordx4 whose lanes walk down the rows of a row-major matrix (the natural fragment
// order) moves 16 B/clk per CU however many are in flight, lanes along 64..512-B row pieces 20-33 B/clk, a contiguous
// KiB 47-60 B/clk; with the row-major layout the kernels below spent 60-70 % of their time in the weight stream.
//
// acc[i][g] += W[tile i][k] . X[token tile g][k]^T over KSTEPS 32-deep steps.  wtile[i]: the tile's first block, already
// offset by this lane's 8 elements; the weight stream runs PF steps ahead of the MFMAs.
template <int NTILE, int KSTEPS, int PF>
struct WStream {
    bf16x8 ring[PF][NTILE];
    const bf16_t* wt[NTILE];
    // the first PF steps of the stream.  Issued EARLY -- before the epilogue stores / the barrier of the phase in front:
    // vmcnt retires in order, so a weight load issued after a batch of stores cannot be waited for without draining
    // those stores, while one issued before them costs the wait nothing.
    __device__ __forceinline__ void start() {
        static_for<PF>([&](auto P_) {
            constexpr int p = decltype(P_)::value;
            if constexpr (p < KSTEPS) {
#pragma unroll
                for (int i = 0; i < NTILE; ++i) ring[p][i] = *(const bf16x8*)(wt[i] + 512 * p);
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
                for (int i = 0; i < NTILE; ++i) ring[ks % PF][i] = *(const bf16x8*)(wt[i] + 512 * (ks + PF));
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
template <int NTILE, int G, int KSTEPS, int PF, int T>
__device__ __forceinline__ void wave_gemm(f32x4 (&acc)[NTILE][G], const bf16_t* const (&wtile)[NTILE],
                                          const unsigned char* act, int t, int gq) {
    WStream<NTILE, KSTEPS, PF> w;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) w.wt[i] = wtile[i];
    w.start();
    w.template run<G, T>(acc, act, t, gq);
}
template <int NTILE, int G>
__device__ __forceinline__ void zero(f32x4 (&acc)[NTILE][G]) {
#pragma unroll
    for (int i = 0; i < NTILE; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 ld4(const float* p) { const float4 v = *(const float4*)p; return (f32x4){v.x, v.y, v.z, v.w}; }
// first packed block of row tile `nt` of a matrix with K columns, at this lane's 8 elements
__device__ __forceinline__ const bf16_t* wtile_ptr(const bf16_t* w, int nt, int K, int lane) {
    return w + ((size_t)nt * (K >> 5)) * 512 + lane * 8;
}

}  // namespace
