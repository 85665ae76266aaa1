// Probe (dev tool, not product): what one CU can stream from L2 into registers with global_load_dwordx4, by access
// pattern and waves per workgroup.  Every workgroup streams the SAME buffer (the fused Swin block's situation: all
// workgroups read the block's weights), one workgroup per CU (LDS-limited), DEPTH loads in flight per wave.
//   mode 0: lane (t = l&15, gq = l>>4) reads 16 B of row t at k-chunk gq: 16 rows x 64 B per instruction, rows PITCH
//           bytes apart (the MFMA A fragment of a row-major [out][in] weight)
//   mode 1: fully contiguous 1 KiB per instruction (fragment-major "packed" weights)
//   mode 2: 8 rows x 128 B per instruction (full cache lines, rows PITCH apart)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_stream.hip -o tools/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int DEPTH>
__global__ void stream_kernel(const unsigned char* __restrict__ buf, size_t bytes_per_wave, int pitch, unsigned* sink,
                              unsigned long long* cycles) {
    __shared__ unsigned char pad[96 * 1024];                 // one workgroup per CU
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned char* base = buf + (size_t)wid * bytes_per_wave;
    const int n = (int)(bytes_per_wave / 1024);              // load instructions per wave
    u32x4 acc = {0, 0, 0, 0};
    auto addr = [&](int i) -> const unsigned char* {
        if (MODE == 1) return base + (size_t)i * 1024 + lane * 16;
        if (MODE == 0) {                                     // tile of 16 rows, k-step i % (pitch/64)
            const int ksteps = pitch / 64, tile = i / ksteps, ks = i % ksteps;
            return base + ((size_t)tile * 16 + (lane & 15)) * pitch + ks * 64 + (lane >> 4) * 16;
        }
        if (MODE == 2) {
            const int per = pitch / 128, tile = i / per, ks = i % per;   // 8 rows x 128 B
            return base + ((size_t)tile * 8 + (lane & 7)) * pitch + ks * 128 + (lane >> 3) * 16;
        }
        // MODE 3 / 4 / 5: SEG = 256 / 512 / 64 contiguous bytes per row, consecutive lanes along the row
        constexpr int SEG = MODE == 3 ? 256 : (MODE == 4 ? 512 : 64), LPR = SEG / 16, ROWS = 64 / LPR;
        const int per = pitch / SEG, tile = i / per, ks = i % per;
        return base + ((size_t)tile * ROWS + lane / LPR) * pitch + ks * SEG + (lane % LPR) * 16;
    };
    u32x4 ring[DEPTH];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = *(const u32x4*)addr(d);
    for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const u32x4 v = ring[d];
            if (i + d + DEPTH < n) ring[d] = *(const u32x4*)addr(i + d + DEPTH);
            acc ^= v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 0x12345 && pad[lane] == 7) sink[0] = acc.y + acc.z + acc.w;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wid] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char* name, int waves, size_t total_bytes, int pitch, int blocks, const unsigned char* buf, unsigned* sink,
         unsigned long long* cyc) {
    const size_t per_wave = total_bytes / waves / 16384 * 16384;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) stream_kernel<MODE, DEPTH><<<blocks, waves * 64>>>(buf, per_wave, pitch, sink, cyc);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) stream_kernel<MODE, DEPTH><<<blocks, waves * 64>>>(buf, per_wave, pitch, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * waves);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double bytes_cu = (double)per_wave * waves;
    printf("%-28s waves %2d depth %2d blocks %3d: %7.1f us/launch, per-wave %8.0f ticks, %6.1f B/tick/CU, chip %6.2f TB/s\n", name,
           waves, DEPTH, blocks, ms * 100, mean, bytes_cu / mean, bytes_cu * blocks / (ms * 1e-4) / 1e12);
}

int main() {
    const size_t total = 885 * 1024;                          // the C = 192 block's weights
    unsigned char* buf; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&buf, 8 << 20); hipMemset(buf, 1, 8 << 20); hipMalloc(&sink, 64); hipMalloc(&cyc, 8 * 4096);
    for (int waves : {6, 12}) {
        run<1, 8>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<1, 12>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<1, 16>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<1, 18>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<1, 20>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<1, 24>("contiguous 1 KiB", waves, total, 384, 256, buf, sink, cyc);
        run<0, 12>("rows x 64 B (pitch 384)", waves, total, 384, 256, buf, sink, cyc);
    }
    run<1, 12>("3.5 MB contiguous", 12, 3584 * 1024, 768, 64, buf, sink, cyc);
    run<1, 16>("3.5 MB contiguous", 12, 3584 * 1024, 768, 64, buf, sink, cyc);
    run<1, 20>("3.5 MB contiguous", 12, 3584 * 1024, 768, 64, buf, sink, cyc);
    return 0;
}
