// dev probe (hipcc --offload-arch=gfx950 tools/probe_ext_event.hip -o tools/probe_ext_event): an EXTERNAL event-record node inside a captured
// HIP graph -- does a stream outside the graph that waits for the event run behind the node of the LATEST launch, and before the graph's tail?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void add1(float* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f; }
__global__ void copyk(const float* p, float* q, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i]; }
int main() {
    const size_t n = 1 << 26, m = 1 << 20;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, m * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t cap, ext;
    CK(hipStreamCreate(&cap)); CK(hipStreamCreate(&ext));
    hipEvent_t ev, t0, tg, to;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&tg)); CK(hipEventCreate(&to));
    add1<<<1024, 256, 0, cap>>>(b, n); copyk<<<256, 256, 0, ext>>>(a, c, m);
    CK(hipDeviceSynchronize()); CK(hipMemset(b, 0, n * 4));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 4; ++i) add1<<<1024, 256, 0, cap>>>(a, n);       // a += 4 per launch
    CK(hipEventRecordWithFlags(ev, cap, hipEventRecordExternal));
    for (int i = 0; i < 40; ++i) add1<<<1024, 256, 0, cap>>>(b, n);      // the tail: much longer than the head
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<float> h(m);
    bool ok = true;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(t0, cap));
        CK(hipGraphLaunch(ge, cap));
        CK(hipEventRecord(tg, cap));
        CK(hipStreamWaitEvent(ext, ev, 0));
        copyk<<<256, 256, 0, ext>>>(a, c, m);
        CK(hipEventRecord(to, ext));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), c, m * 4, hipMemcpyDeviceToHost));
        float tgm, tom;
        CK(hipEventElapsedTime(&tgm, t0, tg)); CK(hipEventElapsedTime(&tom, t0, to));
        bool good = true;
        for (size_t i = 0; i < m; i += 4097) good = good && h[i] == 4.0f * (it + 1);
        ok = ok && good;
        printf("launch %d: outside copy saw %.0f (want %.0f) %s; graph %.3f ms, outside work done at %.3f ms\n", it, h[0], 4.0f * (it + 1), good ? "ok" : "WRONG", tgm, tom);
    }
    printf("external event nodes order outside streams: %s\n", ok ? "yes" : "NO");
    return 0;
}
