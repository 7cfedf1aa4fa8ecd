// Floor of a dependent launch chain on one stream: per-launch time of (a) an empty kernel, (b) one workgroup doing a
// dependent load -> LDS -> barrier -> store round trip, (c) the same with a 232-byte by-value argument; each as plain
// stream launches and as one hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float taps[32]; const float* src; float* dst; int w, h, pitch, pad[19]; };
__global__ void k_empty() {}
__global__ void k_spin(long long cycles) /* keeps the queue busy while the host enqueues the chain behind it */
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(10);
}
__global__ __launch_bounds__(256) void k_touch(const float* __restrict__ src, float* __restrict__ dst)
{
    __shared__ float s[256];
    s[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    dst[threadIdx.x] = s[255 - threadIdx.x] + 1.0f;
}
__global__ __launch_bounds__(256) void k_touch_big(Big a)
{
    __shared__ float s[256];
    s[threadIdx.x] = a.src[threadIdx.x] * a.taps[threadIdx.x & 31];
    __syncthreads();
    a.dst[threadIdx.x] = s[255 - threadIdx.x] + 1.0f;
}
int main()
{
    const int N = 200;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float *p, *q; CK(hipMalloc(&p, 4096)); CK(hipMalloc(&q, 4096)); CK(hipMemset(p, 0, 4096)); CK(hipMemset(q, 0, 4096));
    Big big{}; for (int i = 0; i < 32; i++) big.taps[i] = 1.0f;
    for (int mode = 0; mode < 3; mode++) {
        auto launch = [&](int i) {
            float* x = (i & 1) ? q : p; float* y = (i & 1) ? p : q;
            if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
            else if (mode == 1) hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, s, x, y);
            else { big.src = x; big.dst = y; hipLaunchKernelGGL(k_touch_big, dim3(1), dim3(256), 0, s, big); }
        };
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < N; i++) launch(i);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("mode %d stream launches: %.2f us per launch\n", mode, ms * 1e3 / N);
        }
        for (int rep = 0; rep < 3; rep++) { /* the same with every launch already queued when the device gets to the chain */
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000LL); /* 3 ms at 100 MHz */
            CK(hipEventRecord(a, s));
            for (int i = 0; i < N; i++) launch(i);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("mode %d stream launches, pre-queued: %.2f us per launch\n", mode, ms * 1e3 / N);
        }
        if (mode) { /* the chain really is a chain: 2 N increments survive */
            float h[256]; CK(hipMemcpy(h, (N & 1) ? q : p, sizeof(h), hipMemcpyDeviceToHost));
            printf("mode %d value after the chains: %.0f (a dependent chain counts every launch)\n", mode, h[7]);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("mode %d graph: %.2f us per node\n", mode, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
