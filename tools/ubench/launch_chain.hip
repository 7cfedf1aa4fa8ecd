// Floor of a dependent launch chain on one stream: per-launch time of (0) an empty kernel, (1) one workgroup doing a
// dependent load -> LDS -> barrier -> store round trip, (2) the same with a 232-byte by-value argument, (3) 1024 lanes,
// 32 KB of LDS and two argument structs, (4) three such kernels taking turns, (5, 6) as 3, 4 with 256 lanes and 1 KB of
// LDS; each as plain stream launches, pre-queued stream launches and one hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
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
/* what a level launch of the small octaves adds to that: 1024 lanes, 32 KB of LDS, two argument structs, three code
 * objects taking turns */
template <int V>
__global__ __launch_bounds__(1024) void k_fat(Big a, Big b)
{
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) s[i] = a.src[i & 255] * a.taps[V];
    __syncthreads();
    if (threadIdx.x < 256) b.dst[threadIdx.x] = s[8191 - threadIdx.x * V] + 1.0f;
}
template <int V>
__global__ __launch_bounds__(256) void k_lean(Big a, Big b) /* the same with 256 lanes and 1 KB of LDS */
{
    __shared__ float s[256];
    s[threadIdx.x] = a.src[threadIdx.x] * a.taps[V];
    __syncthreads();
    b.dst[threadIdx.x] = s[255 - threadIdx.x] + 1.0f;
}
int main()
{
    const int N = 200;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float *p, *q; CK(hipMalloc(&p, 4096)); CK(hipMalloc(&q, 4096)); CK(hipMemset(p, 0, 4096)); CK(hipMemset(q, 0, 4096));
    Big big{}; for (int i = 0; i < 32; i++) big.taps[i] = 1.0f;
    for (int mode = 0; mode < 7; mode++) {
        auto launch = [&](int i) {
            float* x = (i & 1) ? q : p; float* y = (i & 1) ? p : q;
            if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
            else if (mode == 1) hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, s, x, y);
            else if (mode == 2) { big.src = x; big.dst = y; hipLaunchKernelGGL(k_touch_big, dim3(1), dim3(256), 0, s, big); }
            else {
                big.src = x; big.dst = y;
                const int which = (mode == 3 || mode == 5) ? 1 : 1 + i % 3; /* modes 4, 6: three kernels take turns */
                if (mode <= 4) {
                    if (which == 1) hipLaunchKernelGGL(k_fat<1>, dim3(1), dim3(1024), 0, s, big, big);
                    else if (which == 2) hipLaunchKernelGGL(k_fat<2>, dim3(1), dim3(1024), 0, s, big, big);
                    else hipLaunchKernelGGL(k_fat<3>, dim3(1), dim3(1024), 0, s, big, big);
                } else {
                    if (which == 1) hipLaunchKernelGGL(k_lean<1>, dim3(1), dim3(256), 0, s, big, big);
                    else if (which == 2) hipLaunchKernelGGL(k_lean<2>, dim3(1), dim3(256), 0, s, big, big);
                    else hipLaunchKernelGGL(k_lean<3>, dim3(1), dim3(256), 0, s, big, big);
                }
            }
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
    /* does the number of workgroups (all XCDs take part from 8 on) change the floor?  k_fat<1>, pre-queued */
    for (int wgs : {1, 8, 46, 170, 512}) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000LL);
            CK(hipEventRecord(a, s));
            for (int i = 0; i < N; i++) {
                big.src = (i & 1) ? q : p; big.dst = (i & 1) ? p : q;
                hipLaunchKernelGGL(k_fat<1>, dim3(wgs), dim3(1024), 0, s, big, big);
            }
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("k_fat, %d workgroups, pre-queued: %.2f us per launch\n", wgs, ms * 1e3 / N);
        }
    }
    return 0;
}
