// Microbenchmark: cost of LDS atomic adds (f32 / u32) by address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int PATTERN>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    __shared__ float h[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) h[wave][i] = 0;
    __syncthreads();
    int idx;
    if (PATTERN == 0) idx = lane;              // distinct banks
    else if (PATTERN == 1) idx = 0;            // all same address
    else if (PATTERN == 2) idx = lane >> 3;    // 8 lanes per address
    else idx = (lane * 37) & 127;              // pseudo-random over 128
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) atomicAdd(&h[wave][idx], 1.0f);
        else if (KIND == 1) atomicAdd((unsigned*)&h[wave][idx], 1u);
        else h[wave][idx] += 1.0f;             // plain RMW (racy; timing only)
        idx = (idx + (PATTERN == 3 ? 17 : 0)) & 127;
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = h[wave][lane];
}
template <int KIND, int PATTERN> void run(const char* name)
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 1024;
    k<KIND, PATTERN><<<blocks, 256>>>(d, 10);
    hipEventRecord(a); k<KIND, PATTERN><<<blocks, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per CU: blocks*4*iters/256 ; cycles at ~2.4GHz
    double winstr_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%-28s %8.3f ms  -> %.1f cycles per wave-instr per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / winstr_per_cu);
    hipFree(d);
}
int main()
{
    run<0, 0>("f32 atomic distinct"); run<0, 1>("f32 atomic same-addr"); run<0, 2>("f32 atomic 8/addr"); run<0, 3>("f32 atomic random128");
    run<1, 0>("u32 atomic distinct"); run<1, 1>("u32 atomic same-addr"); run<1, 2>("u32 atomic 8/addr"); run<1, 3>("u32 atomic random128");
    run<2, 0>("plain rmw distinct");  run<2, 3>("plain rmw random128");
    return 0;
}
