// The shader clock a long, chip-filling kernel actually runs at (MI355X): s_memtime ticks of a wave against the wall time of
// the launch (HIP events).  rocm-smi shows 2.2 GHz for the timed loop of bench.py, but it samples every 0.2 s; cycle stamps
// inside co-resident blur workgroups said 1.5-1.6 GHz.  Kernels: v_fma_f32 only; v_fma + LDS atomics
// (ds_add_u64, 1 per 24 FMAs, as in k_descriptor); each at 8 waves per SIMD on every CU, ~5 ms per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_under_load.hip -o tools/ubench/clock_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA8 "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
             "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"

template <int MODE>
__global__ __launch_bounds__(64) void k_load(long long* out, float a, float b, int iters)
{
    __shared__ unsigned long long s_h[64 * 4];
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    s_h[threadIdx.x] = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        asm volatile(FMA8 FMA8 FMA8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (MODE == 1) atomicAdd(&s_h[(threadIdx.x * 5 + it) & 255], (unsigned long long)it);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = (long long)s_h[3];
    if (threadIdx.x == 0) out[1 + blockIdx.x] = t1 - t0;
}

int main()
{
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const int blocks = cus * 4 * 8;
    long long* d;
    (void)hipMalloc(&d, sizeof(long long) * (1 + blocks));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("%-28s %10s %12s %10s %s\n", "kernel", "wall us", "ticks/wave", "ticks/ns", "(ticks per 24 FMAs per SIMD)");
    for (int mode = 0; mode < 2; mode++) {
        for (int iters : {20000, 100000, 500000, 100000, 20000}) {
            for (int rep = 0; rep < 2; rep++) {
                (void)hipMemset(d, 0, sizeof(long long) * (1 + blocks));
                (void)hipEventRecord(e0, 0);
                if (mode == 0)
                    hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.9999f, iters);
                else
                    hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.9999f, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
            }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(1 + blocks);
            (void)hipMemcpy(h.data(), d, sizeof(long long) * (1 + blocks), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < blocks; i++) sum += (double)h[1 + i];
            const double ticks = sum / blocks;
            printf("%-28s %10.1f %12.0f %10.3f %8.2f\n", mode ? "24 v_fma + 1 ds_add_u64" : "24 v_fma", ms * 1e3, ticks, ticks / (ms * 1e6),
                   ticks / ((double)iters * 8));
        }
    }
    return 0;
}
