// The practical floor of one Gaussian level launch: a chain of plain 16-byte copies plane l-1 -> plane l of a
// 3840 x 2160 float plane (33.2 MB read + 33.2 MB written per launch), six planes in one arena like an octave, each launch
// reading what the previous one wrote -- the cache state of the level chain.  Shapes: grid-stride over 2048 x 256 lanes,
// one 16-byte chunk per lane (8100 workgroups), and 4 chunks in flight per lane.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/plane_copy tools/ubench/plane_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 3840, H = 2160;
__global__ __launch_bounds__(256) void k_stride(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_one(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n4) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_four(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
    const size_t b = blockIdx.x * (size_t)1024 + threadIdx.x;
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (b + 256 * k < n4) v[k] = s[b + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; k++) if (b + 256 * k < n4) d[b + 256 * k] = v[k];
}
int main()
{
    const size_t n = (size_t)W * H, n4 = n / 4;
    float* p; hipMalloc(&p, n * 4 * 6);
    hipMemset(p, 0, n * 4 * 6);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 3; shape++) {
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int l = 1; l < 6; l++) {
                const f4* s = (const f4*)(p + (l - 1) * n); f4* d = (f4*)(p + l * n);
                if (shape == 0) hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, s, d, n4);
                if (shape == 1) hipLaunchKernelGGL(k_one, dim3((n4 + 255) / 256), dim3(256), 0, 0, s, d, n4);
                if (shape == 2) hipLaunchKernelGGL(k_four, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, s, d, n4);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("shape %d (%s): %.2f us per level copy incl. launch gap = %.2f TB/s\n", shape, shape == 0 ? "grid-stride" : shape == 1 ? "one chunk per lane" : "four chunks per lane", ms * 1000 / 5, 2 * n * 4 / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
