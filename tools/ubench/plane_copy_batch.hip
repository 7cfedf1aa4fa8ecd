// The floor of a level launch IN A BATCH: sixteen 3840 x 2160 float planes copied per launch (plane l-1 -> plane l of sixteen
// images, 531 MB read + 531 MB written), level after level like the batched pyramid -- every launch reads what the launch before
// wrote, but sixteen planes are more than the 256 MB last-level cache holds, so the sources come from HBM.  Compare
// tools/ubench/plane_copy.hip (one image: the source is cache-resident, 9.5 us per plane).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/plane_copy_batch tools/ubench/plane_copy_batch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 3840, H = 2160, NB = 16, L = 6;
__global__ __launch_bounds__(256) void k_four(const f4* __restrict__ s0, f4* __restrict__ d0, size_t n4, size_t img_stride4)
{
    const f4*    s = s0 + blockIdx.y * img_stride4;
    f4*          d = d0 + blockIdx.y * img_stride4;
    const size_t b = blockIdx.x * (size_t)1024 + threadIdx.x;
    f4           v[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (b + 256 * k < n4) v[k] = s[b + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (b + 256 * k < n4) d[b + 256 * k] = v[k];
}
// read only: six planes of sixteen images (3.2 GB), the way k_detect reads them (every value once), the sum kept alive by a
// store that never happens
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ s0, float* __restrict__ out, size_t n4, size_t img_stride4, int planes)
{
    const size_t b = blockIdx.x * (size_t)1024 + threadIdx.x;
    f4           acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int pl = 0; pl < planes; pl++) {
        const f4* s = s0 + blockIdx.y * img_stride4 + (size_t)pl * n4;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (b + 256 * k < n4) acc += s[b + 256 * k];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}
int main()
{
    const size_t n = (size_t)W * H, n4 = n / 4;
    float*       p;
    if (hipMalloc(&p, n * 4 * L * NB) != hipSuccess) return 1; /* image-major: L planes per image */
    (void)hipMemset(p, 0, n * 4 * L * NB);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int l = 1; l < L; l++)
            hipLaunchKernelGGL(k_four, dim3((n4 + 1023) / 1024, NB), dim3(256), 0, 0, (const f4*)(p + (l - 1) * n), (f4*)(p + l * n), n4, (size_t)L * n4);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%d planes per launch: %.2f us per plane = %.2f TB/s (read + written)\n", NB, ms * 1000 / (L - 1) / NB, 2 * n * 4 / (ms / (L - 1) / NB * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 4; rep++) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_read, dim3((n4 + 1023) / 1024, NB), dim3(256), 0, 0, (const f4*)p, p, n4, (size_t)L * n4, L);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("read only, %d planes of %d images in one launch: %.2f us per image = %.2f TB/s\n", L, NB, ms * 1000 / NB, (double)L * n * 4 / (ms / NB * 1e-3) / 1e12);
    }
    return 0;
}
