// Cold read bandwidth of five 3840x2160 float planes (the DoG planes of octave 0) for three access patterns:
//   flat   grid-stride 16-byte loads over the whole range
//   strip  one wave per 64-column x 32-row strip, walking down the rows of all five planes (k_detect's walk)
//   band   as strip, but a wave walks ONE row of 64 columns x 5 planes and the grid is in raster order
// Between runs a 1 GiB buffer is overwritten so that nothing of the planes is left in L2 / MALL.
// build: hipcc --offload-arch=gfx950 -O3 -o build_variants/stream_read tools/ubench/stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 3840, H = 2160, NP = 5;
__global__ void k_fill(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 255); }
__global__ void k_fill_nt(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v + (float)(i & 255), &p[i]); }
__global__ void k_flat(const f4* p, size_t n4, float* out)
{
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { f4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e30f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_strip(const float* p, float* out)
{
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strips = W / 64, cy = unit / strips, sx = unit - cy * strips;
    if (cy * 32 >= H) return;
    float acc = 0;
    for (int y0 = cy * 32; y0 < min(cy * 32 + 32, H); y0 += 4) {
        float q[4][NP];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int z = 0; z < NP; z++) q[k][z] = p[(size_t)z * W * H + (size_t)min(y0 + k, H - 1) * W + sx * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int z = 0; z < NP; z++) acc += q[k][z];
    }
    if (acc == 1.2345e30f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_band(const float* p, float* out)
{   /* raster order: workgroup = 256 columns of one row group of 4 rows, all planes */
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strips = W / 64, rg = unit / strips, sx = unit - rg * strips;
    if (rg * 4 >= H) return;
    float q[4][NP], acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int z = 0; z < NP; z++) q[k][z] = p[(size_t)z * W * H + (size_t)min(rg * 4 + k, H - 1) * W + sx * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int z = 0; z < NP; z++) acc += q[k][z];
    if (acc == 1.2345e30f) out[0] = acc;
}
int main()
{
    const size_t n = (size_t)W * H * NP, nflush = (size_t)1 << 28;
    float *p, *flush, *out;
    hipMalloc(&p, n * 4); hipMalloc(&flush, nflush * 4); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double mb = n * 4 / 1e6;
    for (int nt = 0; nt < 2; nt++)
        for (int pat = 0; pat < 3; pat++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                if (nt) hipLaunchKernelGGL(k_fill_nt, dim3(4096), dim3(256), 0, 0, p, n, (float)rep);
                else    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, n, (float)rep);
                hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, flush, nflush, (float)rep);
                hipEventRecord(e0, 0);
                if (pat == 0) hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, (const f4*)p, n / 4, out);
                if (pat == 1) hipLaunchKernelGGL(k_strip, dim3((W / 64) * ((H + 31) / 32) / 4 + 1), dim3(256), 0, 0, p, out);
                if (pat == 2) hipLaunchKernelGGL(k_band, dim3((W / 64) * ((H + 3) / 4) / 4 + 1), dim3(256), 0, 0, p, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s-written planes, %-5s read: %7.1f us  %6.2f TB/s\n", nt ? "nt" : "st", pat == 0 ? "flat" : pat == 1 ? "strip" : "band", best * 1e3, mb / best / 1e3);
        }
    /* warm: no flush, planes (166 MB) just written */
    for (int pat = 0; pat < 2; pat++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, n, (float)rep);
            hipEventRecord(e0, 0);
            if (pat == 0) hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, (const f4*)p, n / 4, out);
            if (pat == 1) hipLaunchKernelGGL(k_strip, dim3((W / 64) * ((H + 31) / 32) / 4 + 1), dim3(256), 0, 0, p, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("just written (no flush), %-5s read: %7.1f us  %6.2f TB/s\n", pat == 0 ? "flat" : "strip", best * 1e3, mb / best / 1e3);
    }
    return 0;
}
