// Period of a dependent chain of the library's own small-plane level launches (popsift_hip::launch_blur on a tiny plane),
// pre-queued behind a spinning kernel.  Build (repo root):
//   hipcc --offload-arch=gfx950 -O3 -Ipopsift_amd/csrc -Iinclude -o tools/ubench/blur_chain tools/ubench/blur_chain.hip \
//         -Lpopsift_amd -lpopsift_hip -Wl,-rpath,$PWD/popsift_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "sift_types.h"
#include "kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(long long cycles)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(10);
}
int main(int argc, char** argv)
{
    using namespace popsift_hip;
    const int N = 200;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    for (int dim = 0; dim < 3; dim++) {
        const int w = dim == 0 ? 15 : dim == 1 ? 120 : 480, h = dim == 0 ? 9 : dim == 1 ? 68 : 270;
        const int pitch = (w + 63) / 64 * 64;
        float *p, *q; CK(hipMalloc(&p, (size_t)pitch * h * 4)); CK(hipMalloc(&q, (size_t)pitch * h * 4));
        CK(hipMemset(p, 0, (size_t)pitch * h * 4)); CK(hipMemset(q, 0, (size_t)pitch * h * 4));
        for (int span : {6, 9, 14}) {
            BlurArgs a{};
            a.w = w; a.h = h; a.pitch = pitch;
            a.tiles_x = (w + blur_tile_w() - 1) / blur_tile_w(); a.tiles_y = (h + 31) / 32;
            for (int i = 0; i < span; i++) a.taps.g[i] = 1.0f / (2 * span - 1);
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000LL);
                CK(hipEventRecord(ea, s));
                for (int i = 0; i < N; i++) {
                    a.src = (i & 1) ? q : p; a.dst = (i & 1) ? p : q;
                    CK(launch_blur(a, 0, span, 32, s));
                }
                CK(hipEventRecord(eb, s));
                CK(hipStreamSynchronize(s));
                float ms = 0; CK(hipEventElapsedTime(&ms, ea, eb));
                if (rep == 2) printf("plane %d x %d, span %d (%d tiles): %.2f us per launch\n", w, h, span, a.tiles_x * a.tiles_y, ms * 1e3 / N);
            }
        }
        CK(hipFree(p)); CK(hipFree(q));
    }
    return 0;
}
