/* blur_common.h -- small device helpers shared by the blur kernels (pyramid.hip, blur_march.hip). Internal. */
#pragma once
#include <hip/hip_runtime.h>

namespace popsift_hip {

constexpr int BLUR_TW = 128; /* tile / strip width (outputs) */

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* XCD-aware tile order: blocks b and b+8 share an XCD (and its 4 MiB L2), so
 * give each XCD a contiguous run of tiles -- neighbouring tiles share halos. */
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

}  // namespace popsift_hip
