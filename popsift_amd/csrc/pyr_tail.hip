/*
 * pyr_tail.hip -- the smallest octaves of the pyramid in ONE launch (gfx950, wave64).
 *
 * Octaves whose plane is a few thousand pixels are latency, not throughput: pyramid.hip filters them with one or two
 * workgroups per level launch, 3 dependent launches of ~5 us per octave (levels 1 .. L-3 feed the next octave) plus the
 * trailing levels -- 12 launches and ~60 us for the octaves from 120 x 68 down of a 1080p image, a quarter of a single
 * image's pyramid stage for 0.1 % of its pixels.  The reference gives every octave a stream of its own for the same
 * reason (s_pyramid_build.cu:549-588).
 *
 * Here ONE workgroup of 1024 lanes per image takes every octave from the first one that fits it down to the last: the
 * plane lives in LDS (two buffers with a replicated border, so that "clamp addressing" is an ordinary read), every level
 * is a horizontal pass A -> B and a vertical pass B -> A with two workgroup barriers -- no launch boundary, no
 * trip through global memory -- and each finished level is stored to its plane in HBM for the detection / keypoint
 * kernels.  Level L-3, sampled at every second pixel, is kept in a third LDS buffer and becomes level 0 of the next octave
 * (get_by_2_pick_every_second, s_pyramid_build.cu:50-71).
 *
 * Arithmetic: gauss::absoluteSource::horiz / ::vert (s_pyramid_build_aa.cu:17-91) in their order of operations -- centre
 * tap first and (left + right) * g pairs in the horizontal pass; outermost tap first, upper then lower sample, centre last
 * in the vertical pass; explicit fmaf, -ffp-contract=off: planes bit-identical to the tile kernels' and the oracle's.
 */
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sift_types.h"
#include "kernels.h"
#include "blur_common.h"

namespace popsift_hip {

namespace {

constexpr int TNT = 1024;           /* lanes of the workgroup */
constexpr int TPAD = PYR_TAIL_PAD;  /* border of the LDS plane on every side (>= the largest halo, a multiple of 4) */
constexpr int TPLANE = PYR_TAIL_PLANE_FLOATS;
constexpr int TNEXT = PYR_TAIL_PLANE_FLOATS / 4 + 256; /* level 0 of the next octave, compact */

/* workgroup barrier that waits for this wave's LDS operations only: the stores of the finished level to HBM -- which nothing
 * in this launch reads -- stay in flight (__syncthreads() drains them: ~1.5 us of write latency at each of the four
 * barriers of a level, 65 instead of 25 us for the tail of a 1080p image) */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* One level: A (level l-1, border replicated in x) -> B (horizontal pass, then rows 0 / h-1 repeated above / below:
 * intm(x, clamp(y +- k))) -> A (level l) + the plane in HBM.  want_next: also every second pixel into nx (compact, row pitch
 * w2) and into level 0 of the next octave in HBM.
 *
 * The tap loops are ROLLED and there is one body for all filter lengths (the taps come from LDS): a kernel's instruction
 * cache starts cold, this workgroup runs every piece of its code a few times only, and it waits for every cache line of
 * code it walks into.  The first version -- eight fully unrolled HALO instances like the tile kernels', 27 KB of code --
 * took ~5 us for the FIRST level through each instance whatever the size of the plane: 29 us for the ten levels of the two
 * smallest octaves (30 x 17 and 15 x 9 pixels). */
__device__ __forceinline__ void tail_level(float* __restrict__ A, float* __restrict__ B, float* __restrict__ nx, bool want_next,
                                           const float* __restrict__ g, int halo, int w, int h, int PW, float* __restrict__ dst,
                                           int pitch, float* __restrict__ next0, int next_pitch)
{
    const int tid = threadIdx.x;
    const int cw = (w + 3) >> 2; /* 16-byte chunks per row */
    const int PW4 = PW >> 2;

    /* horizontal pass, one output per item: centre tap first, then (left + right) * g from the outermost pair inwards
     * (s_pyramid_build_aa.cu:32-48) */
    for (int it = tid; it < h * w; it += TNT) {
        const int    y = it / w, x = it - y * w;
        const float* r = &A[(TPAD + y) * PW + TPAD + x];
        float        acc = r[0] * g[0];
#pragma unroll 4
        for (int k = halo; k > 0; k--) acc = fmaf(r[-k] + r[k], g[k], acc);
        B[(TPAD + y) * PW + TPAD + x] = acc;
    }
    lds_barrier();
    for (int it = tid; it < 2 * halo * cw; it += TNT) {
        const int r = it / cw, c = it - r * cw;
        const int ysrc = r < halo ? 0 : h - 1, ydst = r < halo ? -1 - r : h + (r - halo);
        *reinterpret_cast<v4f*>(&B[(TPAD + ydst) * PW + TPAD + 4 * c]) = *reinterpret_cast<const v4f*>(&B[(TPAD + ysrc) * PW + TPAD + 4 * c]);
    }
    lds_barrier();
    /* vertical pass, one row of 4 outputs per item: outermost tap first, upper then lower sample, centre last
     * (s_pyramid_build_aa.cu:69-86) */
    for (int it = tid; it < h * cw; it += TNT) {
        const int  y = it / cw, c = it - y * cw;
        const v4f* p = reinterpret_cast<const v4f*>(&B[(TPAD + y) * PW + TPAD + 4 * c]);
        v4f        acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
        for (int k = halo; k > 0; k--) {
            const v4f   up = p[-k * PW4], dn = p[k * PW4];
            const float gk = g[k];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = fmaf(up[q], gk, acc[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = fmaf(dn[q], gk, acc[q]);
        }
        {
            const v4f   ce = p[0];
            const float g0 = g[0];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = fmaf(ce[q], g0, acc[q]);
        }
        *reinterpret_cast<v4f*>(&A[(TPAD + y) * PW + TPAD + 4 * c]) = acc;
        float* d = &dst[(size_t)y * pitch + 4 * c];
        if (4 * c + 3 < w) {
            *reinterpret_cast<v4f*>(d) = acc;
        } else { /* the row's last, partial chunk: its lanes beyond the plane hold nothing (the horizontal pass stops at w) */
            d[0] = acc.x;
            if (4 * c + 1 < w) d[1] = acc.y;
            if (4 * c + 2 < w) d[2] = acc.z;
        }
        if (want_next && (y & 1) == 0) {
            /* pixel (2x, 2y): the next octave is ceil(w / 2) wide, so 2x <= w - 1 and the reference's min() never clamps */
            const int w2 = (w + 1) >> 1, x2 = 2 * c;
            nx[(y >> 1) * w2 + x2] = acc.x;
            next0[(size_t)(y >> 1) * next_pitch + x2] = acc.x;
            if (4 * c + 2 < w) {
                nx[(y >> 1) * w2 + x2 + 1] = acc.z;
                next0[(size_t)(y >> 1) * next_pitch + x2 + 1] = acc.z;
            }
        }
    }
    lds_barrier();
}

/* columns left / right of the plane repeat its first / last column (rows 0 .. h-1): data(clamp(x +- k), y) */
__device__ __forceinline__ void tail_pad_cols(float* __restrict__ A, int w, int h, int PW)
{
    for (int it = threadIdx.x; it < h * 2 * TPAD; it += TNT) {
        const int y = it / (2 * TPAD), r = it - y * (2 * TPAD);
        float*    row = &A[(TPAD + y) * PW + TPAD];
        if (r < TPAD)
            row[-1 - r] = row[0];
        else
            row[w + (r - TPAD)] = row[w - 1];
    }
    lds_barrier();
}

__global__ __launch_bounds__(TNT) void k_pyr_tail(TailArgs ka, BatchDesc bd)
{
    __shared__ __attribute__((aligned(16))) float s_a[TPLANE];
    __shared__ __attribute__((aligned(16))) float s_b[TPLANE];
    __shared__ __attribute__((aligned(16))) float s_n[TNEXT];
    /* the arguments in LDS: the taps are indexed by the (run-time) level and tap number */
    __shared__ TailArgs a;
    {
        const int* src = reinterpret_cast<const int*>(&ka);
        int*       dstw = reinterpret_cast<int*>(&a);
        for (int i = threadIdx.x; i < (int)(sizeof(TailArgs) / sizeof(int)); i += TNT) dstw[i] = src[i];
    }
    lds_barrier();
    float* const arena = bd.s[blockIdx.y].arena;
    const int    tid = threadIdx.x;

    for (int o = a.first_oct; o < a.n_oct; o++) {
        const int w = a.w[o], h = a.h[o], pitch = a.pitch[o];
        const int PW = ((w + 3) & ~3) + 2 * TPAD;
        float*    planes = arena + a.data_off[o];
        /* level 0: from HBM for the first octave of the tail (the level L-3 launch of the octave before wrote it), from
         * the LDS copy afterwards */
        if (o == a.first_oct) {
            for (int it = tid; it < w * h; it += TNT) {
                const int y = it / w, x = it - y * w;
                s_a[(TPAD + y) * PW + TPAD + x] = planes[(size_t)y * pitch + x];
            }
        } else {
            for (int it = tid; it < w * h; it += TNT) {
                const int y = it / w, x = it - y * w;
                s_a[(TPAD + y) * PW + TPAD + x] = s_n[it];
            }
        }
        lds_barrier();
        for (int l = 1; l < a.L; l++) {
            tail_pad_cols(s_a, w, h, PW);
            float*     dst = planes + (int64_t)l * a.plane_stride[o];
            const bool want_next = l == a.L - 3 && o + 1 < a.n_oct;
            float*     next0 = want_next ? arena + a.data_off[o + 1] : nullptr;
            const int  next_pitch = want_next ? a.pitch[o + 1] : 0;
            tail_level(s_a, s_b, s_n, want_next, a.g[l], a.halo[l], w, h, PW, dst, pitch, next0, next_pitch);
        }
    }
}

}  // namespace

/* a plane of w x h with its border fits the tail's LDS buffers */
bool pyr_tail_fits(int w, int h)
{
    const long PW = ((w + 3) & ~3) + 2 * TPAD, PH = h + 2 * TPAD;
    /* ... and is small enough for ONE compute unit: a level of w x h pixels costs the workgroup about w * h / 3000 us of
     * vector time (measured: 2.7 us per level at 120 x 68, where three level launches of ~5 us were no slower) */
    return PW * PH <= TPLANE && (long)((w + 1) / 2) * ((h + 1) / 2) <= TNEXT && (long)w * h <= PYR_TAIL_MAX_PX;
}

hipError_t launch_pyr_tail(const TailArgs& a, const BatchDesc& bd, int nb, hipStream_t s)
{
    if (a.first_oct < 1 || a.first_oct >= a.n_oct || a.L < 2 || a.L > PYR_TAIL_MAX_L) return hipErrorInvalidValue;
    for (int l = 1; l < a.L; l++)
        if (a.halo[l] < 1 || a.halo[l] > TPAD) return hipErrorInvalidValue;
    for (int o = a.first_oct; o < a.n_oct; o++)
        if (!pyr_tail_fits(a.w[o], a.h[o])) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pyr_tail, dim3(1, nb), dim3(TNT), 0, s, a, bd);
    return hipGetLastError();
}

}  // namespace popsift_hip
