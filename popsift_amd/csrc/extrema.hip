/*
 * extrema.hip -- 3x3x3 DoG extrema, sub-pixel / edge refinement and wave64
 * ballot compaction, all octaves and levels in ONE launch.
 *
 * Replaces find_extrema_in_dog<HEIGHT,mode> + extrema_count + is_extremum +
 * ModeFunctions<> + solve (s_extrema.cu:22-562, s_solve.h:24-85) and the
 * per-octave host loop Pyramid::find_extrema (s_extrema.cu:565-644).
 *
 * Differences by design: planes are plain HBM rows (coalesced 256 B per wave
 * row, clamp done in software, neighbours served by L1/L2), the compaction is
 * a 64-bit __ballot + one atomicAdd per wave (the reference's 32-bit lane
 * masks are wrong on wave64), and the per-octave counter is clamped to
 * max_extrema by its consumers instead of by a "last block" epilogue.
 * The refinement arithmetic follows the reference expression by expression
 * (compiled with -ffp-contract=off) so that, given identical DoG planes, the
 * accepted set and the (x, y, level) values equal the CPU oracle's bit for bit.
 */
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "sift_types.h"

namespace popsift_hip {
namespace {

constexpr int ETW = 64; /* pixels per block row  */
constexpr int ETH = 16; /* rows per block        */

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct DogView {
    const float* base;
    int64_t      ps;
    int          w, h, pitch, nl; /* nl = number of DoG planes */
    __device__ __forceinline__ float at(int x, int y, int z) const
    {
        x = clampi(x, 0, w - 1);
        y = clampi(y, 0, h - 1);
        z = clampi(z, 0, nl - 1);
        return base[z * ps + (int64_t)y * pitch + x];
    }
    /* caller guarantees 0<=x<w, 0<=y<h, 0<=z<nl */
    __device__ __forceinline__ float raw(int x, int y, int z) const
    {
        return base[z * ps + (int64_t)y * pitch + x];
    }
};

/* s_solve.h:24-85 */
__device__ __forceinline__ bool solve3(float i[3][3], float b[3])
{
    float det0b = -i[1][2] * i[1][2];
    float det0a = i[1][1] * i[2][2];
    float det0 = det0b + det0a;

    float det1b = -i[0][1] * i[2][2];
    float det1a = i[1][2] * i[0][2];
    float det1 = det1b + det1a;

    float det2b = -i[1][1] * i[0][2];
    float det2a = i[0][1] * i[1][2];
    float det2 = det2b + det2a;

    float det3b = -i[0][2] * i[0][2];
    float det3a = i[0][0] * i[2][2];
    float det3 = det3b + det3a;

    float det4b = -i[0][0] * i[1][2];
    float det4a = i[0][1] * i[0][2];
    float det4 = det4b + det4a;

    float det5b = -i[0][1] * i[0][1];
    float det5a = i[0][0] * i[1][1];
    float det5 = det5b + det5a;

    float det;
    det = (i[0][0] * det0);
    det += (i[0][1] * det1);
    det += (i[0][2] * det2);

    if (det == 0) return false;

    const float rsd = 1.0f / det; /* __frcp_rn: correctly rounded reciprocal */

    const float m00 = det0 * rsd, m10 = det1 * rsd, m20 = det2 * rsd;
    const float m11 = det3 * rsd, m12 = det4 * rsd, m22 = det5 * rsd;

    float v0 = 0, v1 = 0, v2 = 0;
    v0 += (m00 * b[0]);
    v0 += (m10 * b[1]);
    v0 += (m20 * b[2]);
    v1 += (m10 * b[0]);
    v1 += (m11 * b[1]);
    v1 += (m12 * b[2]);
    v2 += (m20 * b[0]);
    v2 += (m12 * b[1]);
    v2 += (m22 * b[2]);
    b[0] = v0;
    b[1] = v1;
    b[2] = v2;
    return true;
}

__device__ __forceinline__ int f2i_sat(float f)
{
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* s_extrema.cu:300-504 (after the contrast + 26-neighbour tests) */
template <int MODE>
__device__ bool refine(const DogView& dog, const SiftConsts& sc, int x, int y, int level, float val,
                       int maxlevel, InitExt& ec)
{
    const int width = dog.w, height = dog.h;
    float     Dx = 0, Dy = 0, Dz = 0, DDx = 0, DDy = 0, DXx = 0;
    float     d0 = 0, d1 = 0, d2 = 0;
    int       nx = x, ny = y, nz = level;
    int       iter = 0;
    constexpr int MAX_ITERATIONS = 5;

#define R(dx, dy, dz) dog.at(nx + (dx), ny + (dy), nz + (dz))
    do {
        iter++;
        const float x2y1z1 = R(1, 0, 0), x0y1z1 = R(-1, 0, 0);
        const float x1y2z1 = R(0, 1, 0), x1y0z1 = R(0, -1, 0);
        const float x1y1z2 = R(0, 0, 1), x1y1z0 = R(0, 0, -1);
        Dx = 0.5f * (x2y1z1 - x0y1z1);
        Dy = 0.5f * (x1y2z1 - x1y0z1);
        Dz = 0.5f * (x1y1z2 - x1y1z0);

        const float x1y1z1 = R(0, 0, 0);
        DDx = x2y1z1 + x0y1z1 - 2.0f * x1y1z1;
        DDy = x1y2z1 + x1y0z1 - 2.0f * x1y1z1;
        const float DDz = x1y1z2 + x1y1z0 - 2.0f * x1y1z1;

        const float x0y0z1 = R(-1, -1, 0), x0y1z0 = R(-1, 0, -1), x0y1z2 = R(-1, 0, 1);
        const float x0y2z1 = R(-1, 1, 0), x1y0z0 = R(0, -1, -1), x1y0z2 = R(0, -1, 1);
        const float x1y2z0 = R(0, 1, -1), x1y2z2 = R(0, 1, 1), x2y0z1 = R(1, -1, 0);
        const float x2y1z0 = R(1, 0, -1), x2y1z2 = R(1, 0, 1), x2y2z1 = R(1, 1, 0);
        DXx = 0.25f * (x2y2z1 + x0y0z1 - x0y2z1 - x2y0z1);
        const float DXy = 0.25f * (x2y1z2 + x0y1z0 - x0y1z2 - x2y1z0);
        const float DXz = 0.25f * (x1y2z2 + x1y0z0 - x1y2z0 - x1y0z2);

        float b[3];
        float A[3][3];
        A[0][0] = DDx;
        A[1][1] = DDy;
        A[2][2] = DDz;
        A[1][0] = A[0][1] = DXx;
        A[2][0] = A[0][2] = DXy;
        A[2][1] = A[1][2] = DXz;
        b[0] = -Dx;
        b[1] = -Dy;
        b[2] = -Dz;

        if (!solve3(A, b)) {
            d0 = d1 = d2 = 0;
            break;
        }
        d0 = b[0];
        d1 = b[1];
        d2 = b[2];

        const bool last_it = (iter == MAX_ITERATIONS);
        int        retval;
        if (MODE == POPSIFT_HIP_SIFT_OPENCV) {
            if (fabsf(d0) < 0.5f && fabsf(d1) < 0.5f && fabsf(d2) < 0.5f) {
                retval = 1;
            } else {
                nx = f2i_sat((float)nx + roundf(d0));
                ny = f2i_sat((float)ny + roundf(d1));
                nz = f2i_sat((float)nz + roundf(d2));
                retval = (nx < 5 || nx >= width - 5 || ny < 5 || ny >= height - 5 || nz < 1 ||
                          nz > maxlevel - 2)
                             ? -1
                             : 0;
            }
        } else if (MODE == POPSIFT_HIP_SIFT_VLFEAT) {
            if (last_it) {
                retval = 0;
            } else {
                const float tx = ((d0 >= 0.6f && nx < width - 2) ? 1.0f : 0.0f) +
                                 ((d0 <= -0.6f && nx > 1) ? -1.0f : 0.0f);
                const float ty = ((d1 >= 0.6f && ny < height - 2) ? 1.0f : 0.0f) +
                                 ((d1 <= -0.6f && ny > 1) ? -1.0f : 0.0f);
                if (tx == 0 && ty == 0) {
                    retval = 1;
                } else {
                    nx = (int)((float)nx + tx);
                    ny = (int)((float)ny + ty);
                    retval = 0;
                }
            }
        } else {
            if (last_it) {
                retval = 0;
            } else {
                const int tx = ((d0 >= 0.6f && nx < width - 2) ? 1 : 0) + ((d0 <= -0.6f && nx > 1) ? -1 : 0);
                const int ty = ((d1 >= 0.6f && ny < height - 2) ? 1 : 0) + ((d1 <= -0.6f && ny > 1) ? -1 : 0);
                const int tz =
                    ((d2 >= 0.6f && nz < maxlevel - 1) ? 1 : 0) + ((d2 <= -0.6f && nz > 1) ? -1 : 0);
                if (tx == 0 && ty == 0 && tz == 0) {
                    retval = 1;
                } else {
                    nx += tx;
                    ny += ty;
                    nz += tz;
                    retval = 0;
                }
            }
        }
        if (retval == -1) return false;
        if (retval == 1) break;
    } while (iter < MAX_ITERATIONS);
#undef R

    if (iter >= MAX_ITERATIONS && MODE == POPSIFT_HIP_SIFT_OPENCV) return false;

    if (MODE == POPSIFT_HIP_SIFT_POPSIFT || MODE == POPSIFT_HIP_SIFT_VLFEAT) {
        if (d0 >= 1.5f || d1 >= 1.5f || d2 >= 1.5f) return false;
    }

    const float xn = nx + d0;
    const float yn = ny + d1;
    const float sn = nz + d2;

    if (MODE != POPSIFT_HIP_SIFT_OPENCV) {
        if (xn < 0.0f || xn > width - 1.0f || yn < 0.0f || yn > height - 1.0f || sn < 0.0f || sn > maxlevel)
            return false;
    }

    const float contr = val + 0.5f * (Dx * d0 + Dy * d1 + Dz * d2);
    const float tr = DDx + DDy;
    const float det = DDx * DDy - DXx * DXx;
    const float edgeval = tr * tr / det;

    if (det <= 0.0f) return false;
    if (fabsf(contr) < 2.0f * sc.threshold) return false;
    if (edgeval >= (sc.edge_limit + 1.0f) * (sc.edge_limit + 1.0f) / sc.edge_limit) return false;

    const float wdiv = (float)width / sc.grid_size;
    const float hdiv = (float)height / sc.grid_size;
    ec.xpos = xn;
    ec.ypos = yn;
    ec.lpos = (int)roundf(sn);
    ec.sigma = sc.sigma0 * powf(sc.sigma_k, sn);
    ec.cell = (int)(floorf(yn / hdiv) * sc.grid_size + floorf(xn / wdiv));
    return true;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_extrema(PyrDesc pd, SiftConsts sc, Counters* __restrict__ ct,
                                                 InitExt* __restrict__ iext)
{
    /* block -> (octave, level, tile) */
    int o = 0;
    while (o + 1 < pd.n_oct && (int)blockIdx.x >= pd.o[o + 1].tile_begin) o++;
    const OctDesc& od = pd.o[o];
    const int      tiles_x = (od.w + ETW - 1) / ETW;
    const int      tiles_y = (od.h + ETH - 1) / ETH;
    int            t = blockIdx.x - od.tile_begin;
    const int      level = t / (tiles_x * tiles_y) + 1;
    t -= (level - 1) * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;

    DogView dog;
    dog.base = od.dog;
    dog.ps = od.plane_stride;
    dog.w = od.w;
    dog.h = od.h;
    dog.pitch = od.pitch;
    dog.nl = pd.L - 1;
    const int maxlevel = pd.L - 1; /* s_extrema.cu:608: _levels-1 */

    const int   lane = threadIdx.x & 63;
    const int   x = tx * ETW + lane;
    const float first_thr = (MODE == POPSIFT_HIP_SIFT_OPENCV)   ? floorf(sc.threshold)
                            : (MODE == POPSIFT_HIP_SIFT_VLFEAT) ? 0.8f * 2.0f * sc.threshold
                                                                : 1.6f * sc.threshold;
    InitExt* out = iext + (size_t)o * sc.max_extrema;

    for (int ry = (threadIdx.x >> 6); ry < ETH; ry += 4) {
        const int y = ty * ETH + ry;
        bool      found = false;
        InitExt   ec;
        /* the reference scans x,y >= 1; pixels on the last row/column compare
         * against their own clamped copy and can never be strict extrema */
        bool cand = (x >= 1 && y >= 1 && x <= od.w - 2 && y <= od.h - 2);
        if (MODE == POPSIFT_HIP_SIFT_OPENCV)
            cand = cand && !(x < 5 || y < 5 || x >= od.w - 5 || y >= od.h - 5);
        if (cand) {
            const float val = dog.raw(x, y, level);
            if (fabsf(val) >= first_thr) {
                /* strict max or strict min of the 26 neighbours (s_extrema.cu:56-120) */
                bool gt = true, lt = true;
#pragma unroll
                for (int dz = -1; dz <= 1; dz++)
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            if (dx == 0 && dy == 0 && dz == 0) continue;
                            if (gt || lt) {
                                const float f = dog.raw(x + dx, y + dy, level + dz);
                                gt = gt && (val > f);
                                lt = lt && (val < f);
                            }
                        }
                if (gt || lt) found = refine<MODE>(dog, sc, x, y, level, val, maxlevel, ec);
            }
        }
        /* wave64 compaction (replaces extrema_count, s_extrema.cu:22-44) */
        const unsigned long long mask = __ballot(found);
        if (mask) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ct->ext_ct[o], __popcll(mask));
            base = __shfl(base, 0);
            if (found) {
                const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (idx < sc.max_extrema) out[idx] = ec;
            }
        }
    }
}

}  // namespace

int extrema_tile_w() { return ETW; }
int extrema_tile_h() { return ETH; }

hipError_t launch_extrema(const PyrDesc& pd, const SiftConsts& sc, Counters* ct, InitExt* iext, hipStream_t s)
{
    if (pd.total_tiles <= 0) return hipSuccess;
    const dim3 grid(pd.total_tiles), block(256);
    switch (sc.sift_mode) {
    case POPSIFT_HIP_SIFT_OPENCV:
        hipLaunchKernelGGL((k_extrema<POPSIFT_HIP_SIFT_OPENCV>), grid, block, 0, s, pd, sc, ct, iext);
        break;
    case POPSIFT_HIP_SIFT_VLFEAT:
        hipLaunchKernelGGL((k_extrema<POPSIFT_HIP_SIFT_VLFEAT>), grid, block, 0, s, pd, sc, ct, iext);
        break;
    default:
        hipLaunchKernelGGL((k_extrema<POPSIFT_HIP_SIFT_POPSIFT>), grid, block, 0, s, pd, sc, ct, iext);
        break;
    }
    return hipGetLastError();
}

}  // namespace popsift_hip
