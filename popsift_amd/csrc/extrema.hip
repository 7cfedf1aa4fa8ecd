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
 * a 64-bit __ballot into a per-wave LDS queue + one atomicAdd per strip into one
 * of 64 sub-queues (the reference's 32-bit lane masks are wrong on wave64, and a
 * single hot counter saturates), and the per-octave counter is clamped to
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


__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool FLY>
struct DogView {
    const float* base;  /* stored DoG planes, or (fly) the Gaussian planes: DoG(z) = G(z+1) - G(z), the subtraction
                         * make_dog does (s_pyramid_build.cu:74-92) on the very same f32 values */
    int64_t      ps;
    int          w, h, pitch, nl; /* nl = number of DoG planes */
    __device__ __forceinline__ float raw(int x, int y, int z) const /* caller guarantees 0<=x<w, 0<=y<h, 0<=z<nl */
    {
        const float* q = base + z * ps + (int64_t)y * pitch + x;
        return FLY ? q[ps] - q[0] : q[0];
    }
    __device__ __forceinline__ float at(int x, int y, int z) const
    {
        return raw(clampi(x, 0, w - 1), clampi(y, 0, h - 1), clampi(z, 0, nl - 1));
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

/* slot of neighbour (dx, dy) in refine()'s value cache: centre, +x, -x, +y, -y, then the four diagonals */
__device__ constexpr int nb_index(int dx, int dy)
{
    return (dx == 0 && dy == 0) ? 0 : (dx == 1 && dy == 0) ? 1 : (dx == -1 && dy == 0) ? 2 : (dx == 0 && dy == 1) ? 3
         : (dx == 0 && dy == -1) ? 4 : (dx == -1 && dy == -1) ? 5 : (dx == -1 && dy == 1) ? 6 : (dx == 1 && dy == -1) ? 7 : 8;
}

/* s_extrema.cu:300-504 (after the contrast + 26-neighbour tests) */
template <int MODE, bool FLY>
__device__ bool refine(const DogView<FLY>& dog, const SiftConsts& sc, int x, int y, int level, int maxlevel, InitExt& ec)
{
    const int width = dog.w, height = dog.h;
    float     Dx = 0, Dy = 0, Dz = 0, DDx = 0, DDy = 0, DXx = 0;
    float     d0 = 0, d1 = 0, d2 = 0;
    float     val = 0; /* the candidate's own DoG value: the centre of the FIRST iteration's neighbourhood */
    int       nx = x, ny = y, nz = level;
    int       iter = 0;
    constexpr int MAX_ITERATIONS = 5;

    /* FLY: the 19 DoG values of an iteration come from 28 Gaussian values fetched once (levels nz-1 .. nz+2: the
     * centre cross on the outer two, the full 3x3 on the inner two) instead of 2 x 19 loads; same clamping as
     * DogView::at. */
    float D[3][9];
#define R(dx, dy, dz) (FLY ? D[(dz) + 1][nb_index(dx, dy)] : dog.at(nx + (dx), ny + (dy), nz + (dz)))
    do {
        iter++;
        if (FLY) {
            const int    xs[3] = {clampi(nx - 1, 0, width - 1), clampi(nx, 0, width - 1), clampi(nx + 1, 0, width - 1)};
            const int    ys[3] = {clampi(ny - 1, 0, height - 1), clampi(ny, 0, height - 1), clampi(ny + 1, 0, height - 1)};
            /* nz is 1 .. nl-1 here (the PopSift mode may step up to nl-1, where the z+1 plane clamps back onto
             * plane nl-1 as in DogView::at) */
            const int     zc = clampi(nz, 1, dog.nl - 1);
            const bool    top = (zc + 1 > dog.nl - 1);
            const float*  g = dog.base + (int64_t)(zc - 1) * dog.ps;
            const int64_t ps3 = (top ? 2 : 3) * dog.ps; /* Gaussian level zc+2 does not exist when top */
            constexpr int DX[9] = {0, 1, -1, 0, 0, -1, -1, 1, 1}, DY[9] = {0, 0, 0, 1, -1, -1, 1, -1, 1};
            float         g0[5], g1[9], g2[9], g3[5];
            if (nx >= 1 && nx <= width - 2 && ny >= 1 && ny <= height - 2) {
                /* Every candidate and every step of the refinement stays inside [1, w-2] x [1, h-2] (a pixel on the border
                 * cannot pass the strict 26-neighbour test against its own clamped copy, and the step rules stop at 1 and
                 * w - 2): no coordinate clamps, and the three neighbours of a row are ONE 12-byte load.  The kernel is bound by
                 * the number of scattered requests its lanes put to the L1 (28 dwords per candidate and iteration before:
                 * 36 us for 150 000 candidates), not by their bytes: 12 requests instead of 28. */
                struct f3 { float a, b, c; };
                auto row3 = [](const float* p) -> f3 { f3 v; __builtin_memcpy(&v, p, 12); return v; };
                const float* c0 = g + (int64_t)ny * dog.pitch + (nx - 1);
                const float* c1 = c0 + dog.ps;
                const float* c2 = c1 + dog.ps;
                const float* c3 = g + ps3 + (int64_t)ny * dog.pitch + (nx - 1);
                const int    pt = dog.pitch;
                const f3 a0 = row3(c0), a3 = row3(c3);
                const f3 m1 = row3(c1 - pt), z1 = row3(c1), p1 = row3(c1 + pt);
                const f3 m2 = row3(c2 - pt), z2 = row3(c2), p2 = row3(c2 + pt);
                g0[2] = a0.a, g0[0] = a0.b, g0[1] = a0.c, g0[3] = c0[pt + 1], g0[4] = c0[1 - pt];
                g3[2] = a3.a, g3[0] = a3.b, g3[1] = a3.c, g3[3] = c3[pt + 1], g3[4] = c3[1 - pt];
                g1[2] = z1.a, g1[0] = z1.b, g1[1] = z1.c, g1[6] = p1.a, g1[3] = p1.b, g1[8] = p1.c, g1[5] = m1.a, g1[4] = m1.b, g1[7] = m1.c;
                g2[2] = z2.a, g2[0] = z2.b, g2[1] = z2.c, g2[6] = p2.a, g2[3] = p2.b, g2[8] = p2.c, g2[5] = m2.a, g2[4] = m2.b, g2[7] = m2.c;
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const int64_t off = (int64_t)ys[DY[i] + 1] * dog.pitch + xs[DX[i] + 1];
                    if (i < 5) g0[i] = g[off];
                    g1[i] = g[dog.ps + off];
                    g2[i] = g[2 * dog.ps + off];
                    if (i < 5) g3[i] = g[ps3 + off];
                }
            }
#pragma unroll
            for (int i = 0; i < 9; i++) {
                D[1][i] = g2[i] - g1[i];
                if (i < 5) {
                    D[0][i] = g1[i] - g0[i];
                    D[2][i] = top ? D[1][i] : g3[i] - g2[i];
                }
            }
        }
        const float x2y1z1 = R(1, 0, 0), x0y1z1 = R(-1, 0, 0);
        const float x1y2z1 = R(0, 1, 0), x1y0z1 = R(0, -1, 0);
        const float x1y1z2 = R(0, 0, 1), x1y1z0 = R(0, 0, -1);
        Dx = 0.5f * (x2y1z1 - x0y1z1);
        Dy = 0.5f * (x1y2z1 - x1y0z1);
        Dz = 0.5f * (x1y1z2 - x1y1z0);

        const float x1y1z1 = R(0, 0, 0);
        if (iter == 1) val = x1y1z1;
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

/*
 * Detection: one wave owns a strip of 64 candidate columns and marches down DET_RH rows.  Per plane a row
 * is three overlapping row loads (left neighbour, self, right neighbour: the same cache lines), reduced to
 * max3/min3; three consecutive rows of these give the 3x3 neighbourhood extremes of the planes above and
 * below, and (with the centre excluded) of the own plane -- the strict 26-neighbour test of
 * s_extrema.cu:56-120 without any divergent load.  (A first version shifted lanes with wave-wide DPP
 * instead of loading the neighbours; those shifts stalled the wave for tens of cycles each on gfx950.)
 */
__device__ __forceinline__ int xcd_run(int b, int n)
{
    const int q = n >> 3, r = n & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

constexpr int DET_W = 64;
constexpr int DET_RH = 32;

template <int NP>
struct RowRed {
    float mx[NP], mn[NP];
};

/*
 * LEVELS = DoG search levels; NP = LEVELS + 2 DoG planes.
 *
 * Memory pipeline.  A strip is walked in groups of DET_G rows: all DET_G * NP row loads of a group
 * are issued together and consumed one row at a time (counted vmcnt waits), and NOTHING else
 * touches the vector-memory queue inside a group.  An earlier version flushed the candidate queue
 * to global memory from inside the row loop; the mere presence of that conditional store path made
 * the compiler drain the queue (s_waitcnt vmcnt(0)) at every row, i.e. one exposed memory latency
 * per row (73 % of the wave cycles were waits).  Now candidates only go to the wave's LDS queue in
 * the loop and are flushed once, after the strip.  A strip with more than DET_Q candidates (never
 * seen outside synthetic stress images; typical is ~50) is handed to the SLOW instantiation of this
 * kernel, which re-does it with a flush per row.
 */
constexpr int DET_G = 4;   /* rows per load group (2 and 8, and a double-buffered variant, measured no better) */
constexpr int DET_Q = 512; /* per-wave candidate queue (entries) */

template <int MODE, int LEVELS, bool SLOW, bool FLY>
__global__ __launch_bounds__(256) void k_detect(const PyrDesc* __restrict__ pdp, BatchDesc bd, SiftConsts sc, int cand_cap)
{
    /* this image's planes and lists (the slot of blockIdx.y) */
    const float* __restrict__ arena = bd.s[blockIdx.y].arena;
    Counters* __restrict__    ct = bd.s[blockIdx.y].ct;
    int2* __restrict__        cand = bd.s[blockIdx.y].cand;
    int* __restrict__         ovf = bd.s[blockIdx.y].ovf;
    constexpr int   NP = LEVELS + 2;
    constexpr int   QCAP = SLOW ? ((128 * LEVELS > DET_Q) ? 128 * LEVELS : DET_Q) : DET_Q;
    __shared__ int2 s_queue[4][QCAP];
    const int       lane = threadIdx.x & 63;
    int2*           queue = s_queue[threadIdx.x >> 6];
    const int       n_oct = pdp->n_oct;
    const int       n_units = SLOW ? min(ct->pad[1], pdp->total_tiles) : pdp->total_tiles;
    const float     first_thr = (MODE == POPSIFT_HIP_SIFT_OPENCV)   ? floorf(sc.threshold)
                                : (MODE == POPSIFT_HIP_SIFT_VLFEAT) ? 0.8f * 2.0f * sc.threshold
                                                                    : 1.6f * sc.threshold;

    /* Workgroups b and b+8 share an XCD and its L2: hand each XCD a contiguous run of units (a horizontal band of
     * the plane), so the cache lines that neighbouring strips share -- the column to the left and right of a
     * workgroup's 256 columns, the row above and below its 32 rows -- are fetched from HBM once instead of
     * once per XCD. */
    const int wg = SLOW ? (int)blockIdx.x : xcd_run(blockIdx.x, gridDim.x);
    for (int ui = wg * 4 + (threadIdx.x >> 6); ui < n_units; ui += gridDim.x * 4) {
        const int unit = SLOW ? ovf[ui] : ui;
        int       o = 0;
        while (o + 1 < n_oct && unit >= pdp->o[o + 1].tile_begin) o++;
        const OctDesc od = pdp->o[o];
        const int     w = od.w, h = od.h;
        const int     strips = (w - 2) / DET_W + 1; /* lanes cover x = 0 .. strips*64-1, candidates are 1 .. w-2 */
        const int     u = unit - od.tile_begin;
        const int     cy = u / strips, sx = u - cy * strips;
        const int     x = sx * DET_W + lane;
        const int     xc = min(x, w - 1);
        const int     yb = 1 + cy * DET_RH;
        const int     ye = min(yb + DET_RH - 1, h - 2);
        if (strips <= 0 || yb > ye) continue;

        bool lane_ok = (x >= 1 && x <= w - 2);
        if (MODE == POPSIFT_HIP_SIFT_OPENCV) lane_ok = lane_ok && (x >= 5 && x < w - 5);

        int       n_buf = 0;
        bool      overflow = false;
        const int qcap = SLOW ? QCAP : min(sc.det_qcap, QCAP);
        auto flush = [&](int n) -> int {
            if (n == 0) return 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            /* the sub-queue is the strip's cell of an 8 x 8 grid over the octave (sift_types.h, Counters::qcnt) */
            const int urows = (h - 2 + DET_RH - 1) / DET_RH;
            const int subq = (cy * 8 / urows) * 8 + (sx * 8 / strips), seg = cand_cap / DET_SUBQ;
            int       basei = 0;
            if (lane == 0) basei = atomicAdd(&ct->qcnt[subq].n, n);
            basei = __shfl(basei, 0);
            for (int i = lane; i < n; i += 64)
                if (basei + i < seg) cand[(size_t)subq * seg + basei + i] = queue[i];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return 0;
        };

        /* left / centre / right neighbours are three overlapping row loads (same cache lines, the
         * texture path sorts it out) rather than lane shifts: wave-wide DPP shifts turned out to
         * stall the wave for tens of cycles each on gfx950 */
        const float* base = arena + (FLY ? od.data_off : od.dog_off);
        constexpr int NRAW = FLY ? 3 * (NP + 1) : 3 * NP; /* FLY: NP + 1 Gaussian planes, subtracted when a row is consumed */
        const int    xl = max(x - 1, 0), xr = min(x + 1, w - 1);
        RowRed<NP>   A, B, C;
        float        vB[NP], smx[NP], smn[NP];

        auto fetch_row = [&](int y, float* raw) {
            const float* p = base + (int64_t)y * od.pitch;
#pragma unroll
            for (int z = 0; z < NRAW / 3; z++) {
                raw[3 * z + 0] = p[z * od.plane_stride + xc];
                raw[3 * z + 1] = p[z * od.plane_stride + xl];
                raw[3 * z + 2] = p[z * od.plane_stride + xr];
            }
        };
        auto reduce_row = [&](const float* raw, RowRed<NP>& R, float* v, float* sx_, float* sn_) {
#pragma unroll
            for (int z = 0; z < NP; z++) {
                const float c = FLY ? raw[3 * z + 3] - raw[3 * z + 0] : raw[3 * z + 0];
                const float l = FLY ? raw[3 * z + 4] - raw[3 * z + 1] : raw[3 * z + 1];
                const float r = FLY ? raw[3 * z + 5] - raw[3 * z + 2] : raw[3 * z + 2];
                v[z] = c;
                sx_[z] = fmaxf(l, r);
                sn_[z] = fminf(l, r);
                R.mx[z] = fmaxf(sx_[z], c);
                R.mn[z] = fminf(sn_[z], c);
            }
        };
        {
            float tv[NP], ts[NP], tn[NP], ra[NRAW], rb[NRAW];
            fetch_row(yb - 1, ra);
            fetch_row(yb, rb);
            reduce_row(ra, A, tv, ts, tn);
            reduce_row(rb, B, vB, smx, smn);
        }
        /* one row: q holds row y+1 */
        auto step = [&](int y, const float* q) {
            float vC[NP], cmx[NP], cmn[NP];
            reduce_row(q, C, vC, cmx, cmn);
            bool row_ok = lane_ok && (y <= ye);
            if (MODE == POPSIFT_HIP_SIFT_OPENCV) row_ok = row_ok && (y >= 5 && y < h - 5);
            /* full 3x3 extremes of every plane (centre column included) */
            float fmx[NP], fmn[NP];
#pragma unroll
            for (int z = 0; z < NP; z++) {
                fmx[z] = fmaxf(fmaxf(A.mx[z], B.mx[z]), C.mx[z]);
                fmn[z] = fminf(fminf(A.mn[z], B.mn[z]), C.mn[z]);
            }
#pragma unroll
            for (int z = 1; z <= LEVELS; z++) {
                const float v = vB[z];
                const float omx = fmaxf(fmaxf(A.mx[z], smx[z]), C.mx[z]); /* own plane, centre excluded */
                const float omn = fminf(fminf(A.mn[z], smn[z]), C.mn[z]);
                const float nmax = fmaxf(fmaxf(fmx[z - 1], fmx[z + 1]), omx);
                const float nmin = fminf(fminf(fmn[z - 1], fmn[z + 1]), omn);
                const bool  hit = row_ok && (fabsf(v) >= first_thr) && (v > nmax || v < nmin);
                const unsigned long long mask = __ballot(hit);
                if (mask) {
                    /* stage in the wave's LDS queue; n_buf is wave-uniform */
                    const int cnt = __popcll(mask);
                    if (n_buf + cnt <= qcap) {
                        if (hit)
                            queue[n_buf + __popcll(mask & ((1ull << lane) - 1ull))] =
                                make_int2(x | (y << 16), z | (o << 8));
                        n_buf += cnt;
                    } else {
                        overflow = true;
                    }
                }
            }
            A = B;
            B = C;
#pragma unroll
            for (int z = 0; z < NP; z++) {
                vB[z] = vC[z];
                smx[z] = cmx[z];
                smn[z] = cmn[z];
            }
        };
        if (SLOW) {
            for (int y = yb; y <= ye; y++) {
                float q[NRAW];
                fetch_row(y + 1, q);
                step(y, q);
                if (n_buf > QCAP - 64 * LEVELS) n_buf = flush(n_buf);
            }
            flush(n_buf);
        } else {
            for (int y0 = yb; y0 <= ye; y0 += DET_G) {
                float q[DET_G][NRAW];
#pragma unroll
                for (int k = 0; k < DET_G; k++) fetch_row(min(y0 + k + 1, ye + 1), q[k]);
#pragma unroll
                for (int k = 0; k < DET_G; k++) step(y0 + k, q[k]);
            }
            if (overflow) {
                /* too many candidates for the queue: leave the whole strip to the SLOW pass */
                if (lane == 0) ovf[atomicAdd(&ct->pad[1], 1)] = unit;
            } else {
                flush(n_buf);
            }
        }
    }
}

/*
 * Refinement of the candidates: one lane per candidate in dense waves.  Survivors are appended
 * to their octave's list with ONE returning global atomic per workgroup, octave and step (LDS counters
 * gather the four waves first) -- the same hot-counter limit as in the detection kernel.
 */
template <int MODE, bool FLY>
__global__ __launch_bounds__(256) void k_refine(const PyrDesc* __restrict__ pdp, BatchDesc bd, SiftConsts sc, int cand_cap)
{
    const float* __restrict__ arena = bd.s[blockIdx.y].arena;
    Counters* __restrict__    ct = bd.s[blockIdx.y].ct;
    const int2* __restrict__  cand = bd.s[blockIdx.y].cand;
    InitExt* __restrict__     iext = bd.s[blockIdx.y].iext;
    __shared__ int s_cnt[PS_MAX_OCT], s_base[PS_MAX_OCT];
    __shared__ int s_pref[DET_SUBQ + 1]; /* 256-candidate steps before each sub-queue */
    __shared__ int s_tot[DET_SUBQ];      /* candidates in each sub-queue */
    /* The kernel is ONE step per workgroup (1024 workgroups for the ~590 steps of a 1080p image): its time is the
     * chain of dependent memory round trips of that step, ~1-2 us each.  So everything a candidate needs besides the
     * planes is in LDS before its record arrives -- the sub-queue sizes and the octaves' geometry -- and its own DoG
     * value is taken from the first iteration's loads (refine()) instead of a round trip of its own: cand -> planes ->
     * solve where it was qcnt -> cand -> octave record -> value -> planes -> solve. */
    __shared__ int64_t s_off[PS_MAX_OCT], s_ps[PS_MAX_OCT];
    __shared__ int     s_w[PS_MAX_OCT], s_h[PS_MAX_OCT], s_pitch[PS_MAX_OCT];
    const int      lane = threadIdx.x & 63;
    const int      L = pdp->L, n_oct = pdp->n_oct;
    const int      seg = cand_cap / DET_SUBQ;
    if (threadIdx.x < 64) {
        const int tot = min(ct->qcnt[lane].n, seg);
        const int steps = (tot + 255) >> 8;
        int       incl = steps;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int v = __shfl_up(incl, s);
            if (lane >= s) incl += v;
        }
        s_pref[lane + 1] = incl;
        s_tot[lane] = tot;
        if (lane == 0) s_pref[0] = 0;
    } else if ((int)threadIdx.x - 64 < n_oct) {
        const int      o = (int)threadIdx.x - 64;
        const OctDesc* od = &pdp->o[o];
        s_off[o] = FLY ? od->data_off : od->dog_off;
        s_ps[o] = od->plane_stride;
        s_w[o] = od->w;
        s_h[o] = od->h;
        s_pitch[o] = od->pitch;
    }
    __syncthreads();
    const int n_steps = s_pref[DET_SUBQ];
    /* the steps of all sub-queues form one list that the workgroups share out; trip counts are
     * workgroup-uniform, so the barriers below are reached by all four waves */
    {
        for (int w = blockIdx.x; w < n_steps; w += gridDim.x) {
            /* the sub-queue of step w: the number of sub-queues that end at or before it (one LDS read and a ballot
             * instead of a walk over up to 64 dependent reads) */
            const int q = __popcll(__ballot(s_pref[lane + 1] <= w));
            const int total = s_tot[q];
            const int b0 = (w - s_pref[q]) << 8;
            if (threadIdx.x < PS_MAX_OCT) s_cnt[threadIdx.x] = 0;
            __syncthreads();
            const int i = b0 + threadIdx.x;
            bool      found = false;
            InitExt   ec;
            int       o = 0, slot = 0;
            if (i < total) {
                const int2 cd = cand[(size_t)q * seg + i];
                const int  x = cd.x & 0xffff, y = cd.x >> 16, level = cd.y & 0xff;
                o = cd.y >> 8;
                DogView<FLY>   dog;
                dog.base = arena + s_off[o];
                dog.ps = s_ps[o];
                dog.w = s_w[o];
                dog.h = s_h[o];
                dog.pitch = s_pitch[o];
                dog.nl = L - 1;
                found = refine<MODE, FLY>(dog, sc, x, y, level, L - 1, ec);
            }
            /* wave64 compaction per octave (replaces extrema_count, s_extrema.cu:22-44), wave -> workgroup in LDS */
            unsigned long long todo = __ballot(found);
            while (todo) {
                const int                leader = __ffsll((long long)todo) - 1;
                const int                lo = __shfl(o, leader);
                const unsigned long long mask = __ballot(found && o == lo);
                int                      wbase = 0;
                if (lane == leader) wbase = atomicAdd(&s_cnt[lo], __popcll(mask));
                wbase = __shfl(wbase, leader);
                if (found && o == lo) slot = wbase + __popcll(mask & ((1ull << lane) - 1ull));
                todo &= ~mask;
            }
            __syncthreads();
            if ((int)threadIdx.x < n_oct && s_cnt[threadIdx.x] > 0)
                s_base[threadIdx.x] = atomicAdd(&ct->ext_ct[threadIdx.x], s_cnt[threadIdx.x]);
            __syncthreads();
            if (found) {
                const int idx = s_base[o] + slot;
                if (idx < sc.max_extrema) iext[(size_t)o * sc.max_extrema + idx] = ec;
            }
            /* s_cnt is reset after the next barrier at the top of the loop; s_base is only read by lanes that found */
        }
    }
}

}  // namespace

int extrema_units(int w, int h)
{
    if (w < 3 || h < 3) return 0;
    return ((w - 2) / DET_W + 1) * ((h - 2 + DET_RH - 1) / DET_RH);
}

template <int MODE>
static void launch_detect(const PyrDesc& pd, const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc,
                          int cand_cap, hipStream_t s)
{
    const dim3 grid((pd.total_tiles + 3) / 4, nb), block(256), sgrid(64, nb);
    switch (pd.levels) {
#define PS_LV(N)                                                                                              \
    case N:                                                                                                   \
        if (pd.dog_fly) {                                                                                     \
            hipLaunchKernelGGL((k_detect<MODE, N, false, true>), grid, block, 0, s, d_pd, bd, sc, cand_cap);  \
            hipLaunchKernelGGL((k_detect<MODE, N, true, true>), sgrid, block, 0, s, d_pd, bd, sc, cand_cap);  \
        } else {                                                                                              \
            hipLaunchKernelGGL((k_detect<MODE, N, false, false>), grid, block, 0, s, d_pd, bd, sc, cand_cap);  \
            hipLaunchKernelGGL((k_detect<MODE, N, true, false>), sgrid, block, 0, s, d_pd, bd, sc, cand_cap);  \
        }                                                                                                     \
        break;
        PS_LV(2) PS_LV(3) PS_LV(4) PS_LV(5) PS_LV(6) PS_LV(7) PS_LV(8) PS_LV(9)
#undef PS_LV
    }
}

hipError_t launch_extrema(const PyrDesc& pd, const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc, int cand_cap,
                          bool /*filtered*/, hipStream_t s, hipEvent_t mid)
{
    if (pd.total_tiles <= 0) {
        if (mid) (void)hipEventRecord(mid, s); /* the stage boundary exists even when there is nothing to detect */
        return hipSuccess;
    }
    if (pd.levels < 2 || pd.levels > 9) return hipErrorInvalidValue;
    const dim3 block(256), rgrid(1024, nb);
    switch (sc.sift_mode) {
    case POPSIFT_HIP_SIFT_OPENCV:
        launch_detect<POPSIFT_HIP_SIFT_OPENCV>(pd, d_pd, bd, nb, sc, cand_cap, s);
        if (mid) (void)hipEventRecord(mid, s);
        if (pd.dog_fly)
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_OPENCV, true>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        else
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_OPENCV, false>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        break;
    case POPSIFT_HIP_SIFT_VLFEAT:
        launch_detect<POPSIFT_HIP_SIFT_VLFEAT>(pd, d_pd, bd, nb, sc, cand_cap, s);
        if (mid) (void)hipEventRecord(mid, s);
        if (pd.dog_fly)
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_VLFEAT, true>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        else
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_VLFEAT, false>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        break;
    default:
        launch_detect<POPSIFT_HIP_SIFT_POPSIFT>(pd, d_pd, bd, nb, sc, cand_cap, s);
        if (mid) (void)hipEventRecord(mid, s);
        if (pd.dog_fly)
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_POPSIFT, true>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        else
            hipLaunchKernelGGL((k_refine<POPSIFT_HIP_SIFT_POPSIFT, false>), rgrid, block, 0, s, d_pd, bd, sc, cand_cap);
        break;
    }
    return hipGetLastError();
}

}  // namespace popsift_hip
