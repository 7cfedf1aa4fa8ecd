/*
 * keypoint.hip -- orientation assignment, orientation prefix sum, the 128-D descriptors
 * (loop, grid, notile / igrid, iloop) + normalisation, Feature assembly.  All kernels size themselves
 * from the device-resident counters (grid-stride), so the host never has to
 * read a count back between stages (the reference blocks on the counters twice
 * per image: s_orientation.cu:351, sift_desc.cu:60-61).
 *
 * Replaces:
 *   ori_par               s_orientation.cu:60-242   -> k_orientation (1 wave / extremum)
 *   ori_prefix_sum        s_orientation.cu:303-345  -> k_scan_local + k_scan_apply (256 extrema / workgroup)
 *   ext_desc_loop(+_sub)  s_desc_loop.cu:19-161     -> k_descriptor (1 wave / descriptor)
 *   ext_desc_grid         s_desc_grid.cu:19-147     -> k_descriptor_grid
 *   ext_desc_notile/igrid s_desc_notile.cu:28-166, s_desc_igrid.cu:20-109 -> k_descriptor_notile<false>
 *   ext_desc_iloop        s_desc_iloop.cu:18-154    -> k_descriptor_notile<true>
 *   normalize_histogram   s_desc_normalize.h:14-33, s_desc_norm_rs.h, s_desc_norm_l2.h
 *                                                   -> fused into the descriptor kernels
 *   prep_features         sift_pyramid.cu:249-279   -> fused into k_scan_apply
 */
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "sift_types.h"

namespace popsift_hip {
namespace {

/* sift_constants.h:22-29: float constants */
constexpr float F_PI = 3.14159265358979323846f;
constexpr float F_PI2 = 2.0f * 3.14159265358979323846f;
constexpr float            ORI_WINFACTOR = 1.5f;
constexpr float            DESC_MAGNIFY = 3.0f;

/*
 * Histogram accumulation: on gfx950 an LDS float atomic add (ds_add_f32) costs
 * ~200 cycles per wave instruction whatever the address pattern, an integer one
 * ~6 (tools/ubench/lds_atomic.hip).  Bins are therefore 64-bit fixed point
 * (2^-32 resolution, exact and order-independent sums => bit-reproducible
 * histograms, unlike the reference's float atomicAdd, s_orientation.cu:136).
 */
typedef unsigned long long fix64;
/* one sample weighs < 2^11 (|gradient| <= 255*2*sqrt2, window weights <= 1): w * 2^20 fits 32 bits */
__device__ __forceinline__ fix64 to_fix(float w) { return (fix64)(unsigned int)(w * 1048576.0f); }
__device__ __forceinline__ float from_fix(fix64 v) { return (float)((double)v * (1.0 / 1048576.0)); }


__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* Inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15
 * and row_bcast:31 carry the row totals across), register to register -- no LDS round trips as with six shuffles. */
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); /* row_shr:1, out-of-row lanes read 0 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
    return v;
}


/* The sum over the 64 lanes, in every lane: four DPP adds inside the rows of 16 (quad butterflies, then the two mirrors),
 * the four row sums through scalar registers.  Register to register -- six __shfl_xor are six dependent ds_bpermute round
 * trips through the LDS queue (~150 cycles each behind whatever the wave has queued there), three times per descriptor
 * with the L2 normalisation.  A fixed order of additions like the butterfly's, so results stay reproducible. */
__device__ __forceinline__ float wave_allsum(float v)
{
#define PS_DPP_ADD(CTRL) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false))
    PS_DPP_ADD(0xB1);  /* quad_perm [1,0,3,2] */
    PS_DPP_ADD(0x4E);  /* quad_perm [2,3,0,1] */
    PS_DPP_ADD(0x141); /* row_half_mirror     */
    PS_DPP_ADD(0x140); /* row_mirror          */
#undef PS_DPP_ADD
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

/* a value every lane of the wave holds identically, moved to scalar registers */
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ long long uniform64(long long v)
{
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)(v & 0xffffffffll));
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

/* The four gradient taps around plane element `off` (>= pitch + 1, so every tap is in the plane).  `layer` and
 * `pitch` are wave-uniform: with an UNSIGNED 32-bit byte offset the loads take the scalar-base + vector-offset
 * form (three scalar bases: the row, the row below, the row above) instead of 64-bit vector address arithmetic
 * per tap.  A plane is far below 4 GiB (the arena's largest is 7680 x 4320 floats). */
/* ISA: taps */
struct Taps {
    const char* row;
    const char* dn;
    const char* up;
    __device__ __forceinline__ Taps(const float* layer, int pitch)
        : row((const char*)layer), dn((const char*)(layer + pitch)), up((const char*)(layer - pitch)) {}
    /* the same with the BYTE offset of the element */
    __device__ __forceinline__ void load_b(unsigned int b, float& xp, float& xm, float& yp, float& ym) const
    {
        xp = *(const float*)(row + (size_t)b + 4);
        xm = *(const float*)(row + (size_t)b - 4);
        yp = *(const float*)(dn + (size_t)b);
        ym = *(const float*)(up + (size_t)b);
    }
    __device__ __forceinline__ void load(int off, float& xp, float& xm, float& yp, float& ym) const
    {
        const size_t b = (unsigned int)off * 4u;
        xp = *(const float*)(row + b + 4);
        xm = *(const float*)(row + b - 4);
        yp = *(const float*)(dn + b);
        ym = *(const float*)(up + b);
    }
};

/* ISA: end */
/* element `idx` >= 0 of a wave-uniform plane through an unsigned 32-bit byte offset (scalar base + vector offset) */
__device__ __forceinline__ float plane_at(const float* pl, int idx)
{
    return *(const float*)((const char*)pl + (size_t)((unsigned int)idx * 4u));
}

/*
 * Work distribution of the keypoint kernels.  Workgroups b and b + 8 share an XCD and its 4 MiB L2 (round-robin
 * dispatch -- relied on for speed only, never for correctness).  The lists come out of refinement grouped by image
 * region (extrema.hip: a detection sub-queue is a region, refinement appends one sub-queue batch at a time), so they
 * are cut into chunks of KP_CHUNK consecutive keypoints, chunk c goes to XCD c mod 8, and the workgroups of an XCD
 * walk its chunks front to back: the patches of the waves in flight on one XCD overlap, while every XCD still gets
 * the same mix of levels and octaves (the work per keypoint grows with sigma^2; handing each XCD one contiguous
 * eighth of a level-major list left the XCD with the coarse levels running 35 % longer than the others).
 * gridDim.x is a multiple of 8; NW = waves per workgroup.  Position s of XCD x is list element
 * ((s / KP_CHUNK) * 8 + x) * KP_CHUNK + s % KP_CHUNK.
 * (Tried and dropped, round 2: resident waves pulling positions from work queues in device memory -- one returning
 * atomicAdd per keypoint on 32 padded heads, next position requested ahead of time.  592 us against 505 us for the
 * descriptor kernel, 164 against 106 for orientation: the hardware dispatcher already balances 65536 single-wave
 * workgroups, and it does so without a memory round trip in the wave's in-order load queue.  Round 3 repeated it with
 * FEWER resident waves -- 8 .. 32 per CU, one counter per XCD, one descriptor / four histograms per fetch -- so that wave
 * slots, registers and LDS stay free for the other images' bandwidth-bound kernels: at 32 per CU it equals this
 * distribution (k_descriptor 421 us, 2.90 Gpix/s), at 24 / 16 / 12 / 8 it loses (2.89 / 2.77 / 2.58 / 2.47 Gpix/s), and
 * the level launches of the other images take as long in the mix as before (107 us against 108 us per octave-0 level,
 * profiles/r03_timeline_resident16.txt): what slows them beside a descriptor kernel is the shared vector issue, LDS and
 * L2, not the lack of a free slot.  DESIGN 6.4.)
 */
#ifndef KP_CHUNK
#define KP_CHUNK 256
#endif
struct XcdSlice {
    int x, s, step;
    __device__ __forceinline__ int index() const { return ((s / KP_CHUNK) * 8 + x) * KP_CHUNK + (s % KP_CHUNK); }
    /* positions whose element lies beyond the list are skipped: later chunks of the same XCD may still be inside */
    __device__ __forceinline__ bool more(int total) const { return (s / KP_CHUNK) * 8 * KP_CHUNK < total; }
};
template <int NW>
__device__ __forceinline__ XcdSlice xcd_slice()
{
    XcdSlice sl;
    sl.x = blockIdx.x & 7;
    sl.s = (int)(blockIdx.x >> 3) * NW + (int)(threadIdx.x >> 6);
    sl.step = (int)(gridDim.x >> 3) * NW;
    return sl;
}

/* clamped extrema counts -> exclusive prefix (uniform, <= 20 entries) */
__device__ __forceinline__ int ext_prefix(const Counters* ct, const SiftConsts& sc, int n_oct, int* ps)
{
    int acc = 0;
    for (int o = 0; o < n_oct; o++) {
        ps[o] = acc;
        acc += min(ct->ext_ct[o], sc.max_extrema);
    }
    ps[n_oct] = acc;
    return acc;
}

/* s_gradiant.h:55-69; the caller keeps (x, y) inside [1,w-2]x[1,h-2] */
__device__ __forceinline__ void get_gradiant(float& grad, float& theta, int x, int y, const float* pl, int pitch)
{
    const float* c = pl + (size_t)y * pitch + x;
    const float  dx = c[1] - c[-1];
    const float  dy = c[pitch] - c[-pitch];
    grad = hypotf(dx, dy);
    theta = atan2f(dy, dx);
}

/* ------------------------------------------------------------ orientation */

/* atan2 for the HARD 36-bin orientation histogram (s_orientation.cu:129): needs libm-grade
 * accuracy, because an error of e rad moves a sample across a bin edge with probability
 * e / 0.1745.  Cephes atanf scheme (|error| ~ 1.5e-7 rad, about 1 ulp at pi/2 -- the level at
 * which device and host libm already disagree) with Newton-refined v_rcp quotients instead of
 * the ~12-instruction IEEE division sequence of the library routine. */
__device__ __forceinline__ float div_nr(float a, float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ float atan2_acc(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float       a = (mx == 0.0f) ? 0.0f : div_nr(mn, mx);
    const bool  big = a > 0.4142135623730950f; /* tan(pi/8) */
    const float t = big ? div_nr(a - 1.0f, a + 1.0f) : a;
    const float z = t * t;
    const float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z,
                         -3.33329491539e-1f);
    float       r = fmaf(p * z, t, t);
    r = big ? r + 0.7853981633974483f : r;
    r = (ay > ax) ? 1.5707963267948966f - r : r;
    r = (x < 0.0f) ? 3.14159265358979323846f - r : r;
    return (y < 0.0f) ? -r : r;
}

/*
 * cos / sin of the keypoint orientation, evaluated in double and rounded once: the same two floats as the oracle's
 * sincos_cr (correctly rounded up to a ~1e-8 chance of double rounding).  |ang| <= pi (s_orientation.cu:222), so a
 * quadrant reduction with pi/2 split in two doubles and Taylor polynomials on [-pi/4, pi/4] (truncation < 1e-19,
 * rounding ~3e-16 relative) do it in ~35 double operations -- the library's double sincos with its large-argument
 * path is far bigger.  Done once per descriptor in k_scan_apply and stored next to the descriptor -> extremum map: inside the descriptor kernels
 * (64-register budget) any double arithmetic spilled and cost 5-9 %, inside k_orientation it raised the register
 * count from 62 to 86.
 */
__device__ __forceinline__ void sincos_cr(float ang, float& s, float& c)
{
    const double x = (double)ang;
    const double q = rint(x * 0.63661977236758134308);             /* 2/pi */
    double       r = fma(-q, 1.57079632679489655800e+00, x);       /* pi/2, high part */
    r = fma(-q, 6.12323399573676603587e-17, r);                     /* pi/2, low part  */
    const double z = r * r;
    double       ps = 2.81145725434552075980e-15;                    /* 1/17! */
    ps = fma(ps, z, -7.64716373181981647590e-13);                    /* 1/15! */
    ps = fma(ps, z, 1.60590438368216145994e-10);                     /* 1/13! */
    ps = fma(ps, z, -2.50521083854417187751e-08);                    /* 1/11! */
    ps = fma(ps, z, 2.75573192239858906526e-06);                     /* 1/9!  */
    ps = fma(ps, z, -1.98412698412698412698e-04);                    /* 1/7!  */
    ps = fma(ps, z, 8.33333333333333333333e-03);                     /* 1/5!  */
    ps = fma(ps, z, -1.66666666666666666667e-01);                    /* 1/3!  */
    const double sr = fma(ps * z, r, r);
    double       pc = -1.56192069685862264622e-16;                   /* 1/18! */
    pc = fma(pc, z, 4.77947733238738529744e-14);                     /* 1/16! */
    pc = fma(pc, z, -1.14707455977297247139e-11);                    /* 1/14! */
    pc = fma(pc, z, 2.08767569878680989792e-09);                     /* 1/12! */
    pc = fma(pc, z, -2.75573192239858906526e-07);                    /* 1/10! */
    pc = fma(pc, z, 2.48015873015873015873e-05);                     /* 1/8!  */
    pc = fma(pc, z, -1.38888888888888888889e-03);                    /* 1/6!  */
    pc = fma(pc, z, 4.16666666666666666667e-02);                     /* 1/4!  */
    pc = fma(pc, z, -0.5);
    const double cr = fma(pc, z, 1.0);
    const int    n = (int)q & 3; /* quadrant: sin/cos of r + n pi/2 */
    const double sd = (n & 1) ? cr : sr, cd = (n & 1) ? sr : cr;
    s = (float)((n & 2) ? -sd : sd);
    c = (float)(((n + 1) & 2) ? -cd : cd);
}

/* ISA: magnitude + angle */
/* t = atan(min / max) * 4/pi in [0, 1] -> the angle of (x, y) in (-4, 4], without compares and selects (half rate on
 * gfx950; and / xor / add / sub are full rate, tools/ubench/valu_rate.hip):
 *   2 - sx * (1 + sd * (1 - t)),   sd = -1 where |y| > |x|, sx = -1 where x < 0
 * (2 - t where the roles of x and y were swapped, then 4 - that where x < 0), each sign applied by an XOR with the sign
 * bit of |x| - |y| resp. of x; the sign of y is copied last (one v_bfi).  1 - (1 - t) differs from t by < 6e-8. */
__device__ __forceinline__ float octants(float t, float x, float y, float ax, float ay)
{
    const unsigned int sgn = 0x80000000u;
    const float a = __uint_as_float(__float_as_uint(1.0f - t) ^ (__float_as_uint(ax - ay) & sgn));
    const float b = __uint_as_float(__float_as_uint(1.0f + a) ^ (__float_as_uint(x) & sgn));
    return copysignf(2.0f - b, y);
}

/* atan2(y, x) * 4/pi in (-4, 4]: octant reduction + odd degree-11 minimax polynomial of
 * atan(r) * 4/pi on [0, 1] (max error 2.2e-6 bins = 1.7e-6 rad, fitted offline), v_rcp instead of
 * a division.  Only used for the descriptor's SOFT orientation binning, which is continuous in
 * the angle (the reference uses fast intrinsics there too, s_desc_loop.cu:48,97). */
/* ISA: magnitude + angle */
__device__ __forceinline__ float atan2_bins(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    /* the larger magnitude is kept off zero inside the v_max3 (0 / 1e-30 = 0: the same result as the test for a zero
     * gradient it replaces), the sign of y is copied with one v_bfi (t >= 0; y = -0 gives -t, the same angle mod 8) */
    const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
    const float r = mn * __builtin_amdgcn_rcpf(mx);
    const float s = r * r;
    float       p = fmaf(-0.01492126751691103f, s, 0.06703268736600876f);
    p = fmaf(p, s, -0.14823880791664124f);
    p = fmaf(p, s, 0.24642325937747955f);
    p = fmaf(p, s, -0.42350852489471436f);
    p = fmaf(p, s, 1.2732105255126953f);
    return octants(p * r, x, y, ax, ay);
}

/* The same with a degree-9 polynomial (max error 1.5e-5 bins = 1.2e-5 rad, fitted offline as a minimax problem) for the
 * descriptor's soft binning alone: the angle only splits a sample's weight between two neighbouring bins, so 1e-5 bins
 * moves 1e-5 of a weight.  The largest magnitude is kept off zero inside the v_max3 (a zero gradient has zero weight
 * whatever its angle), the sign of y is copied with one v_bfi. */
__device__ __forceinline__ float atan2_bins9(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
    const float r = mn * __builtin_amdgcn_rcpf(mx);
    const float s = r * r;
    float       p = fmaf(0.026540832594037056f, s, -0.1084246039390564f);
    p = fmaf(p, s, 0.22938621044158936f);
    p = fmaf(p, s, -0.4205572307109833f);
    p = fmaf(p, s, 1.2730693817138672f);
    return octants(p * r, x, y, ax, ay);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) fix64 lds_fix64;
typedef __attribute__((address_space(3))) unsigned int lds_u32;

/* ISA: end */
#ifndef KP_NW
#define KP_NW 1 /* waves per workgroup of k_orientation / k_descriptor: the waves are independent (one keypoint each) */
#endif
constexpr int ORI_COPIES = 4; /* private histogram copies by lane: neighbouring pixels often share a bin */

/*
 * ori_par, first half (s_orientation.cu:60-139): the weighted 36-bin gradient-orientation histogram of one
 * extremum, one wave per extremum.  The second half -- smoothing, peak interpolation, the four best peaks --
 * is serial per extremum and runs with one LANE per extremum in k_scan_local (as a wave-wide epilogue here it
 * was ~350 of the kernel's ~1250 vector instructions per extremum, 62 of them dependent cross-lane shuffles).
 * The raw histogram crosses in `ohist` (36 floats per extremum).
 */
__global__ __launch_bounds__(64 * KP_NW) void k_orientation(const PyrDesc* __restrict__ pdp, BatchDesc bd, SiftConsts sc,
                                                     int filtered, int hist_cap)
{
    /* this image's planes and lists (the slot of blockIdx.y) */
    const float* __restrict__   arena = bd.s[blockIdx.y].arena;
    Counters* __restrict__      ct = bd.s[blockIdx.y].ct;
    const InitExt* __restrict__ iext = filtered ? bd.s[blockIdx.y].iext2 : bd.s[blockIdx.y].iext;
    float* __restrict__         ohist = bd.s[blockIdx.y].ohist;
    const int n_oct = pdp->n_oct, L = pdp->L;
    /* bin 36 is bin 0 (the histogram is circular): a sample that rounds up to 36 is added there and joins bin 0 at the
     * read-out, so the loop needs no wrap-around select */
    constexpr int    NB = PS_ORI_NBINS + 2;
    __shared__ fix64 s_hist[KP_NW][ORI_COPIES][NB];
    const int        wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fix64*           hall = &s_hist[wave][0][0];
    fix64*           hist = s_hist[wave][lane & (ORI_COPIES - 1)];

    __shared__ int ps[PS_MAX_OCT + 1];
    if (threadIdx.x == 0) ext_prefix(ct, sc, n_oct, ps);
    __syncthreads();
    const int total = min(ps[n_oct], hist_cap);

    for (XcdSlice sl = xcd_slice<KP_NW>(); sl.more(total); sl.s += sl.step) {
        const int g = sl.index();
        if (g >= total) continue;
        int o = 0;
        while (o + 1 < n_oct && g >= ps[o + 1]) o++;
        o = uniform(o); /* the octave record and the list entry are then fetched through the scalar cache */
        const InitExt  ie = iext[(size_t)o * sc.max_extrema + (g - uniform(ps[o]))];
        const OctDesc* od = &pdp->o[o];
        const int      w = uniform(od->w), h = uniform(od->h), pitch = uniform(od->pitch);
        const int      lvl = uniform(min(max(ie.lpos, 0), L - 1));
        const float*   layer = arena + uniform64(od->data_off + lvl * od->plane_stride);

        for (int k = lane; k < ORI_COPIES * NB; k += 64) hall[k] = 0ull;
        wave_lds_sync();

        const float x = uniformf(ie.xpos), y = uniformf(ie.ypos), sig = uniformf(ie.sigma);
        const float sigw = ORI_WINFACTOR * sig;
        const int   rad = (int)roundf(3.0f * sigw);
        const float factor2 = (-0.5f / (sigw * sigw)) * 1.4426950408889634f; /* in powers of two */
        const int   sq_thres = rad * rad;
        const int   xmin = max(1, (int)roundf(x) - rad);
        const int   xmax = min(w - 2, (int)roundf(x) + rad);
        const int   ymin = max(1, (int)roundf(y) - rad);
        const int   ymax = min(h - 2, (int)roundf(y) + rad);
        const int   wx = xmax - xmin + 1;
        const int   hy = ymax - ymin + 1;
#ifdef ORI_PROBE_NOLOOP
        const int   loops = 0;
#else
        const int   loops = (wx > 0 && hy > 0) ? wx * hy : 0;
#endif
        /* row of flat index i = floor((i + 0.5) / wx): the half keeps the quotient at least 0.5 / wx away from an integer,
         * far more than the ulp by which v_rcp may miss 1 / wx, so the approximate reciprocal gives the exact row */
        const float inv_wx = __builtin_amdgcn_rcpf((float)max(wx, 1));
        const float wxf = (float)wx, xminf = (float)xmin, yminf = (float)ymin, pitch4f = (float)(4 * pitch);
        const float* corner = layer + (size_t)(ymin * pitch + xmin);

        /* software pipeline as in k_descriptor: request the taps of sample i+64 while binning sample i.  Row, column
         * and the byte offset of the taps are formed in floats (exact small integers): one conversion, one floor and FMAs
         * instead of two conversions, two 24-bit multiplies and a shift-add -- half-rate instructions on gfx950
         * (tools/ubench/valu_rate.hip).  (fx, fy) = the pixel, as the floats the reference converts its ints to. */
        auto coord = [&](int i, float& fx, float& fy, unsigned int& off) {
            const float fi = (float)i;
            const float frow = floorf((fi + 0.5f) * inv_wx);
            const float fcol = fmaf(-frow, wxf, fi);
            fx = fcol + xminf;
            fy = frow + yminf;
            off = (unsigned int)fmaf(frow, pitch4f, fcol * 4.0f);
        };
        /* one sample: pixel (fx, fy) with gradient (gdx, gdy) -> at most one fixed-point LDS atomic */
        auto bin = [&](float fx, float fy, float gdx, float gdy, bool live) {
            const float dx = fx - x;
            const float dy = fy - y;
            /* int truncation, s_orientation.cu:123 -- as a float (the values are far below 2^24, so truncf, the
             * comparison with rad^2 and the product with `factor` give what the int round trip gives) */
            const float sq_dist = truncf(dx * dx + dy * dy);
            if (live && sq_dist <= (float)sq_thres) {
                /* hypotf * expf(sq_dist * factor) * 2^20 (the fixed-point scale of to_fix): the scale and log2(e) go into
                 * the exponent, so the window weight is one FMA and one v_exp */
                const float wfix = __builtin_amdgcn_sqrtf(fmaf(gdx, gdx, gdy * gdy)) * __builtin_amdgcn_exp2f(fmaf(sq_dist, factor2, 20.0f));
                /* The bin is a HARD decision, so the angle needs libm-grade accuracy -- but only when it lies next to a
                 * bin edge: the cheap atan2 (the descriptor's one-reciprocal degree-11 polynomial in units of pi / 4:
                 * error < 1.7e-6 rad = 1e-5 of these bins) decides every sample farther than 1e-4 bins from an edge
                 * identically to the accurate one, which the others (1 in 5000) then take. */
                float fb = fmaf(atan2_bins(gdy, gdx), (float)PS_ORI_NBINS / 8.0f, (float)PS_ORI_NBINS / 2.0f);
                /* next to an edge the value is formed exactly as the oracle forms it (product, then IEEE quotient): gradients
                 * of exactly 45 degrees -- frequent in level 0 of the up-scaled octave -- sit ON the edge 22.5, 31.5, ...,
                 * where the two formulas round to different sides */
                if (fabsf((fb - floorf(fb)) - 0.5f) < 1e-4f) fb = (float)PS_ORI_NBINS * (atan2_acc(gdy, gdx) + F_PI) / F_PI2;
                /* roundf(fb) for fb in [0, 36]: floor(fb + 0.5), one instruction; 36 is bin 0 (NB above) */
                int bidx;
                asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(bidx) : "v"(fb));
                atomicAdd(&hist[bidx], (fix64)(unsigned int)wfix);
            }
        };
        /* two register sets take turns as "being binned" and "in flight", as in k_descriptor (unrolled by two, so no
         * value is copied from one role to the other) */
        float        xa = 0.0f, ya = 0.0f, xb = 0.0f, yb = 0.0f;
        unsigned int oa = 0, ob = 0;
        float        a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
        const Taps   taps(corner, pitch);
        if (loops > 0) {
            coord(min(lane, loops - 1), xa, ya, oa);
            taps.load_b(oa, a0, a1, a2, a3);
        }
        for (int i = lane; i < loops; i += 128) {
            coord(min(i + 64, loops - 1), xb, yb, ob);
            taps.load_b(ob, b0, b1, b2, b3);
            bin(xa, ya, a0 - a1, a2 - a3, true);
            coord(min(i + 128, loops - 1), xa, ya, oa);
            taps.load_b(oa, a0, a1, a2, a3);
            bin(xb, yb, b0 - b1, b2 - b3, i + 64 < loops);
        }
        wave_lds_sync();

        if (lane < PS_ORI_NBINS) {
            fix64 acc = 0ull;
#pragma unroll
            for (int k = 0; k < ORI_COPIES; k++) acc += hall[k * NB + lane] + (lane == 0 ? hall[k * NB + PS_ORI_NBINS] : 0ull);
            ohist[(size_t)g * PS_ORI_NBINS + lane] = from_fix(acc);
        }
        wave_lds_sync();
    }
}

/*
 * ori_par, second half (s_orientation.cu:142-240), FOUR lanes per extremum, nine of the 36 bins each: six circular
 * box-filter passes, parabolic peak interpolation, the (at most four) best peaks within 80 % of the best.  The lanes of
 * a quad exchange their edge bins and their candidates with DPP quad permutes (register-to-register, no LDS);
 * BitonicSort::Warp32::sort64 (common/warp_bitonic_sort.h:35-78) becomes four selection rounds, ties to the lower
 * bin.  Everything is indexed statically, so the bins live in registers.  (One LANE per extremum -- 36 bins in
 * registers, no exchange at all -- needs only 1200 waves for a 1080p image: 24 us at barely one wave per SIMD.)
 * All four lanes of a quad must be active; all return the same result.
 */
__device__ __forceinline__ float quad_from_prev(float v) /* lane j of a quad reads lane (j + 3) & 3 */
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93, 0xf, 0xf, false));
}
__device__ __forceinline__ float quad_from_next(float v) /* lane j reads lane (j + 1) & 3 */
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x39, 0xf, 0xf, false));
}
template <int CTRL> /* 0xB1: lane ^ 1, 0x4E: lane ^ 2 */
__device__ __forceinline__ int quad_xchg(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
/* x / 3.0f, correctly rounded for normal x (Markstein: q = x * RN(1/3), one FMA residual, one FMA correction; checked
 * against IEEE division for every float of six binades, tools/check_div3.py) instead of the ten-instruction division */
__device__ __forceinline__ float div3(float x)
{
    const float r = 0.333333343267440796f; /* RN(1/3) */
    const float q = x * r;
    return fmaf(fmaf(-3.0f, q, x), r, q);
}

__device__ __forceinline__ int ori_peaks_quad(const float* __restrict__ hsrc, int j, bool have, float* angle_out)
{
    constexpr int N = PS_ORI_NBINS, M = N / 4; /* 9 bins per lane: 9j .. 9j+8 */
    float         h[M], t[M];
#pragma unroll
    for (int b = 0; b < M; b++) h[b] = have ? hsrc[M * j + b] : 0.0f;
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        {
            const float lft = quad_from_prev(h[M - 1]), rgt = quad_from_next(h[0]);
#pragma unroll
            for (int b = 0; b < M; b++) t[b] = div3(((b == 0 ? lft : h[b - 1]) + h[b]) + (b == M - 1 ? rgt : h[b + 1]));
        }
        {
            const float lft = quad_from_prev(t[M - 1]), rgt = quad_from_next(t[0]);
#pragma unroll
            for (int b = 0; b < M; b++) h[b] = div3(((b == 0 ? lft : t[b - 1]) + t[b]) + (b == M - 1 ? rgt : t[b + 1]));
        }
    }
    float refined[M], yval[M];
    {
        const float lft = quad_from_prev(h[M - 1]), rgt = quad_from_next(h[0]);
#pragma unroll
        for (int b = 0; b < M; b++) {
            const int   bin = M * j + b;
            const int   prev = (bin == 0) ? N - 1 : bin - 1;
            const float hp = (b == 0) ? lft : h[b - 1], hv = h[b], hn = (b == M - 1) ? rgt : h[b + 1];
            bool        predicate = hv > fmaxf(hp, hn);
            const float num = predicate ? 3.0f * hp - 4.0f * hv + 1.0f * hn : 0.0f;
            const float denB = predicate ? 2.0f * (hp - 2.0f * hv + hn) : 1.0f;
            const float newbin = num / denB;
            predicate = (predicate && newbin >= 0.0f && newbin <= 2.0f);
            refined[b] = predicate ? prev + newbin : -1.0f;
            yval[b] = predicate ? -(num * num) / (4.0f * denB) + hp : -INFINITY;
        }
    }
    unsigned int used = 0u; /* of this lane's nine bins */
    float        best0 = 0.0f;
    int          angles = 0;
#pragma unroll
    for (int k = 0; k < POPSIFT_HIP_ORI_MAX; k++) {
        /* best unused bin of this lane: largest value, lowest bin on ties (every lane has unused bins: 9 > 4) */
        float bv = 0.0f, br = -1.0f;
        int   bi = -1;
#pragma unroll
        for (int b = 0; b < M; b++) {
            const bool take = !((used >> b) & 1u) && (bi < 0 || yval[b] > bv);
            bv = take ? yval[b] : bv;
            br = take ? refined[b] : br;
            bi = take ? M * j + b : bi;
        }
        /* ... of the quad */
        {
            const float ov = __int_as_float(quad_xchg<0xB1>(__float_as_int(bv))), orf = __int_as_float(quad_xchg<0xB1>(__float_as_int(br)));
            const int   oi = quad_xchg<0xB1>(bi);
            const bool  take = ov > bv || (ov == bv && oi < bi);
            bv = take ? ov : bv;
            br = take ? orf : br;
            bi = take ? oi : bi;
        }
        {
            const float ov = __int_as_float(quad_xchg<0x4E>(__float_as_int(bv))), orf = __int_as_float(quad_xchg<0x4E>(__float_as_int(br)));
            const int   oi = quad_xchg<0x4E>(bi);
            const bool  take = ov > bv || (ov == bv && oi < bi);
            bv = take ? ov : bv;
            br = take ? orf : br;
            bi = take ? oi : bi;
        }
        const int mine = bi - M * j;
        if (mine >= 0 && mine < M) used |= 1u << mine;
        if (k == 0) best0 = bv;
        /* the candidates come in descending order, so the accepted ones are a prefix */
        if (bv >= 0.8f * best0) {
            float chosen_bin = br;
            if (chosen_bin >= N) chosen_bin -= N;
            angle_out[k] = fmaf(F_PI2 * chosen_bin, 1.0f / N, -F_PI);
            angles = k + 1;
        } else {
            angle_out[k] = 0.0f;
        }
    }
    return angles;
}

/* The per-descriptor constants of the loop descriptor (k_descriptor), computed where one lane serves one extremum.
 * fbits: each cell word packs two 32-bit fixed-point sums (low = share of orientation bin b, high = share of bin b+1
 * from samples whose lower bin is b), so ONE 64-bit LDS atomic per cell serves both bins of a sample; the scale 2^fbits
 * keeps the worst-case low sum below 2^32 (no carry into the high half): a sample weighs <= 361 (|gradient| of a
 * 0..255 plane, unit window weights) and a cell sees at most (2.83*SBP+1)^2 pixels. */
__device__ __forceinline__ DescRec make_desc_rec(const PyrDesc* __restrict__ pdp, const Ext& e, float angle, float cos_t, float sin_t)
{
    DescRec        r;
    const OctDesc* od = &pdp->o[e.octave];
    const int      lvl = min(max(e.lpos, 0), pdp->L - 1);
    const long long off = od->data_off + lvl * od->plane_stride;
    const float    x = e.xpos, y = e.ypos;
    const float    SBP = fabsf(DESC_MAGNIFY * e.sigma);
    const float    cell_px = (2.83f * SBP + 1.0f) * (2.83f * SBP + 1.0f);
    /* ... and below 2^23 per sample (361 * 2^14 < 2^23): k_descriptor rounds by adding 2^23 */
    const int      fbits = min(max(31 - (int)ceilf(log2f(361.0f * cell_px)), 2), 14);
    const float    csbp = cos_t * SBP, ssbp = sin_t * SBP;
    const float    bsz = fabsf(csbp) + fabsf(ssbp);
    /* union of the 16 cell boxes: cell centres reach 1.5 * bsz, each box adds bsz */
    const float    ext_r = 2.5f * bsz;
    const int      xmin = max(1, (int)floorf(x - ext_r));
    const int      ymin = max(1, (int)floorf(y - ext_r));
    const int      xmax = min(od->w - 2, (int)floorf(x + ext_r) + 1);
    const int      ymax = min(od->h - 2, (int)floorf(y + ext_r) + 1);
    r.x = x;
    r.y = y;
    r.crsbp = cos_t / SBP;
    r.srsbp = sin_t / SBP;
    r.ang_bins = angle * (4.0f / F_PI);
    r.fscale = scalbnf(1.0f, fbits);
    r.xymin = ((unsigned int)xmin & 0xffffu) | ((unsigned int)ymin << 16);
    r.xymax = ((unsigned int)xmax & 0xffffu) | ((unsigned int)ymax << 16);
    r.off_lo = (unsigned int)((unsigned long long)off & 0xffffffffull);
    r.off_hi = (unsigned int)((unsigned long long)off >> 32);
    r.misc = (unsigned int)od->pitch | ((unsigned int)fbits << 16) | ((SBP != 0.0f) ? (1u << 24) : 0u);
    r.pad = 0u;
    return r;
}

/* ------------------------------------------------------------------- scan */

/*
 * Exclusive prefix sum of num_ori over all extrema -> idx_ori, the reverse map
 * descriptor -> extremum, and the totals (replaces ori_prefix_sum's single
 * 32x32 block looping over everything, s_orientation.cu:303-345).  Two launches:
 * k_scan_local scans 64-extrema chunks and leaves one partial per chunk;
 * k_scan_apply (256 extrema per workgroup) adds the sum of the preceding partials
 * (<= 14 100 values for the default 9 x 100000 capacity, ~1 200 for a 1080p image,
 * summed redundantly per workgroup) and writes the map and the counters.
 */
/* extrema per lane: 1 -- the per-extremum work of k_scan_apply (feature record, double-precision cos / sin)
 * is serial per lane, so wide beats deep (8 per lane: 22.8 us, 2: 10.2 us, 1: 7.5 us) */
constexpr int SCAN_ITEMS = 1;
constexpr int SCAN_CHUNK = 256 * SCAN_ITEMS;
/* k_scan_local: lanes per workgroup (four per extremum) and the extrema it scans; SCAN_SUB local chunks per apply chunk.
 * The kernel is a latency chain per workgroup (record, histogram, peaks, scan, store), so many small workgroups beat
 * few large ones: 128 / 256 / 512 / 1024 lanes -> 14.0 / 13.2 / 14.3 / 19.9 us for the 77 000 extrema of a 1080p image */
constexpr int SCAN_LT = 256;
constexpr int SCAN_LCHUNK = SCAN_LT / 4;
constexpr int SCAN_SUB = SCAN_CHUNK / SCAN_LCHUNK;

__device__ __forceinline__ int clamped_total(const Counters* ct, const SiftConsts& sc, int n_oct)
{
    int acc = 0;
    for (int o = 0; o < n_oct; o++) acc += min(ct->ext_ct[o], sc.max_extrema);
    return acc;
}

__global__ __launch_bounds__(SCAN_LT) void k_scan_local(const PyrDesc* __restrict__ pdp, SiftConsts sc, BatchDesc bd,
                                                     int filtered, int hist_cap)
{
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const InitExt* __restrict__  iext = filtered ? bd.s[blockIdx.y].iext2 : bd.s[blockIdx.y].iext;
    const float* __restrict__    ohist = bd.s[blockIdx.y].ohist;
    Ext* __restrict__            ext = bd.s[blockIdx.y].ext;
    int* __restrict__            partial = bd.s[blockIdx.y].partial;
    static_assert(SCAN_CHUNK % SCAN_LCHUNK == 0 && SCAN_LT % 64 == 0, "four lanes per extremum, whole waves");
    __shared__ int s_wsum[SCAN_LT / 64];
    __shared__ int s_ps[PS_MAX_OCT + 1];
    const int      n_oct = pdp->n_oct;
    if (threadIdx.x == 0) ext_prefix(ct, sc, n_oct, s_ps);
    __syncthreads();
    const int total = s_ps[n_oct];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = tid & 3;
    /* the launch is sized for the device, not for the capacity of the lists: workgroups stride over the chunks */
    for (int chunk = blockIdx.x; chunk * SCAN_LCHUNK < total; chunk += gridDim.x) {
    const int  base = chunk * SCAN_LCHUNK;
    const int  g = base + (tid >> 2);
    const bool valid = g < total;
    const int  gc = min(g, total - 1); /* lanes beyond the list compute along (the quad exchanges need them) */
    /* the second half of ori_par for extremum g, then its Extremum record (sift_extremum.h:40-51) */
    int o = 0;
    while (o + 1 < n_oct && gc >= s_ps[o + 1]) o++;
    const InitExt ie = iext[(size_t)o * sc.max_extrema + (gc - s_ps[o])];
    Ext           e;
    e.xpos = ie.xpos;
    e.ypos = ie.ypos;
    e.lpos = ie.lpos;
    e.sigma = ie.sigma;
    e.octave = o;
    e.idx_ori = 0;
    /* beyond the histogram buffer (the host grows it and re-runs the stages): no orientation */
    const bool have = gc < hist_cap;
    const int  n = ori_peaks_quad(ohist + (size_t)min(gc, max(hist_cap - 1, 0)) * PS_ORI_NBINS, j, have, e.orientation);
    e.num_ori = have ? n : 0;
    if (!have)
        for (int q = 0; q < POPSIFT_HIP_ORI_MAX; q++) e.orientation[q] = 0.0f;
    const int self = (valid && j == 0) ? e.num_ori : 0;

    const int incl = wave_incl_scan(self);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += s_wsum[w];
    if (valid && j == 0) {
        e.idx_ori = woff + incl - self; /* chunk-local for now */
        ext[g] = e;
    }
    if (tid == SCAN_LT - 1) partial[chunk] = woff + incl;
    __syncthreads(); /* s_wsum is reused by the next chunk */
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(const PyrDesc* __restrict__ pdp, SiftConsts sc, BatchDesc bd,
                                                    int with_drec, int desc_cap)
{
    Counters* __restrict__            ct = bd.s[blockIdx.y].ct;
    Ext* __restrict__                 ext = bd.s[blockIdx.y].ext;
    const int* __restrict__           partial = bd.s[blockIdx.y].partial;
    int* __restrict__                 map = bd.s[blockIdx.y].map;
    float2* __restrict__              rot = bd.s[blockIdx.y].rot;
    DescRec* __restrict__             drec = with_drec ? bd.s[blockIdx.y].drec : nullptr;
    popsift_hip_feature* __restrict__ feats = bd.s[blockIdx.y].feats;
    __shared__ int s_red[4];
    __shared__ int s_ps[PS_MAX_OCT + 1];
    const int      n_oct = pdp->n_oct;
    const int      tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int acc = 0;
        for (int o = 0; o < PS_MAX_OCT; o++) {
            s_ps[o] = acc;
            acc += (o < n_oct) ? min(ct->ext_ct[o], sc.max_extrema) : 0;
        }
        s_ps[PS_MAX_OCT] = acc;
    }
    __syncthreads();
    const int total = s_ps[PS_MAX_OCT];
    const int nb = (total + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nb == 0 && blockIdx.x == 0 && tid == 0) { /* no extrema at all: the counters still have to be written */
        ct->ori_total = 0;
        ct->ext_total = 0;
        for (int o = 0; o < PS_MAX_OCT; o++) {
            ct->ext_ct[o] = 0;
            ct->ext_ps[o] = 0;
        }
    }
    /* the launch is sized for the device, not for the capacity of the lists: workgroups stride over the chunks */
    for (int chunk = blockIdx.x; chunk < nb; chunk += gridDim.x) {
    const int base = chunk * SCAN_CHUNK;

    /* offset of this chunk = sum of the partials of all preceding chunks (SCAN_SUB local chunks per chunk here) */
    int acc = 0;
    for (int b = tid; b < chunk * SCAN_SUB; b += 256) acc += partial[b];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    const int chunk_offset = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    /* ... plus the local chunks of this chunk that precede the lane's own */
    int offset = chunk_offset;
    for (int k = 0; k < (tid * SCAN_ITEMS) / SCAN_LCHUNK; k++) offset += partial[chunk * SCAN_SUB + k];

    const int g0 = base + tid * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int g = g0 + k;
        if (g < total) {
            Ext       e = ext[g];
            const int idx = e.idx_ori + offset;
            const int n = e.num_ori;
            ext[g].idx_ori = idx;
            /* prep_features (sift_pyramid.cu:249-279): positions and scale in input-image coordinates,
             * descriptor indices instead of pointers */
            popsift_hip_feature f;
            const float         scl = powf(2.0f, (float)(e.octave - sc.up_fac_int));
            f.debug_octave = e.octave;
            f.xpos = e.xpos * scl;
            f.ypos = e.ypos * scl;
            f.sigma = e.sigma * scl;
            f.num_ori = n;
#pragma unroll
            for (int q = 0; q < POPSIFT_HIP_ORI_MAX; q++) {
                const bool on = q < n && idx + q < desc_cap;
                f.desc_idx[q] = on ? idx + q : -1;
                f.orientation[q] = (q < n) ? e.orientation[q] : 0.0f;
                if (on) {
                    map[idx + q] = g;
                    /* the rotation of the descriptor frame, in double, here where registers are plentiful */
                    float sn, cs;
                    sincos_cr(e.orientation[q], sn, cs);
                    rot[idx + q] = make_float2(cs, sn);
                    if (drec) drec[idx + q] = make_desc_rec(pdp, e, e.orientation[q], cs, sn);
                }
            }
            feats[g] = f;
            /* first extremum of an octave: start of that octave's descriptors (dct.ori_ps) */
            for (int o = 0; o < n_oct; o++)
                if (g == s_ps[o] && s_ps[o + 1] > s_ps[o]) ct->ori_ps[o] = idx;
        }
    }
    if (tid == 0 && chunk == nb - 1) {
        int last = chunk_offset;
        for (int k = 0; k < SCAN_SUB && (chunk * SCAN_SUB + k) * SCAN_LCHUNK < total; k++) last += partial[chunk * SCAN_SUB + k];
        ct->ori_total = last;
        ct->ext_total = total;
        for (int o = 0; o < PS_MAX_OCT; o++) {
            /* the reference clamps with atomicMin in the extrema kernel, s_extrema.cu:558 */
            ct->ext_ct[o] = (o < n_oct) ? min(ct->ext_ct[o], sc.max_extrema) : 0;
            ct->ext_ps[o] = s_ps[o];
        }
    }
    __syncthreads(); /* s_red is reused by the next chunk */
    }
}

/* ------------------------------------------------------------- descriptor */

/*
 * One wave per (extremum, orientation), no workgroup barrier.  The reference gives every one of the 16 cells
 * its own warp, which re-computes the gradient of each patch pixel for up to four overlapping cells
 * (s_desc_loop.cu:78-122).  Here each patch pixel is visited once: its gradient is computed once and its
 * contribution is spread to the (at most) 2x2 cells whose unit square contains it -- the same sample set and
 * weights, a different summation order.  The 128-bin histogram lives in LDS (per wave, fixed point: exact and
 * order-independent sums), is normalised in registers and leaves as two coalesced 256 B rows.
 *
 * LDS atomics set the pace of the first version (rocprofv3, round 2: the LDS was busy 73 % of the kernel's cycles,
 * 63 % of that bank-conflict cycles): 64 lanes on 64 consecutive pixels of a patch row mostly fall into the same
 * cell row, a few cells and -- gradients being smooth -- one or two orientation bins, i.e. up to 16 lanes on one
 * address, and the two private copies sat exactly 1 KiB apart, on the same banks.  Now
 *   - the wave works as four groups of 16 lanes, each on its own quarter of the sample list (different cell rows),
 *   - four private copies by lane (neighbouring lanes never share a word),
 *   - the word of (copy c, cell iy ix, bin b) is c*140 + iy*36 + ix*8 + ((b + c + 4*(ix & 1)) & 7) (DESC_CS, DESC_RS
 *     below): for lanes that agree in the bin, the copies, the two cell columns a group straddles and the four groups'
 *     cell rows land on different bank pairs, and the four words of a sample are one address + immediate offsets.
 */
#ifndef DESC_NCOPY
#define DESC_NCOPY 4
#endif
constexpr int DESC_COPIES = DESC_NCOPY;
/* LINEAR histogram layout: the word (8 bytes) of (copy c, cell row iy, cell column ix, slot s) is
 * c * DESC_CS + iy * DESC_RS + ix * 8 + s with a cell-row stride of 36 words instead of 32.  The four words of a sample
 * are then ONE address plus the immediate offsets 0 / 64 / 288 / 352 bytes (no per-cell index arithmetic, cells -1 and
 * 4 fall out naturally), and the four extra words per row shift consecutive cell rows by four bank pairs, which does
 * what the (ix + iy) rotation of the 32-word layout did for the four lane groups working on different rows. */
constexpr int DESC_RS = 36;                /* words per cell row */
constexpr int DESC_CS = 3 * DESC_RS + 32;  /* words per copy (the last row needs no padding): 140 */
constexpr int DESC_MAXROWS = 128; /* patch rows handled by the span path */
#ifndef DESC_GROUPS
/* Lane groups, each on its own 1 / DESC_GROUPS of the sample list.  Round 2 ran FOUR (16 lanes each, on four distant
 * parts of the patch: fewer lanes of a wave fall into one cell, 61 % fewer LDS bank conflicts) while the kernel was bound
 * by vector issue and LDS.  Round 3 took a fifth of the issue slots out of the loop, and what bound the kernel next was the
 * L1 -> L2 request stream of its taps (L1 hit rate 68 %, the L1 stalled on pending data 46 % of the time,
 * tools/desc_batch_counters.sh): four streams touch four times the cache lines per wave step.  ONE group -- 64 lanes on
 * 64 consecutive samples, i.e. on one and a half to two neighbouring patch rows, whose up / down taps are each other's
 * centre rows -- leaves the kernel's own time unchanged and makes the timed loop 4.8 % faster (1 / 2 / 4 / 8 groups:
 * 3.36 / 3.36 / 3.20 / 2.74 Gpix/s): the requests it no longer makes are L2 bandwidth for the kernels beside it. */
#define DESC_GROUPS 1
#endif
constexpr int DESC_GL = 64 / DESC_GROUPS; /* lanes per group */

__global__ __launch_bounds__(64 * KP_NW, 8) void k_descriptor(BatchDesc bd, SiftConsts sc, int desc_cap)
{
    const float* __restrict__   arena = bd.s[blockIdx.y].arena;
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const DescRec* __restrict__ drec = bd.s[blockIdx.y].drec;
    float* __restrict__         desc = bd.s[blockIdx.y].desc;
    /* LDS of a wave: the row records first, then the histogram copies.  The records -- per patch row of the current pass:
     * flat index of its first sample (low 16 bits) | that index minus the first column of its span relative to xmin
     * (high 16 bits, signed), so that column = flat index - (word >> 16) -- lie IN FRONT of the histograms so that the
     * address of cell (-1, -1) of copy 0, which the cell arithmetic forms although no weight ever goes there (352 bytes
     * before the copy), is still a non-negative LDS address. */
    constexpr int ROW_WORDS = (DESC_MAXROWS + 1 + 3) & ~3; /* 132: the histograms start on a 16-byte boundary */
    static_assert(ROW_WORDS * 4 >= 352, "cell (-1, -1) of copy 0 stays inside the wave's LDS");
    __shared__ __attribute__((aligned(16))) unsigned int s_lds[KP_NW][ROW_WORDS + 2 * DESC_COPIES * DESC_CS];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int     grp = lane / DESC_GL, sub = lane % DESC_GL, cpy = lane & (DESC_COPIES - 1);
    unsigned int* rinfo = s_lds[wave];
    fix64*        hall = (fix64*)(s_lds[wave] + ROW_WORDS);
    /* 32-bit LDS addresses: the row records, this lane's histogram copy (as a float: the cell addresses are formed by FMAs) */
    const unsigned int rbase32 = (unsigned int)(size_t)(lds_u32*)rinfo;
    const float        hbasef = (float)(unsigned int)(size_t)(lds_fix64*)(hall + cpy * DESC_CS);
    const float        cpy8f = (float)(cpy << 3);
    const int     total = min(ct->ori_total, desc_cap);
    /* rows per pass: DESC_MAXROWS, or fewer when a test asks for it (popsift_hip_debug_set DESC_ROWS) */
    const int     maxrows = min(max(sc.desc_rows, 4), DESC_MAXROWS);

    for (XcdSlice sl = xcd_slice<KP_NW>(); sl.more(total); sl.s += sl.step) {
        const int d = sl.index();
        if (d >= total) continue;
        /* everything about the descriptor is wave-uniform and was worked out by k_scan_apply (DescRec): three 16-byte
         * loads of one address, moved to scalar registers (the taps become scalar-base + 32-bit-offset loads) */
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4*          rp = reinterpret_cast<const u4*>(drec + d);
        const u4           q0 = rp[0], q1 = rp[1], q2 = rp[2];
        const float        x = uniformf(__uint_as_float(q0.x)), y = uniformf(__uint_as_float(q0.y));
        const float        crsbp = uniformf(__uint_as_float(q0.z)), srsbp = uniformf(__uint_as_float(q0.w));
        const float        ang_bins = uniformf(__uint_as_float(q1.x));
        const int          pmin = uniform((int)q1.z), pmax = uniform((int)q1.w);
        const unsigned int misc = (unsigned int)uniform((int)q2.z);
        const int          pitch = (int)(misc & 0xffffu), fbits = (int)((misc >> 16) & 0xffu);
        const float*       layer = arena + uniform64((long long)(((unsigned long long)q2.y << 32) | q2.x));

        {
            /* the zero pair is made here, every time: left to itself the compiler keeps one alive across the whole sample
             * loop, which is the 65th and 66th vector register (a spill, and with it scratch set-up for every wave) */
            unsigned int zlo, zhi;
            asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(zlo), "=v"(zhi));
            const fix64 zero = ((fix64)zhi << 32) | zlo;
#pragma unroll
            for (int k = 0; k < (DESC_COPIES * DESC_CS + 63) / 64; k++)
                if (lane + 64 * k < DESC_COPIES * DESC_CS) hall[lane + 64 * k] = zero;
        }

        if (misc & (1u << 24)) { /* DESC_MAGNIFY * sigma != 0 */
            const int   xmin = (int)(short)(pmin & 0xffff), ymin0 = pmin >> 16;
            const int   xmax = (int)(short)(pmax & 0xffff), ymax = pmax >> 16;
            const int   wx = xmax - xmin + 1;
            const int   hy_all = ymax - ymin0 + 1;
            const float inv_c = (fabsf(crsbp) > 1e-20f) ? __builtin_amdgcn_rcpf(crsbp) : 0.0f;
            const float inv_s = (fabsf(srsbp) > 1e-20f) ? __builtin_amdgcn_rcpf(srsbp) : 0.0f;
            /* the Gaussian window weight comes out of v_exp already multiplied by the fixed-point scale 2^fbits */
            const float ffbits = (float)fbits;

            /* A patch is walked in passes of at most `maxrows` rows (ONE pass for every patch of sigma0 <= 1.78 at three
             * levels; the largest the library accepts -- sigma0 = 2, two levels -- spans 173 rows).  The histogram sums are
             * integers, so the split changes nothing. */
            for (int rb = 0; rb < hy_all && wx > 0; rb += maxrows) {
            const int   ymin = ymin0 + rb;
            const int   hy = min(hy_all - rb, maxrows);

            /* Row spans.  The samples that count lie in the square |u|,|v| < 2.5 (cell units, rotated
             * by ang), which fills only 1/(|cos|+|sin|)^2 = 50..100 % of its bounding box.  Per patch
             * row the square cuts out ONE column interval; the rows' intervals are laid end to end
             * (prefix sum) and the wave walks that flat list, so nearly every lane holds a sample
             * that passes the exact test below.  The intervals only have to be a superset of the
             * samples inside the square. */
            int T = 0;
            {
                int         carry = 0;
                for (int r0 = 0; r0 < hy; r0 += 64) {
                    const int r = r0 + lane;
                    int       len = 0, jlo = xmin;
                    if (r < hy) {
                        const float dy = (float)(ymin + r) - y;
                        const float a = srsbp * dy, b = crsbp * dy;
                        float       lo = -1e30f, hi = 1e30f;
                        bool        ok = true;
                        if (inv_c != 0.0f) { /* |crsbp*dx + a| < 2.5 */
                            const float t0 = (-2.5f - a) * inv_c, t1 = (2.5f - a) * inv_c;
                            lo = fminf(t0, t1);
                            hi = fmaxf(t0, t1);
                        } else {
                            ok = fabsf(a) < 2.5f;
                        }
                        if (inv_s != 0.0f) { /* |b - srsbp*dx| < 2.5 */
                            const float t0 = (b - 2.5f) * inv_s, t1 = (b + 2.5f) * inv_s;
                            lo = fmaxf(lo, fminf(t0, t1));
                            hi = fminf(hi, fmaxf(t0, t1));
                        } else {
                            ok = ok && (fabsf(b) < 2.5f);
                        }
                        /* columns j with lo < j - x < hi: floor(x + lo) + 1 .. ceil(x + hi) - 1, the ends moved out by 1e-3
                         * pixels -- a hundred times what the reciprocals and products above can be off by (a sample ON the
                         * edge |u| = 2.5 has weight zero in the cells that exist, so even a miss would be invisible).  Widening
                         * by a whole pixel on either side, as rounds 1 and 2 did, walked two samples per row for nothing:
                         * 5 % of the list (k_descriptor 370 -> 352 us, timed loop + 1.4 %). */
                        jlo = max(xmin, (int)floorf(x + lo - 1e-3f) + 1);
                        const int jhi = min(xmax, (int)ceilf(x + hi + 1e-3f) - 1);
                        len = ok ? max(jhi - jlo + 1, 0) : 0;
                        if (len == 0) jlo = xmin;
                    }
                    const int incl = wave_incl_scan(len);
                    const int start = carry + incl - len;
                    if (r < hy) rinfo[r] = (unsigned int)start | ((unsigned int)(start - (jlo - xmin)) << 16);
                    carry += __builtin_amdgcn_readlane(incl, 63);
                }
                T = carry;
                /* Two sentinels close the list: a flat index at or beyond T ends up in "row" hy at column -32767 -- a sample far
                 * outside the patch, which the in-square test drops (its taps are read at byte offset 0 of the patch: the
                 * negative offset saturates in the conversion) -- and stops there: the record after it starts at 65535.  So
                 * no index needs clamping and no lane needs a "past the end" flag. */
                if (lane == 0) {
                    rinfo[hy] = (unsigned int)T | 0x7fff0000u;
                    rinfo[hy + 1] = 0xffffu;
                }
            }
            wave_lds_sync();

            const int   loops = T;
            /* every lane group takes its own part of the list */
            /* ... a whole number of double steps each, so that no lane of a group ever walks into the next group's part: what
             * lies beyond the list's end drops out by itself (above) */
            const int   quarter = ((loops + DESC_GROUPS - 1) / DESC_GROUPS + 2 * DESC_GL - 1) & ~(2 * DESC_GL - 1);
            const int   ibeg = grp * quarter;
#ifdef DESC_PROBE_NOLOOP
            const int   iters = 0;
#else
            const int   iters = quarter / DESC_GL;
#endif
            /* the lane's place in the row records: LDS address of its current row's record, that row (counted from the top of
             * the PATCH, as a float: it only feeds FMAs), the record and the one of the row below */
            unsigned int rp = rbase32;
            float        fr = (float)rb;
            unsigned int cur = 0;
            unsigned int nxt = 0x7fff0000u | 0xffffu;
            if (loops > 0) {
                /* The row that holds the first sample of this lane's GROUP (coord() walks on from there to the lane's
                 * own): the one non-empty row r with start(r) <= key < start(r + 1).  Every row looks at its own
                 * interval, a ballot per group finds it -- two LDS reads instead of the seven dependent ones of a
                 * binary search per lane. */
                int grow = 0;
                for (int rbk = 0; rbk < hy; rbk += 64) {
                    const int  r = rbk + lane;
                    const bool in = r < hy;
                    const int  st = in ? (int)(rinfo[r] & 0xffffu) : 0x7fffffff;
                    const int  en = in ? (int)(rinfo[r + 1] & 0xffffu) : 0x7fffffff;
#pragma unroll
                    for (int g = 0; g < DESC_GROUPS; g++) {
                        const int                key = min(g * quarter, loops - 1);
                        const unsigned long long m = __ballot(st <= key && key < en);
                        if (m != 0ull && grp == g) grow = rbk + __ffsll((long long)m) - 1;
                    }
                }
                rp = rbase32 + 4u * (unsigned int)grow;
                fr = (float)(rb + grow); /* rows are counted from the top of the PATCH, whatever the pass: see coord() */
                cur = rinfo[grow];
                nxt = rinfo[grow + 1];
            }

            /* Sample (row r, column c), both counted from the corner (ymin0, xmin) of the PATCH (so that the arithmetic does not
             * depend on how the patch is cut into passes): cell-unit position
             *   u = crsbp * (xmin + c - x) + srsbp * (ymin0 + r - y),  v = crsbp * (ymin0 + r - y) - srsbp * (xmin + c - x)
             * (n = u - off, dn = n + off = u, s_desc_loop.cu:88-99) as two FMAs each on the small integers c and r; the
             * keypoint's offset from the patch corner is wave-uniform and goes into u0 / v0.  The taps are addressed from the
             * patch corner, so the element offset is r * pitch + c. */
            const float  ox = (float)xmin - x, oy = (float)ymin0 - y;
            const float  u0 = fmaf(crsbp, ox, srsbp * oy), v0 = fmaf(crsbp, oy, -srsbp * ox);
            const float* corner = layer + (size_t)(ymin0 * pitch + xmin);
            const float  pitch4f = (float)(4 * pitch);

            /* Two-stage software pipeline: the coordinates of the lane's next sample are computed and its four
             * gradient taps requested while the current one is being binned, so the L2 round trip of the taps
             * overlaps the arithmetic.  The loads are unconditional (in-bounds for every span position) to
             * keep the vmcnt waits counted.
             * What an instruction costs decides the form of everything below (tools/ubench/valu_rate.hip, MI355X): f32
             * add / mul / FMA, 32-bit add / and / xor / arithmetic shift issue at twice the rate of compares, selects,
             * conversions, floor, min / max, shifts-with-add and 24-bit multiplies.  So row and column enter as floats
             * (one conversion, the row advances by a float add), and the byte offset of the taps is one FMA -- exact:
             * 4 * (row * pitch + column) < 2^24 -- and one conversion, instead of a 24-bit multiply and a shift-add. */
            auto coord = [&](int i, unsigned int& off, float& u, float& v) { /* ISA: coordinates */
                /* the record of the row below the lane's current one stays in a register: no LDS read -- which would queue
                 * behind the four atomics the lane has just issued -- unless the lane moves on to another row
                 * (k_descriptor 421 -> 411 us) */
                while (i >= (int)(nxt & 0xffffu)) {
                    fr += 1.0f;
                    rp += 4u;
                    cur = nxt;
                    nxt = *(lds_u32*)(size_t)(rp + 4u);
                }
                const float fc = (float)(i - ((int)cur >> 16));
                u = fmaf(crsbp, fc, fmaf(srsbp, fr, u0));
                v = fmaf(crsbp, fr, fmaf(-srsbp, fc, v0));
                /* the end-of-list sentinel (column -32767 of a row that does not exist) makes this product negative, and the
                 * lane must then read offset 0: v_cvt_u32_f32 saturates a negative input to 0 by definition, whereas the
                 * C++ cast of an out-of-range float is undefined (fptoui poison) -- so the instruction is named, as for
                 * v_cvt_rpi_i32_f32 in k_orientation */
                asm("v_cvt_u32_f32 %0, %1" : "=v"(off) : "v"(fmaf(fr, pitch4f, fc * 4.0f)));
            };
            /* one sample: gradient (gx, gy) at cell-unit position (u, v) -> up to four 64-bit LDS atomics */
            unsigned int probe_acc = 0u; /* DESC_PROBE_NOATOMIC only */
            auto bin = [&](float u, float v, float gx, float gy) {
                if (fabsf(u) < 2.5f && fabsf(v) < 2.5f) { /* ISA: control */
                    /* ISA: magnitude + angle */
                    const float  mod = __builtin_amdgcn_sqrtf(fmaf(gx, gx, gy * gy));
                    /* exp(-(u^2+v^2)/8) * 2^fbits = 2^(fbits - (u^2+v^2) * log2(e)/8) */
                    const float  wm = __builtin_amdgcn_exp2f(fmaf(-0.18033688011112042f, fmaf(u, u, v * v), ffbits)) * mod;
                    /* gradient angle relative to the keypoint orientation in units of one bin; any multiple of 8 may be
                     * missing: floor / fraction / (& 7) below do not care */
                    const float tth = atan2_bins9(gy, gx) - ang_bins;
                    const float ffo = floorf(tth);
                    const float do0 = tth - ffo;

                    /* ISA: weights */
                    /* cell centres sit at integer tu, tv in 0..3; the sample feeds cells
                     * (cx0, cx0+1) x (cy0, cy0+1) with weights (1-fx, fx) x (1-fy, fy)
                     * -- the (1-|n.x|)(1-|n.y|) of s_desc_loop.cu:100-102 -- where inside 0..3 */
                    const float tu = u + 1.5f, tv = v + 1.5f;
                    const float fcx = floorf(tu), fcy = floorf(tv);
                    const float fx = tu - fcx, fy = tv - fcy;
                    const float wx0 = 1.0f - fx, wy0 = (1.0f - fy) * wm, wy1 = fy * wm;
                    const float w0 = 1.0f - do0;
                    /* Byte addresses of the four words: one base, the slot of the even and of the odd cell column, immediate
                     * offsets for the neighbours.  Slot of the lower bin: (bin + copy + 4 * (cx0 & 1)) mod 8 -- 4 * cx0 does
                     * for 4 * (cx0 & 1) under the mask, and the bin needs no mask of its own.  Cell indices and the bin are
                     * small integers held in floats (the floors above), so both addresses are two FMAs and ONE conversion
                     * each -- no conversion of the three floors, no integer multiply, no shift-adds. */
                    const unsigned int s0 = (unsigned int)(int)fmaf(fcx, 32.0f, fmaf(ffo, 8.0f, cpy8f)) & 56u;
                    const unsigned int cb = (unsigned int)(int)fmaf(fcy, (float)(DESC_RS * 8), fmaf(fcx, 64.0f, hbasef));
                    const unsigned int e0 = cb + s0;
                    const unsigned int e1 = (s0 ^ 32u) + cb;
                    /* A cell that does not exist (column / row -1 or 4) is skipped by the exec mask of its atomic -- the four
                     * compares below ARE the existence tests; its weight is not zeroed first, and no weight is tested against
                     * zero (an add of zero is harmless): four compares per sample where there were eight and four selects. */
                    const bool x0 = tu >= 0.0f, x1 = tu < 3.0f, y0 = tv >= 0.0f, y1 = tv < 3.0f;
/* ISA: atomics */
/* round(w * wgt) for both halves of the word by the float trick: w * wgt + 2^23 holds the integer in its mantissa (the
 * products stay below 2^23: make_desc_rec keeps fbits <= 14), one v_and strips the exponent -- an FMA and an AND, both
 * full rate, where v_cvt_u32_f32 runs at half rate */
/* timing probes (results invalid): DESC_PROBE_NOATOMIC folds the words into a register instead of adding them to the LDS
 * histogram, DESC_PROBE_NOLOAD takes the taps from registers instead of memory (tools/build_variants.py) */
#ifdef DESC_PROBE_NOATOMIC
#define PS_CELL_ADD(ADDR, LO, HI) probe_acc ^= (LO) ^ (HI) ^ (ADDR);
#else
#define PS_CELL_ADD(ADDR, LO, HI)                                                                             \
    __hip_atomic_fetch_add((lds_fix64*)(size_t)(ADDR), ((fix64)(HI) << 32) | (fix64)(LO), __ATOMIC_RELAXED,    \
                           __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#define PS_CELL(ADDR, WGT)                                                                                     \
    {                                                                                                          \
        const float        wgt = (WGT);                                                                        \
        const unsigned int lo = __float_as_uint(fmaf(w0, wgt, 8388608.0f)) & 0x7fffffu;                        \
        const unsigned int hi = __float_as_uint(fmaf(do0, wgt, 8388608.0f)) & 0x7fffffu;                       \
        PS_CELL_ADD(ADDR, lo, hi)                                                                              \
    }
                    if (y0) {
                        if (x0) PS_CELL(e0, wy0 * wx0)
                        if (x1) PS_CELL(e1 + 64u, wy0 * fx)
                    }
                    if (y1) {
                        if (x0) PS_CELL(e0 + DESC_RS * 8u, wy1 * wx0)
                        if (x1) PS_CELL(e1 + DESC_RS * 8u + 64u, wy1 * fx)
                    }
#undef PS_CELL
#undef PS_CELL_ADD
                }
            };
            /* ISA: control */
            /* two register sets take turns as "being binned" and "in flight" (the loop is unrolled by two so that no
             * value has to be copied from one role to the other) */
            unsigned int off_a = 0, off_b = 0;
            float u_a = 3.0f, v_a = 3.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            float u_b = 3.0f, v_b = 3.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
            const Taps taps(corner, pitch);
#ifdef DESC_PROBE_NOLOAD
#define PS_TAPS(OFF, A0, A1, A2, A3) { A0 = __uint_as_float(OFF) * 1e-9f; A1 = u0; A2 = v0; A3 = __uint_as_float((OFF) ^ 0x3f800000u); }
#else
#define PS_TAPS(OFF, A0, A1, A2, A3) taps.load_b(OFF, A0, A1, A2, A3);
#endif
            if (loops > 0) {
                coord(ibeg + sub, off_a, u_a, v_a);
                PS_TAPS(off_a, a0, a1, a2, a3)
            }
            for (int t = 0, i = ibeg + sub; t < iters; t += 2, i += 2 * DESC_GL) {
                coord(i + DESC_GL, off_b, u_b, v_b);
                PS_TAPS(off_b, b0, b1, b2, b3)
                bin(u_a, v_a, a0 - a1, a2 - a3);
                coord(i + 2 * DESC_GL, off_a, u_a, v_a);
                PS_TAPS(off_a, a0, a1, a2, a3)
                bin(u_b, v_b, b0 - b1, b2 - b3);
            }
            if (probe_acc == 0x12345u) desc[(size_t)d * 128 + lane] = 1.0f; /* never true: keeps the probe's arithmetic alive */
            wave_lds_sync(); /* ISA: end */
            } /* passes */
        }
        wave_lds_sync();

        /* bin b of cell (iy, ix) = low half of that cell's word for b + high half of its word for b-1 (mod 8),
         * summed over the copies; word positions as in the header comment */
        fix64 a0 = 0ull, a1 = 0ull;
        {
            const int b = lane & 7, ix = (lane >> 3) & 3, iyl = lane >> 5; /* element lane: cell row iyl, lane+64: iyl+2 */
#pragma unroll
            for (int k = 0; k < DESC_COPIES; k++) {
                const int sl = (b + k + ((ix & 1) << 2)) & 7, sh = (b + 7 + k + ((ix & 1) << 2)) & 7;
                const int c0 = k * DESC_CS + iyl * DESC_RS + ix * 8;
                const int c1 = k * DESC_CS + (iyl + 2) * DESC_RS + ix * 8;
                a0 += (hall[c0 + sl] & 0xffffffffull) + (hall[c0 + sh] >> 32);
                a1 += (hall[c1 + sl] & 0xffffffffull) + (hall[c1 + sh] >> 32);
            }
        }
        const float inv_scale = scalbnf(1.0f, -fbits);
        float       v0 = (float)a0 * inv_scale, v1 = (float)a1 * inv_scale;

        /* normalisation (s_desc_norm_rs.h:44-79, s_desc_norm_l2.h:87-134), whole wave.  The reference's RootSift is
         * __fsqrt_rn(__fdividef(v, sum)) and its L2 norms are __fsqrt_rn / __frsqrt_rn: an approximate quotient under
         * correctly rounded roots.  Here the quotient is the hardware reciprocal (1 ulp, like __fdividef's 2 ulp) and the
         * roots are v_sqrt_f32 / v_rsq_f32: 1 ulp where the reference rounds correctly, and a denormal quotient (a bin
         * below 1e-38 of the sum) comes out as 0 -- a divergence of one unit in the last place of single bins, seven orders
         * below the 1e-3 bar, recorded in DESIGN 4; the notile / iloop / grid kernels keep sqrtf and "/".  It replaced 50
         * instructions per descriptor. */
        if (sc.norm_mode == POPSIFT_HIP_NORM_ROOTSIFT) {
            float sum = v0 + v1;
            sum = wave_allsum(sum);
            const float rs = __builtin_amdgcn_rcpf(sum);
            v0 = scalbnf(__builtin_amdgcn_sqrtf(v0 * rs), sc.norm_multi);
            v1 = scalbnf(__builtin_amdgcn_sqrtf(v1 * rs), sc.norm_multi);
        } else {
            float sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            const float norm = __builtin_amdgcn_sqrtf(sq);
            v0 = fminf(v0, 0.2f * norm);
            v1 = fminf(v1, 0.2f * norm);
            sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            const float rn = scalbnf(__builtin_amdgcn_rsqf(sq), sc.norm_multi);
            v0 = v0 * rn;
            v1 = v1 * rn;
        }
        desc[(size_t)d * 128 + lane] = v0;
        desc[(size_t)d * 128 + 64 + lane] = v1;
        wave_lds_sync();
    }
}

/* --------------------------------------------------------- descriptor: grid */

/*
 * DescMode Grid (s_desc_grid.cu:19-147): every cell samples a FIXED 16 x 16 grid of points of its
 * own rotated unit square, each snapped to the nearest pixel, and bins only into its own 8
 * orientation bins.  The reference runs a (16,4,4) block per descriptor: 16 lanes per cell, each
 * looping over 16 rows.  Here one wave owns a descriptor and visits the cells one after the other
 * with all 64 lanes (4 grid rows x 16 columns per step); bins are accumulated in registers with
 * predicated adds and summed over the wave by a butterfly -- no atomics, fixed summation order.
 * The position arithmetic follows the reference operation by operation (the float -> int
 * truncation of "pt + (round(pt + pix) - pt)" included), so the sampled pixels are the oracle's.
 */
__global__ __launch_bounds__(256) void k_descriptor_grid(const PyrDesc* __restrict__ pdp, BatchDesc bd, SiftConsts sc,
                                                         int desc_cap)
{
    const float* __restrict__    arena = bd.s[blockIdx.y].arena;
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const Ext* __restrict__      ext = bd.s[blockIdx.y].ext;
    const int* __restrict__      map = bd.s[blockIdx.y].map;
    const float2* __restrict__   rot = bd.s[blockIdx.y].rot;
    float* __restrict__          desc = bd.s[blockIdx.y].desc;
    __shared__ float s_feat[4][128];
    const int        lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*           feat = s_feat[wave];
    const int        total = min(ct->ori_total, desc_cap);
    const int        L = pdp->L;
    const float      M_4RPI = 4.0f / F_PI;
    const int        xd = lane & 15, ysub = lane >> 4;

    for (XcdSlice sl = xcd_slice<4>(); sl.more(total); sl.s += sl.step) {
        const int d = sl.index();
        if (d >= total) continue;
        const Ext*     e = ext + map[d];
        const float    x = uniformf(e->xpos), y = uniformf(e->ypos), sigma = uniformf(e->sigma);
        const int      ko = min(max(d - e->idx_ori, 0), POPSIFT_HIP_ORI_MAX - 1);
        const float    ang = e->orientation[ko];
        const OctDesc* od = &pdp->o[e->octave];
        const int      width = uniform(od->w), height = uniform(od->h), pitch = uniform(od->pitch);
        const int      lvl = min(max(e->lpos, 0), L - 1);
        const float*   layer = arena + uniform64(od->data_off + lvl * od->plane_stride);
        const float    SBP = fabsf(DESC_MAGNIFY * sigma);

        feat[lane] = 0.0f;
        feat[lane + 64] = 0.0f;
        if (SBP != 0.0f) {
            float sin_t, cos_t;
            cos_t = uniformf(rot[d].x);
            sin_t = uniformf(rot[d].y);
            const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
            const float ldx = -cos_t + sin_t, ldy = -cos_t - sin_t; /* lft_dn  */
            const float rsx = cos_t / 8.0f, rsy = sin_t / 8.0f;     /* rgt_stp */
            const float usx = -sin_t / 8.0f, usy = cos_t / 8.0f;    /* up__stp */
            for (int cell = 0; cell < 16; cell++) {
                const int   ix = cell & 3, iy = cell >> 2;
                const float offx = ix - 1.5f, offy = iy - 1.5f;
                const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
                const float pty = fmaf(csbp, offy, fmaf(ssbp, offx, y));
                float       acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int step = 0; step < 4; step++) {
                    const int yd = step * 4 + ysub;
                    float     pixox = ldx + (xd + 0.5f) * rsx + (yd + 0.5f) * usx;
                    float     pixoy = ldy + (xd + 0.5f) * rsy + (yd + 0.5f) * usy;
                    float     pixx = pixox * SBP, pixy = pixoy * SBP;
                    pixx = roundf(ptx + pixx) - ptx;
                    pixy = roundf(pty + pixy) - pty;
                    pixox = pixx / SBP;
                    pixoy = pixy / SBP;
                    const int    gx = (int)(ptx + pixx), gy = (int)(pty + pixy);
                    /* point texture, clamp addressing */
                    const int    xc = min(max(gx, 0), width - 1), yc = min(max(gy, 0), height - 1);
                    const int    xl = min(max(gx - 1, 0), width - 1), xr = min(max(gx + 1, 0), width - 1);
                    const int    yu = min(max(gy - 1, 0), height - 1), yl = min(max(gy + 1, 0), height - 1);
                    const int    rowc = __mul24(yc, pitch);
                    const float  dxv = plane_at(layer, rowc + xr) - plane_at(layer, rowc + xl);
                    const float  dyv = plane_at(layer, __mul24(yl, pitch) + xc) - plane_at(layer, __mul24(yu, pitch) + xc);
                    const float  mod = __builtin_amdgcn_sqrtf(dxv * dxv + dyv * dyv);
                    float        th = atan2_acc(dyv, dxv);
                    const float  npx = fmaf(cos_t, pixox, sin_t * pixoy);
                    const float  npy = fmaf(cos_t, pixoy, -sin_t * pixox);
                    const float  dnx = npx + offx, dny = npy + offy;
                    const float  ww = __expf(-0.125f * (dnx * dnx + dny * dny));
                    const float  wx_ = 1.0f - fabsf(npx), wy_ = 1.0f - fabsf(npy);
                    const float  wgt = (wx_ < 0.0f || wy_ < 0.0f) ? 0.0f : ww * wx_ * wy_ * mod;
                    th -= ang;
                    th += (th < 0.0f ? F_PI2 : 0.0f);
                    th -= (th >= F_PI2 ? F_PI2 : 0.0f);
                    const float tth = th * M_4RPI;
                    const float ffo = floorf(tth);
                    const float do0 = tth - ffo;
                    const int   b0 = (int)ffo & 7, b1 = (b0 + 1) & 7;
                    const float w1 = do0 * wgt, w0 = wgt - w1;
#pragma unroll
                    for (int b = 0; b < 8; b++) acc[b] += (b == b0 ? w0 : 0.0f) + (b == b1 ? w1 : 0.0f);
                }
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const float v = wave_allsum(acc[b]);
                    if (lane == 0) feat[(cell << 3) + b] = v;
                }
            }
        }
        wave_lds_sync();

        /* normalisation (s_desc_norm_rs.h:44-79, s_desc_norm_l2.h:87-134), whole wave */
        float v0 = feat[lane], v1 = feat[lane + 64];
        if (sc.norm_mode == POPSIFT_HIP_NORM_ROOTSIFT) {
            float sum = v0 + v1;
            sum = wave_allsum(sum);
            v0 = scalbnf(sqrtf(v0 / sum), sc.norm_multi);
            v1 = scalbnf(sqrtf(v1 / sum), sc.norm_multi);
        } else {
            float sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            const float norm = sqrtf(sq);
            v0 = fminf(v0, 0.2f * norm);
            v1 = fminf(v1, 0.2f * norm);
            sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            float rn = 1.0f / sqrtf(sq);
            rn = scalbnf(rn, sc.norm_multi);
            v0 = v0 * rn;
            v1 = v1 * rn;
        }
        desc[(size_t)d * 128 + lane] = v0;
        desc[(size_t)d * 128 + 64 + lane] = v1;
        wave_lds_sync();
    }
}

/* ------------------------------------------------------- descriptor: notile */

/* linear-filter read at pixel-centre coordinates, clamp addressing, 1.8 fixed-point weights
 * (what tex2DLayered on the reference's linear texture returns, common/assist.h:66-81) */
__device__ __forceinline__ float tex_linear(const float* pl, int w, int h, int pitch, float x, float y)
{
    const float fx = floorf(x), fy = floorf(y);
    float       a = x - fx, b = y - fy;
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int    i = (int)fx, j = (int)fy;
    const int    x0 = min(max(i, 0), w - 1), x1 = min(max(i + 1, 0), w - 1);
    const int    r0 = __mul24(min(max(j, 0), h - 1), pitch);
    const int    r1 = __mul24(min(max(j + 1, 0), h - 1), pitch);
    const float  top = (1.0f - a) * plane_at(pl, r0 + x0) + a * plane_at(pl, r0 + x1);
    const float  bot = (1.0f - a) * plane_at(pl, r1 + x0) + a * plane_at(pl, r1 + x1);
    return (1.0f - b) * top + b * bot;
}

/*
 * DescMode NoTile (s_desc_notile.cu:28-128): a 40 x 40 grid of sample points at 1/8-cell pitch in
 * the keypoint's rotated frame; at each point the gradient is taken along the rotated axes from
 * four bilinearly interpolated reads (s_gradiant.h:71-87), weighted by a Gaussian table
 * (desc_gauss, sift_constants.cu:33-41) and by tent weights over the 16 x 16 points of every cell
 * that covers it (desc_tile, :43-46).  The reference evaluates each point once per covering
 * cell (up to four times); here one wave owns the descriptor, evaluates each of the 1600 points
 * once and spreads it to its <= 2 x 2 cells with the packed fixed-point LDS atomics of the loop
 * kernel.
 */
template <bool ILOOP>
__global__ __launch_bounds__(256) void k_descriptor_notile(const PyrDesc* __restrict__ pdp, BatchDesc bd, SiftConsts sc,
                                                           int desc_cap)
{
    const float* __restrict__    arena = bd.s[blockIdx.y].arena;
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const Ext* __restrict__      ext = bd.s[blockIdx.y].ext;
    const int* __restrict__      map = bd.s[blockIdx.y].map;
    const float2* __restrict__   rot = bd.s[blockIdx.y].rot;
    float* __restrict__          desc = bd.s[blockIdx.y].desc;
    constexpr int    DCOPY = 2;
    /* notile: 256 points per cell, iloop: up to 1024, each <= 361 * 1: low halves stay below 2^32 */
    constexpr int    FBITS = ILOOP ? 13 : 14;
    __shared__ fix64 s_hist[4][DCOPY][128];
    const int        lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    fix64*           hist = s_hist[wave][lane & (DCOPY - 1)];
    fix64*           hall = s_hist[wave][0];
    const int        total = min(ct->ori_total, desc_cap);
    const int        L = pdp->L;
    const float      M_4RPI = 4.0f / F_PI;
    const float      stepbase = -2.5f + 1.0f / 16.0f;
    const float      fscale = (float)(1 << FBITS);

    for (XcdSlice sl = xcd_slice<4>(); sl.more(total); sl.s += sl.step) {
        const int d = sl.index();
        if (d >= total) continue;
        const Ext*     e = ext + map[d];
        const float    x = uniformf(e->xpos), y = uniformf(e->ypos), sigma = uniformf(e->sigma);
        const OctDesc* od = &pdp->o[e->octave];
        const int      width = uniform(od->w), height = uniform(od->h), pitch = uniform(od->pitch);
        const int      lvl = min(max(e->lpos, 0), L - 1);
        const float*   layer = arena + uniform64(od->data_off + lvl * od->plane_stride);
        const float    SBP = fabsf(DESC_MAGNIFY * sigma);

#pragma unroll
        for (int k = 0; k < 2 * DCOPY; k++) hall[lane + 64 * k] = 0ull;
        wave_lds_sync();

        if (sigma != 0.0f) {
            float sin_t, cos_t;
            cos_t = uniformf(rot[d].x);
            sin_t = uniformf(rot[d].y);
            if (ILOOP) {
                /* DescMode ILoop (s_desc_iloop.cu:18-133): every cell samples a FIXED 32 x 32 lattice over the
                 * bounding box of its rotated two-cell square and keeps the points inside the square; the
                 * gradient is the interpolated, rotated one of notile, the weights are the loop descriptor's.
                 * The reference gives a cell 32 lanes x 32 steps; a wave here takes two lattice rows a step. */
                const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
                const float bsz = fabsf(cos_t) + fabsf(sin_t);
                const int   j = lane & 31;
                for (int cell = 0; cell < 16; cell++) {
                    const float offx = (float)(cell & 3) - 1.5f, offy = (float)(cell >> 2) - 1.5f;
                    const float ptx = fmaf(csbp, offx, -ssbp * offy);
                    const float pty = fmaf(csbp, offy, ssbp * offx);
                    for (int it = 0; it < 16; it++) {
                        const int   i = 2 * it + (lane >> 5);
                        const float dx = -bsz + (float)j * bsz / 16.0f;
                        const float dy = -bsz + (float)i * bsz / 16.0f;
                        const float nx = fmaf(cos_t, dx, sin_t * dy);
                        const float ny = fmaf(cos_t, dy, -sin_t * dx);
                        const float nnx = fabsf(nx), nny = fabsf(ny);
                        if (nnx < 1.0f && nny < 1.0f) {
                            const float px = x + ptx + dx * SBP, py = y + pty + dy * SBP;
                            const float dxv = tex_linear(layer, width, height, pitch, px + cos_t, py + sin_t) -
                                              tex_linear(layer, width, height, pitch, px - cos_t, py - sin_t);
                            const float dyv = tex_linear(layer, width, height, pitch, px - sin_t, py + cos_t) -
                                              tex_linear(layer, width, height, pitch, px + sin_t, py - cos_t);
                            const float mod = __builtin_amdgcn_sqrtf(dxv * dxv + dyv * dyv);
                            float       th = atan2_acc(dyv, dxv);
                            th += (th < 0.0f ? F_PI2 : 0.0f);
                            th -= (th >= F_PI2 ? F_PI2 : 0.0f);
                            const float tth = th * M_4RPI;
                            const float ffo = floorf(tth);
                            const float do0 = tth - ffo;
                            const int   b0 = (int)ffo & 7;
                            const float dnx = nx + offx, dny = ny + offy;
                            const float ww = __expf(-0.125f * (dnx * dnx + dny * dny));
                            const float wgt = ww * (1.0f - nnx) * (1.0f - nny) * mod * fscale;
                            const float a1 = do0 * wgt, a0 = wgt - a1;
                            atomicAdd(&hist[(cell << 3) + b0],
                                      ((fix64)(unsigned int)(a1 + 0.5f) << 32) | (unsigned int)(a0 + 0.5f));
                        }
                    }
                }
            } else
            for (int p = lane; p < 1600; p += 64) {
                const int   newy = p / 40, newx = p - newy * 40;
                const float stepx = stepbase + 0.125f * (float)newx;
                const float stepy = stepbase + 0.125f * (float)newy;
                const float ptx = cos_t * stepx + -sin_t * stepy;
                const float pty = cos_t * stepy + sin_t * stepx;
                const float px = x + ptx * SBP, py = y + pty * SBP;
                const float dxv = tex_linear(layer, width, height, pitch, px + cos_t, py + sin_t) -
                                  tex_linear(layer, width, height, pitch, px - cos_t, py - sin_t);
                const float dyv = tex_linear(layer, width, height, pitch, px - sin_t, py + cos_t) -
                                  tex_linear(layer, width, height, pitch, px + sin_t, py - cos_t);
                const float mod = __builtin_amdgcn_sqrtf(dxv * dxv + dyv * dyv);
                float       th = atan2_acc(dyv, dxv);
                th += (th < 0.0f ? F_PI2 : 0.0f);
                const float tth = th * M_4RPI;
                const float ffo = floorf(tth);
                const float do0 = tth - ffo;
                const int   b0 = (int)ffo & 7;
                /* desc_gauss[newy][newx]: exp(-(dnx^2 + dny^2) / 8), dn = step position */
                const float ww = __expf(-0.125f * (stepx * stepx + stepy * stepy)) * mod * fscale;
                const float a1 = do0 * ww, a0 = ww - a1;
                /* cells cx with 8cx <= newx <= 8cx+15, tent weight desc_tile[newx - 8cx] */
#pragma unroll
                for (int jy = 0; jy < 2; jy++) {
                    const int cy = (newy >> 3) - jy;
                    if (cy < 0 || cy > 3) continue;
                    const float wy = 1.0f - fabsf(-1.0f + 1.0f / 16.0f + 0.125f * (float)(newy - 8 * cy));
#pragma unroll
                    for (int jx = 0; jx < 2; jx++) {
                        const int cx = (newx >> 3) - jx;
                        if (cx < 0 || cx > 3) continue;
                        const float wx = 1.0f - fabsf(-1.0f + 1.0f / 16.0f + 0.125f * (float)(newx - 8 * cx));
                        const float wgt = wx * wy;
                        const unsigned int lo = (unsigned int)fmaf(a0, wgt, 0.5f);
                        const unsigned int hi = (unsigned int)fmaf(a1, wgt, 0.5f);
                        atomicAdd(&hist[((((cy << 2) + cx) << 3)) + b0], ((fix64)hi << 32) | lo);
                    }
                }
            }
        }
        wave_lds_sync();

        /* bin b of a cell = low half of word b + high half of word b-1 (mod 8), over the copies */
        fix64     a0 = 0ull, a1 = 0ull;
        const int prevw = (lane & ~7) | ((lane + 7) & 7);
#pragma unroll
        for (int k = 0; k < DCOPY; k++) {
            a0 += (hall[k * 128 + lane] & 0xffffffffull) + (hall[k * 128 + prevw] >> 32);
            a1 += (hall[k * 128 + lane + 64] & 0xffffffffull) + (hall[k * 128 + prevw + 64] >> 32);
        }
        float v0 = (float)a0 * (1.0f / (float)(1 << FBITS)), v1 = (float)a1 * (1.0f / (float)(1 << FBITS));

        /* normalisation (s_desc_norm_rs.h:44-79, s_desc_norm_l2.h:87-134), whole wave */
        if (sc.norm_mode == POPSIFT_HIP_NORM_ROOTSIFT) {
            float sum = v0 + v1;
            sum = wave_allsum(sum);
            v0 = scalbnf(sqrtf(v0 / sum), sc.norm_multi);
            v1 = scalbnf(sqrtf(v1 / sum), sc.norm_multi);
        } else {
            float sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            const float norm = sqrtf(sq);
            v0 = fminf(v0, 0.2f * norm);
            v1 = fminf(v1, 0.2f * norm);
            sq = v0 * v0 + v1 * v1;
            sq = wave_allsum(sq);
            float rn = 1.0f / sqrtf(sq);
            rn = scalbnf(rn, sc.norm_multi);
            v0 = v0 * rn;
            v1 = v1 * rn;
        }
        desc[(size_t)d * 128 + lane] = v0;
        desc[(size_t)d * 128 + 64 + lane] = v1;
        wave_lds_sync();
    }
}

/* --------------------------------------------------------------- features */


}  // namespace

hipError_t launch_orientation(const PyrDesc* pd, const BatchDesc& bd, int nb, const SiftConsts& sc, bool filtered, int hist_cap,
                              int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_orientation, dim3(blocks / KP_NW, nb), dim3(64 * KP_NW), 0, s, pd, bd, sc, filtered ? 1 : 0, hist_cap);
    return hipGetLastError();
}

hipError_t launch_scan(const PyrDesc* pd, const BatchDesc& bd, int nb, const SiftConsts& sc, bool filtered, int hist_cap,
                       int n_chunks, int desc_cap, hipStream_t s)
{
    /* n_chunks is the capacity of the lists; a 1080p image fills ~300 chunks */
    hipLaunchKernelGGL(k_scan_local, dim3(std::min(n_chunks * SCAN_SUB, 512 * SCAN_SUB), nb), dim3(SCAN_LT), 0, s, pd, sc, bd,
                       filtered ? 1 : 0, hist_cap);
    hipLaunchKernelGGL(k_scan_apply, dim3(std::min(n_chunks, 1024), nb), dim3(256), 0, s, pd, sc, bd,
                       sc.desc_mode == POPSIFT_HIP_DESC_LOOP ? 1 : 0, desc_cap);
    return hipGetLastError();
}

int scan_chunk() { return SCAN_CHUNK; }
int scan_partials_per_chunk() { return SCAN_SUB; }

hipError_t launch_descriptors(const PyrDesc* pd, const BatchDesc& bd, int nb, const SiftConsts& sc, int desc_cap, int blocks,
                              hipStream_t s)
{
    /* IGrid (s_desc_igrid.cu:20-83) evaluates the same 40 x 40 point lattice with the same weights as NoTile,
     * cell by cell (each point up to four times); the two differ only in summation order (6e-7 relative in the
     * oracle), so both run the one-evaluation-per-point kernel */
    if (sc.desc_mode == POPSIFT_HIP_DESC_NOTILE || sc.desc_mode == POPSIFT_HIP_DESC_IGRID)
        hipLaunchKernelGGL(k_descriptor_notile<false>, dim3(blocks / 4, nb), dim3(256), 0, s, pd, bd, sc, desc_cap);
    else if (sc.desc_mode == POPSIFT_HIP_DESC_ILOOP)
        hipLaunchKernelGGL(k_descriptor_notile<true>, dim3(blocks / 4, nb), dim3(256), 0, s, pd, bd, sc, desc_cap);
    else if (sc.desc_mode == POPSIFT_HIP_DESC_GRID)
        hipLaunchKernelGGL(k_descriptor_grid, dim3(blocks / 4, nb), dim3(256), 0, s, pd, bd, sc, desc_cap);
    else
        hipLaunchKernelGGL(k_descriptor, dim3(blocks / KP_NW, nb), dim3(64 * KP_NW), 0, s, bd, sc, desc_cap);
    return hipGetLastError();
}


}  // namespace popsift_hip
