/*
 * pyramid.hip -- Gaussian scale-space + DoG for gfx950 (wave64, LDS-tiled).
 *
 * Replaces, per level, the reference's separate passes
 *   gauss::normalizedSource::horiz  (s_pyramid_build_ra.cu:17-55)
 *   gauss::absoluteSource::horiz    (s_pyramid_build_aa.cu:17-52)
 *   gauss::absoluteSource::vert     (s_pyramid_build_aa.cu:55-91)
 *   gauss::make_dog                 (s_pyramid_build.cu:74-92)
 *   gauss::get_by_2_pick_every_second (s_pyramid_build.cu:50-71)
 * by ONE kernel per level: stage the source tile (+halo) in LDS, horizontal
 * pass in place in LDS with a register window (4 outputs / lane, ds_read_b128),
 * vertical pass from LDS with a register window (4x4 outputs / lane), write the
 * Gaussian plane and the DoG plane.  No intermediate plane ever reaches HBM.
 *
 * Arithmetic is kept in the reference's order (outermost tap first, explicit
 * fmaf exactly where "out += a * g" appears) and compiled with
 * -ffp-contract=off, so planes are bit-identical to the CPU oracle.  A kernel
 * instantiated for HALO >= span-1 runs shorter filters with zero-padded taps:
 * fmaf(v, 0, out) == out for finite v, so the result does not change.
 */
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sift_types.h"
#include "kernels.h"
#include "blur_common.h"

namespace popsift_hip {

namespace {

constexpr int TW = BLUR_TW; /* tile width  (outputs) */

/* One axis of a CUDA linear-filter fetch at normalised coordinate r
 * (s_image.cu:140-169: normalised coords, clamp, linear, 1.8 fixed-point weight). */
__device__ __forceinline__ void lin_coord(float r, int n, int& i0, float& alpha)
{
    const float xb = r * (float)n - 0.5f;
    const float fl = floorf(xb);
    float       a = xb - fl;
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    i0 = (int)fl;
    alpha = a;
}

template <typename T>
__device__ __forceinline__ float texel(const T* img, int w, int h, int pitch, int x, int y);
template <>
__device__ __forceinline__ float texel<uint8_t>(const uint8_t* img, int w, int h, int pitch, int x, int y)
{
    x = clampi(x, 0, w - 1);
    y = clampi(y, 0, h - 1);
    return (float)img[(size_t)y * pitch + x] / 255.0f;
}
template <>
__device__ __forceinline__ float texel<float>(const float* img, int w, int h, int pitch, int x, int y)
{
    x = clampi(x, 0, w - 1);
    y = clampi(y, 0, h - 1);
    return img[(size_t)y * pitch + x];
}

/*
 * MODE 0: level l >= 1 from plane l-1 (centre tap first in the H pass, DoG out)
 * MODE 1: octave 0 level 0 from the u8 input image (bilinear upscale on the fly)
 * MODE 2: octave 0 level 0 from the f32 input image
 *
 * One LDS buffer of (TH + 2*HALO) rows x (TW + 2*HP) floats:
 *   phase 1  stage the source tile + halo (16 B chunks, every load in flight at once)
 *   phase 2  horizontal pass IN PLACE: a row is owned by 32 lanes of one wave,
 *            which read their 9 x ds_read_b128 windows before the ds_write_b128
 *   phase 3  vertical pass, 4 columns x 4 rows per lane from a register window
 *            of ds_read_b128 rows; 16 B stores of the Gaussian plane and of
 *            DoG = new - old (old kept in registers since phase 2: the same lane
 *            owns the same 4x4 block in both passes)
 */
template <int HALO, int TH>
constexpr int blur_lds_floats()
{
    return (TH + 2 * HALO) * (TW + 2 * ((HALO + 3) & ~3));
}

/* LP lanes per 4x4 output block (NT = LP * TW / 4 * TH / 4 lanes per tile).  LP = 1 is the throughput shape described
 * above.  LP > 1 (plane-to-plane, DoG not stored) is for the small octaves, whose launches are one round of workgroups
 * and take as long as the instruction stream of ONE lane: the staged rows of the horizontal pass are dealt evenly to
 * all NT / 32 half-waves and each block's four output rows of the vertical pass to LP lanes, so a lane executes about
 * 1 / LP of the instructions (the values and their order per output do not change). */
template <int HALO, int MODE, int TH, int NT, int LP = 1>
__device__ __forceinline__ void blur_tile_body(const BlurArgs& a, const BatchDesc& bd, int block, float* __restrict__ s_t)
{
    /* this image's planes (the slot of blockIdx.y) */
    float* const       arena = bd.s[blockIdx.y].arena;
    const float* const a_src = arena + a.src_off;
    float* const       a_dst = arena + a.dst_off;
    float* const       a_dog = a.dog_off >= 0 ? arena + a.dog_off : nullptr;
    float* const       a_next0 = a.next0_off >= 0 ? arena + a.next0_off : nullptr;
    const void* const  a_in = bd.s[blockIdx.y].input;
    static_assert(LP == 1 || (MODE == 0 && (LP == 2 || LP == 4)), "LP > 1: plane-to-plane only");
    constexpr int NB = NT / LP;           /* lanes that own a 4x4 block each   */
    constexpr int HP = (HALO + 3) & ~3;   /* left/right halo, padded to 16 B   */
    constexpr int SW = TW + 2 * HP;       /* LDS row pitch                     */
    constexpr int SR = TH + 2 * HALO;     /* rows staged                       */
    constexpr int NW = 1 + HP / 2;        /* H-pass window in 16 B chunks      */

    const int tile = xcd_remap(block, a.tiles_x * a.tiles_y);
    const int tx0 = (tile % a.tiles_x) * TW;
    const int ty0 = (tile / a.tiles_x) * TH;
    const int tid = threadIdx.x;
    const int w = a.w, h = a.h, pitch = a.pitch;

    /* ---- phase 1: stage source tile (+halo) with clamp addressing -------- */
    if (MODE == 0) {
        constexpr int CH = SW / 4;                 /* chunks per row   */
        constexpr int NCH = SR * CH;               /* chunks per tile  */
        constexpr int NLD = (NCH + NT - 1) / NT;   /* chunks per lane  */
        const float* __restrict__ src = a_src;
        v4f v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int idx = tid + k * NT;
            if (idx < NCH) {
                const int    r = idx / CH, c4 = idx - r * CH;
                const int    gy = clampi(ty0 + r - HALO, 0, h - 1);
                const int    gx0 = tx0 + 4 * c4 - HP;
                const float* row = src + (size_t)gy * pitch;
                if (gx0 >= 0 && gx0 + 3 < w) {
                    v[k] = *reinterpret_cast<const v4f*>(row + gx0);
                } else {
                    v[k].x = row[clampi(gx0 + 0, 0, w - 1)];
                    v[k].y = row[clampi(gx0 + 1, 0, w - 1)];
                    v[k].z = row[clampi(gx0 + 2, 0, w - 1)];
                    v[k].w = row[clampi(gx0 + 3, 0, w - 1)];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int idx = tid + k * NT;
            if (idx < NCH) reinterpret_cast<v4f*>(s_t)[idx] = v[k];
        }
    } else {
        /* s_pyramid_build_ra.cu:38-39: read_x = (x + shift) / dst_w (normalised).
         * Per-column / per-row sample coordinates are computed once per tile. */
        __shared__ int   s_ix[SW];
        __shared__ float s_fa[SW];
        __shared__ int   s_iy[SR];
        __shared__ float s_fb[SR];
        __shared__ float s_lut[256]; /* cudaReadModeNormalizedFloat: v / 255 */
        if (MODE == 1 && tid < 256) s_lut[tid] = (float)tid / 255.0f;
        for (int c = tid; c < SW; c += NT) {
            const int   X = tx0 + c - HP;
            const float read_x = ((float)X + a.shift) / (float)w;
            int         ix;
            float       fa;
            lin_coord(read_x, a.in_w, ix, fa);
            s_ix[c] = ix;
            s_fa[c] = fa;
        }
        for (int r = tid; r < SR; r += NT) {
            const int   Y = clampi(ty0 + r - HALO, 0, h - 1);
            const float read_y = ((float)Y + a.shift) / (float)h;
            int         iy;
            float       fb;
            lin_coord(read_y, a.in_h, iy, fb);
            s_iy[r] = iy;
            s_fb[r] = fb;
        }
        __syncthreads();
        /* Source region touched by this tile: columns [xs0, xs1], rows [ys0, ys1] (sample
         * coordinates are monotone).  When it fits the staging buffer (always for up-scaling), the
         * texels are fetched once into LDS -- every load in flight together -- and the bilinear
         * taps read LDS; otherwise (strong down-sampling) they read global memory directly. */
        constexpr int IN_MAX = (SR * SW) / 2;
        constexpr int IN_LD = (IN_MAX + NT - 1) / NT;
        __shared__ float s_in[IN_MAX];
        const int xs0 = clampi(s_ix[0], 0, a.in_w - 1), xs1 = clampi(s_ix[SW - 1] + 1, 0, a.in_w - 1);
        const int ys0 = clampi(s_iy[0], 0, a.in_h - 1), ys1 = clampi(s_iy[SR - 1] + 1, 0, a.in_h - 1);
        const int RW = xs1 - xs0 + 1, RH = ys1 - ys0 + 1;
        const bool staged = RW * RH <= IN_MAX;
        if (MODE == 1 && a.fast2x == 2) {
            /* Exact 2x upscale of a u8 image whose rows start on 4-byte boundaries (round 3; the general exact-2x path
             * below explains why the weights are {0, 1/2}).  The source window of the tile -- RHC rows of RWP texels,
             * replicated beyond the image edges, which is what the clamped fetch of every out-of-range X / Y gives --
             * is fetched as aligned DWORDS, four texels per lane and load with compile-time row geometry (the byte
             * loads of the general path took a run-time integer division per texel), converted by v_cvt_f32_ubyte +
             * the correctly rounded v / 255 (multiply by RN(1/255), one FMA residual, one FMA correction: equal to the
             * IEEE quotient, i.e. to cudaReadModeNormalizedFloat's table, for all 256 values -- tools/check_div255.py)
             * without the LUT pass through LDS; the expansion reads three texels per 16-byte chunk of U with no clamp
             * or parity arithmetic per element.  U is formed in the same operation order: bit-identical planes. */
            constexpr int RWP = SW / 2 + 4; /* staged texels per row (a multiple of 4)  */
            constexpr int RHC = SR / 2 + 2; /* staged rows                              */
            static_assert(RWP % 4 == 0 && RWP * RHC <= IN_MAX, "the source window fits the staging buffer");
            const int  sx0 = ((tx0 - HP) >> 1) & ~3; /* floor to a multiple of 4 (also for negative values) */
            const int  sy0 = (ty0 - HALO) >> 1;
            const bool inside = sx0 >= 0 && sx0 + RWP <= a.in_w && sy0 >= 0 && sy0 + RHC <= a.in_h;
            const uint8_t* in8 = (const uint8_t*)a_in;
            auto unit = [](float v) -> float { /* v / 255.0f, correctly rounded, for v = 0 .. 255 */
                const float r = 0.003921568859368563f; /* RN(1 / 255) */
                const float q = v * r;
                return fmaf(fmaf(-255.0f, q, v), r, q);
            };
            if (inside) {
                constexpr int DW = RWP / 4, ND = DW * RHC, NL = (ND + NT - 1) / NT;
                unsigned int  pk[NL];
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const int i = tid + k * NT;
                    if (i < ND) {
                        const int r = i / DW, c = i - r * DW;
                        pk[k] = *reinterpret_cast<const unsigned int*>(in8 + (size_t)(sy0 + r) * a.in_pitch + sx0 + 4 * c);
                    }
                }
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const int i = tid + k * NT;
                    if (i < ND) {
                        v4f f;
                        f.x = unit((float)(pk[k] & 255u));
                        f.y = unit((float)((pk[k] >> 8) & 255u));
                        f.z = unit((float)((pk[k] >> 16) & 255u));
                        f.w = unit((float)(pk[k] >> 24));
                        reinterpret_cast<v4f*>(s_in)[i] = f;
                    }
                }
            } else {
                for (int i = tid; i < RWP * RHC; i += NT) {
                    const int r = i / RWP, c = i - r * RWP;
                    const int x = clampi(sx0 + c, 0, a.in_w - 1), y = clampi(sy0 + r, 0, a.in_h - 1);
                    s_in[i] = unit((float)in8[(size_t)y * a.in_pitch + x]);
                }
            }
            __syncthreads();
            constexpr int CH = SW / 4; /* 16-byte chunks per LDS row; a chunk starts at an even X (tx0, HP multiples of 4) */
            for (int idx = tid; idx < SR * CH; idx += NT) {
                const int    r = idx / CH, c4 = idx - r * CH;
                const int    Y = ty0 + r - HALO;
                const float* t = s_in + ((Y >> 1) - sy0) * RWP + (((tx0 + 4 * c4 - HP) >> 1) - sx0);
                const float  a0 = t[0], a1 = t[1], a2 = t[2];
                v4f          out = {a0, 0.5f * a0 + 0.5f * a1, a1, 0.5f * a1 + 0.5f * a2};
                if (Y & 1) {
                    const float b0 = t[RWP], b1 = t[RWP + 1], b2 = t[RWP + 2];
                    const v4f   bot = {b0, 0.5f * b0 + 0.5f * b1, b1, 0.5f * b1 + 0.5f * b2};
                    out = 0.5f * out + 0.5f * bot;
                }
                reinterpret_cast<v4f*>(s_t)[idx] = out;
            }
        } else if (a.fast2x && staged) {
            /* Exact 2x upscale (upscale_factor = +1, PopSift / VLFeat sampling: source coordinate = X / 2): the
             * 1.8 fixed-point weights of the linear filter are exactly 0 or 1/2 (or 1 with the neighbour index, when the
             * coordinate computes an ulp below an integer -- the same texel), so U is a copy / 2-point / 4-point
             * average of the texels, formed in the generic path's operation order: (1-a) t0 + a t1 per row, then per
             * column -- bit-identical, without the per-element coordinate arithmetic and the four LDS gathers.
             * Columns and rows beyond the plane repeat its edge: U(X) = U(clamp X), as the clamped fetch gives. */
            const int n = RW * RH;
            if (MODE == 1) {
                int b[IN_LD];
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) {
                        const int r = i / RW, c = i - r * RW;
                        b[k] = ((const uint8_t*)a_in)[(size_t)(ys0 + r) * a.in_pitch + xs0 + c];
                    }
                }
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) s_in[i] = s_lut[b[k]];
                }
            } else {
                float b[IN_LD];
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) {
                        const int r = i / RW, c = i - r * RW;
                        b[k] = ((const float*)a_in)[(size_t)(ys0 + r) * a.in_pitch + xs0 + c];
                    }
                }
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) s_in[i] = b[k];
                }
            }
            __syncthreads();
            constexpr int CH = SW / 4; /* 16-byte chunks per LDS row; a chunk starts at an even X (tx0, HP multiples of 4) */
            for (int idx = tid; idx < SR * CH; idx += NT) {
                const int r = idx / CH, c4 = idx - r * CH;
                const int Y = clampi(ty0 + r - HALO, 0, h - 1);
                const int j0 = ((Y >> 1) - ys0) * RW, j1 = (min((Y >> 1) + 1, a.in_h - 1) - ys0) * RW;
                const int X0 = tx0 + 4 * c4 - HP; /* even; X0 .. X0 + 3 */
                /* source columns of U(clamp(X0 + q)): i = clamp(X) >> 1, and its right neighbour for odd X */
                int   ia[4], ib[4];
                bool  odd[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int X = clampi(X0 + q, 0, w - 1);
                    ia[q] = (X >> 1) - xs0;
                    ib[q] = min((X >> 1) + 1, a.in_w - 1) - xs0;
                    odd[q] = (X & 1) != 0;
                }
                v4f out;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float t00 = s_in[j0 + ia[q]];
                    float       top = t00;
                    if (odd[q]) top = 0.5f * t00 + 0.5f * s_in[j0 + ib[q]];
                    float val = top;
                    if (Y & 1) {
                        const float t01 = s_in[j1 + ia[q]];
                        float       bot = t01;
                        if (odd[q]) bot = 0.5f * t01 + 0.5f * s_in[j1 + ib[q]];
                        val = 0.5f * top + 0.5f * bot;
                    }
                    out[q] = val;
                }
                reinterpret_cast<v4f*>(s_t)[idx] = out;
            }
        } else if (staged) {
            const int n = RW * RH;
            if (MODE == 1) {
                int b[IN_LD];
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) {
                        const int r = i / RW, c = i - r * RW;
                        b[k] = ((const uint8_t*)a_in)[(size_t)(ys0 + r) * a.in_pitch + xs0 + c];
                    }
                }
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) s_in[i] = s_lut[b[k]];
                }
            } else {
                float b[IN_LD];
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) {
                        const int r = i / RW, c = i - r * RW;
                        b[k] = ((const float*)a_in)[(size_t)(ys0 + r) * a.in_pitch + xs0 + c];
                    }
                }
#pragma unroll
                for (int k = 0; k < IN_LD; k++) {
                    const int i = tid + k * NT;
                    if (i < n) s_in[i] = b[k];
                }
            }
            __syncthreads();
            for (int idx = tid; idx < SR * SW; idx += NT) {
                const int   r = idx / SW, c = idx - r * SW;
                const int   ix = s_ix[c], iy = s_iy[r];
                const float fa = s_fa[c], fb = s_fb[r];
                const int   x0 = clampi(ix, 0, a.in_w - 1) - xs0, x1 = clampi(ix + 1, 0, a.in_w - 1) - xs0;
                const int   y0 = (clampi(iy, 0, a.in_h - 1) - ys0) * RW, y1 = (clampi(iy + 1, 0, a.in_h - 1) - ys0) * RW;
                const float t00 = s_in[y0 + x0], t10 = s_in[y0 + x1], t01 = s_in[y1 + x0], t11 = s_in[y1 + x1];
                const float top = (1.0f - fa) * t00 + fa * t10;
                const float bot = (1.0f - fa) * t01 + fa * t11;
                s_t[idx] = (1.0f - fb) * top + fb * bot;
            }
        } else {
#pragma unroll 4
            for (int idx = tid; idx < SR * SW; idx += NT) {
                const int   r = idx / SW, c = idx - r * SW;
                const int   ix = s_ix[c], iy = s_iy[r];
                const float fa = s_fa[c], fb = s_fb[r];
                const int   x0 = clampi(ix, 0, a.in_w - 1), x1 = clampi(ix + 1, 0, a.in_w - 1);
                const int   y0 = clampi(iy, 0, a.in_h - 1), y1 = clampi(iy + 1, 0, a.in_h - 1);
                float       t00, t10, t01, t11;
                if (MODE == 1) {
                    const uint8_t* r0 = (const uint8_t*)a_in + (size_t)y0 * a.in_pitch;
                    const uint8_t* r1 = (const uint8_t*)a_in + (size_t)y1 * a.in_pitch;
                    const int      b00 = r0[x0], b10 = r0[x1], b01 = r1[x0], b11 = r1[x1];
                    t00 = s_lut[b00];
                    t10 = s_lut[b10];
                    t01 = s_lut[b01];
                    t11 = s_lut[b11];
                } else {
                    const float* r0 = (const float*)a_in + (size_t)y0 * a.in_pitch;
                    const float* r1 = (const float*)a_in + (size_t)y1 * a.in_pitch;
                    t00 = r0[x0];
                    t10 = r0[x1];
                    t01 = r1[x0];
                    t11 = r1[x1];
                }
                const float top = (1.0f - fa) * t00 + fa * t10;
                const float bot = (1.0f - fa) * t01 + fa * t11;
                s_t[idx] = (1.0f - fb) * top + fb * bot;
            }
        }
    }
    __syncthreads();

    /* ---- phase 2: horizontal pass in place, 4 outputs per lane ----------- */
    /* Lane (rg, cx) = (tid / 32, 4 * (tid % 32)) owns the 4x4 output block rows 4rg..4rg+3, columns
     * cx..cx+3 in BOTH passes: while it filters its four centre rows it keeps their source values
     * (the old plane, needed for the DoG) in registers, so the vertical pass does not read the old
     * plane again.  The 2*HALO halo rows are filtered afterwards, spread over the half-waves. */
    constexpr int GROUPS = (TH / 4) / (NB / 32); /* 4-row groups per lane */
    const int     lx = (tid & 31) * 4;           /* first output column of this lane */
    v4f           old[GROUPS][4];
    auto hrow = [&](int r) -> v4f {
        v4f        win[NW];
        const v4f* p = reinterpret_cast<const v4f*>(&s_t[r * SW + lx]);
#pragma unroll
        for (int j = 0; j < NW; j++) win[j] = p[j];
#define PS_W(i) win[(i) >> 2][(i) & 3]
        v4f out;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int cpos = HP + o; /* centre of output o inside the window */
            float     acc;
            if (MODE == 0) {
                acc = PS_W(cpos) * a.taps.g[0];
#pragma unroll
                for (int k = HALO; k > 0; k--) acc = fmaf(PS_W(cpos - k) + PS_W(cpos + k), a.taps.g[k], acc);
            } else {
                acc = 0.0f;
#pragma unroll
                for (int k = HALO; k > 0; k--) acc = fmaf(PS_W(cpos - k) + PS_W(cpos + k), a.taps.g[k], acc);
                acc = fmaf(PS_W(cpos), a.taps.g[0], acc);
                acc = acc * 255.0f;
            }
            out[o] = acc;
        }
#undef PS_W
        /* every lane of this row has issued its reads above (same wave, in order) */
        *reinterpret_cast<v4f*>(&s_t[r * SW + HP + lx]) = out;
        return win[HP / 4]; /* the four source values under this lane's outputs */
    };
    if (LP == 1) {
#pragma unroll
        for (int g = 0; g < GROUPS; g++) {
            const int rg = (tid >> 5) + g * (NT / 32);
#pragma unroll
            for (int j = 0; j < 4; j++) old[g][j] = hrow(HALO + 4 * rg + j);
        }
        for (int hh = tid >> 5; hh < 2 * HALO; hh += NT / 32) (void)hrow(hh < HALO ? hh : TH + hh);
    } else {
        /* no DoG to form, so no lane needs the old values of "its" rows: every staged row to the next free half-wave */
        for (int r = tid >> 5; r < SR; r += NT / 32) (void)hrow(r);
    }
    __syncthreads();

    /* ---- phase 3: vertical pass, 4 columns x 4 rows per lane, + DoG ------ */
    {
        constexpr int NO = 4 / LP;                /* output rows of the block this lane computes */
        constexpr int VWP = NO + 2 * HALO;        /* its window rows */
        const int     bt = tid % NB, o0 = (tid / NB) * NO;
        const int     gx = tx0 + lx;
#pragma unroll
        for (int g = 0; g < GROUPS; g++) {
            const int rg = (bt >> 5) + g * (NB / 32);
            const int r0 = rg * 4; /* first output row of the block (tile-relative) */
            v4f       win[VWP];
#pragma unroll
            for (int j = 0; j < VWP; j++) win[j] = *reinterpret_cast<const v4f*>(&s_t[(r0 + o0 + j) * SW + HP + lx]);
#pragma unroll
            for (int oo = 0; oo < NO; oo++) {
                const int cpos = HALO + oo;
                /* explicit 2-vectors: every FMA of the chain is a v_pk_fma_f32 (left to itself the compiler
                 * goes scalar on the rows whose x / z lanes also feed the next octave: 68 instead of 34) */
                v2f alo = {0.0f, 0.0f}, ahi = {0.0f, 0.0f};
#pragma unroll
                for (int k = HALO; k > 0; k--) {
                    const v2f gk = {a.taps.g[k], a.taps.g[k]};
                    alo = __builtin_elementwise_fma(win[cpos - k].lo, gk, alo);
                    ahi = __builtin_elementwise_fma(win[cpos - k].hi, gk, ahi);
                    alo = __builtin_elementwise_fma(win[cpos + k].lo, gk, alo);
                    ahi = __builtin_elementwise_fma(win[cpos + k].hi, gk, ahi);
                }
                {
                    const v2f g0 = {a.taps.g[0], a.taps.g[0]};
                    alo = __builtin_elementwise_fma(win[cpos].lo, g0, alo);
                    ahi = __builtin_elementwise_fma(win[cpos].hi, g0, ahi);
                }
                const v4f acc = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3);
                const int gy = ty0 + r0 + o0 + oo;
                if (gx < w && gy < h) {
                    /* rows are padded to 64 floats, so a 16 B store at gx < w stays inside the row */
                    /* the Gaussian plane is the next level's input (keep it cached); the DoG plane is not touched again
                     * before the detection kernel: a non-temporal store keeps it from evicting the plane
                     * (measured: level launches -4 %, detection -7 %; non-temporal for both: levels +20 %) */
                    *reinterpret_cast<v4f*>(&a_dst[(size_t)gy * pitch + gx]) = acc;
                    if (LP == 1 && MODE == 0 && a_dog)
                        __builtin_nontemporal_store(acc - old[g][oo], reinterpret_cast<v4f*>(&a_dog[(size_t)gy * pitch + gx]));
                    /* the next octave's level 0 takes pixel (2x, 2y): its width is ceil(w / 2), so 2x <= w - 1 always
                     * and the reference's min(2x, w - 1) never clamps.  gx is a multiple of 4. */
                    if (MODE == 0 && a_next0 && (gy & 1) == 0) {
                        float* q = a_next0 + (size_t)(gy >> 1) * a.next_pitch + (gx >> 1);
                        q[0] = acc.x;
                        if (gx + 2 < w) q[1] = acc.z;
                    }
                }
            }
        }
    }
}

template <int HALO, int MODE, int TH, int NT>
__global__ __launch_bounds__(NT) void k_blur_tile(BlurArgs a, BatchDesc bd)
{
    __shared__ __attribute__((aligned(16))) float s_t[blur_lds_floats<HALO, TH>()];
    if (MODE != 0 && a.zero_words > 0 && blockIdx.x == gridDim.x - 1) {
        int* zero = (int*)bd.s[blockIdx.y].ct;
        for (int i = threadIdx.x; i < a.zero_words; i += NT) zero[i] = 0;
    }
    blur_tile_body<HALO, MODE, TH, NT>(a, bd, blockIdx.x, s_t);
}

/*
 * Two level launches in one: workgroups [0, a.tiles) filter plane a, the rest plane b, both with the HALO-tap
 * instance (a shorter filter runs with zero-padded taps, which does not change its results).  Used for the small
 * octaves, whose launches are latency chains of a few microseconds each: levels L-2 and L-1 of octave o-1 do not
 * depend on octave o, so they ride along with levels 1 and 2 of octave o (s_pyramid_build.cu:549-588 runs the
 * octaves on separate streams for the same reason; one stream and merged launches do it here without the extra
 * streams, which cost throughput when 16 contexts are in flight).
 */
template <int HALO, int LP>
__global__ __launch_bounds__(256 * LP) void k_blur_duo(BlurArgs a, BlurArgs b, BatchDesc bd)
{
    __shared__ __attribute__((aligned(16))) float s_t[blur_lds_floats<HALO, 32>()];
    const int na = a.tiles_x * a.tiles_y;
    if ((int)blockIdx.x < na)
        blur_tile_body<HALO, 0, 32, 256 * LP, LP>(a, bd, blockIdx.x, s_t);
    else
        blur_tile_body<HALO, 0, 32, 256 * LP, LP>(b, bd, blockIdx.x - na, s_t);
}

/* 64-row tiles with LP lanes per 4x4 block (plane-to-plane, DoG not stored) */
template <int HALO, int NT, int LP>
__global__ __launch_bounds__(NT) void k_blur_tile64_lp(BlurArgs a, BatchDesc bd)
{
    __shared__ __attribute__((aligned(16))) float s_t[blur_lds_floats<HALO, 64>()];
    blur_tile_body<HALO, 0, 64, NT, LP>(a, bd, blockIdx.x, s_t);
}

/* one small plane, 32-row tiles, LP lanes per block */
template <int HALO, int LP>
__global__ __launch_bounds__(256 * LP) void k_blur_small(BlurArgs a, BatchDesc bd)
{
    __shared__ __attribute__((aligned(16))) float s_t[blur_lds_floats<HALO, 32>()];
    blur_tile_body<HALO, 0, 32, 256 * LP, LP>(a, bd, blockIdx.x, s_t);
}

/* get_by_2_pick_every_second (s_pyramid_build.cu:50-71) */
template <int MODE, int TH, int NT>
hipError_t launch_blur_mode(const BlurArgs& a, const BatchDesc& bd, int nb, int halo, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, nb), block(NT);
#define PS_CASE(H)                                                                 \
    if (halo <= H) {                                                               \
        hipLaunchKernelGGL((k_blur_tile<H, MODE, TH, NT>), grid, block, 0, s, a, bd);  \
        return hipGetLastError();                                                  \
    }
    PS_CASE(4)
    PS_CASE(5)
    PS_CASE(6)
    PS_CASE(7)
    PS_CASE(8)
    PS_CASE(10)
    PS_CASE(13)
    PS_CASE(16)
    PS_CASE(22)
    PS_CASE(30)
#undef PS_CASE
    return hipErrorInvalidValue;
}

}  // namespace

/* make_dog (s_pyramid_build.cu:74-92) for one plane, on demand: the pipeline itself no longer stores DoG planes */
__global__ void k_dog_plane(float* __restrict__ dog, const float* __restrict__ upper, const float* __restrict__ lower, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dog[i] = upper[i] - lower[i];
}

hipError_t launch_dog_plane(float* dog, const float* upper, const float* lower, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_dog_plane, dim3(2048), dim3(256), 0, s, dog, upper, lower, n);
    return hipGetLastError();
}

int blur_tile_w() { return TW; }

/* 64-row tiles cut the halo re-computation of the horizontal pass (1.4x instead of
 * 1.8x at 27 taps) but need twice the pixels per workgroup: use them where the plane
 * still yields a couple of tiles per CU (measured: +3 % throughput at 1080p) */
int blur_tile_h(int w, int h)
{
    constexpr long min_tiles = 512;
    /* (per image: counting the tiles of all images of a batch -- 64-row tiles for octaves 1 and 2 of a batch of eight 1080p
     * images -- measured 0.7 % slower) */
    const long tiles64 = (long)((w + TW - 1) / TW) * ((h + 63) / 64);
    return tiles64 >= min_tiles ? 64 : 32;
}

/* march kernels from this many pixels per launch (all images of the batch) on, cut into at least this many workgroups */
constexpr long BLUR_MARCH_MIN_PX = 12000000;
constexpr int  BLUR_MARCH_WGS = 768;

/* lanes per 4x4 block for small planes: 1 / 2 / 4 -> pyramid stage of a 1080p image 291 / 295 / 275 us */
constexpr int BLUR_SMALL_LP = 4;
/* planes whose level launch -- for all nb images of the batch -- is one round of workgroups (at most one 128 x 32 tile per
 * CU): latency, not throughput */
bool blur_is_small(int w, int h, int nb) { return (long)((w + TW - 1) / TW) * ((h + 31) / 32) * nb <= 256; }

/* both planes with 32-row tiles and plane-to-plane filtering (mode 0); more lanes per tile when both planes are small
 * and no DoG is stored */
hipError_t launch_blur_duo(const BlurArgs& a, int span_a, const BlurArgs& b, int span_b, const BatchDesc& bd, int nb, hipStream_t s)
{
    const int halo = std::max(span_a, span_b) - 1;
    if (halo < 0 || halo > 30) return hipErrorInvalidValue;
    const bool small = a.dog_off < 0 && b.dog_off < 0 && blur_is_small(a.w, a.h, nb) && blur_is_small(b.w, b.h, nb) && halo <= 16;
    const dim3 grid(a.tiles_x * a.tiles_y + b.tiles_x * b.tiles_y, nb), block(small ? 256 * BLUR_SMALL_LP : 256);
#define PS_CASE(H)                                                                     \
    if (halo <= H) {                                                                   \
        if (small)                                                                     \
            hipLaunchKernelGGL((k_blur_duo<H, BLUR_SMALL_LP>), grid, block, 0, s, a, b, bd); \
        else                                                                           \
            hipLaunchKernelGGL((k_blur_duo<H, 1>), grid, block, 0, s, a, b, bd);            \
        return hipGetLastError();                                                      \
    }
    PS_CASE(4)
    PS_CASE(5)
    PS_CASE(6)
    PS_CASE(7)
    PS_CASE(8)
    PS_CASE(10)
    PS_CASE(13)
    PS_CASE(16)
#undef PS_CASE
#define PS_CASE(H)                                                        \
    if (halo <= H) {                                                      \
        hipLaunchKernelGGL((k_blur_duo<H, 1>), grid, block, 0, s, a, b, bd);   \
        return hipGetLastError();                                         \
    }
    PS_CASE(22)
    PS_CASE(30)
#undef PS_CASE
    return hipErrorInvalidValue;
}

/* one small plane (blur_is_small, 32-row tiles, plane-to-plane, no stored DoG, at most 33 taps) */
static hipError_t launch_blur_small(const BlurArgs& a, const BatchDesc& bd, int nb, int halo, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, nb), block(256 * BLUR_SMALL_LP);
#define PS_CASE(H)                                                                  \
    if (halo <= H) {                                                                \
        hipLaunchKernelGGL((k_blur_small<H, BLUR_SMALL_LP>), grid, block, 0, s, a, bd);  \
        return hipGetLastError();                                                   \
    }
    PS_CASE(4)
    PS_CASE(5)
    PS_CASE(6)
    PS_CASE(7)
    PS_CASE(8)
    PS_CASE(10)
    PS_CASE(13)
    PS_CASE(16)
#undef PS_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_blur(const BlurArgs& a, const BatchDesc& bd, int nb, int mode, int span, int tile_h, hipStream_t s, BlurTune tune)
{
    const int halo = span - 1;
    if (halo < 0 || halo > 30) return hipErrorInvalidValue;
    /* Batches of large planes: the march kernel (blur_march.hip).  It pays where a workgroup makes many steps and the launch
     * still fills the device -- several images per launch (a batch of sixteen 1080p planes: 224-row segments, 4800
     * workgroups of seven steps: 13-19 us per plane against 17-26).  A SINGLE plane has no such operating point (3840 x 2160: 18-27 us
     * against 17-26; 7680 x 4320: config 3's image 2.83-2.86 ms against 2.76 with the tile kernels): DESIGN 6.2. */
    if (mode == 0 && tune.path != 1 && blur_march_supported(a, halo)) {
        const long px = (long)a.w * a.h * nb;
        if (tune.path == 2 || (nb >= 2 && px >= BLUR_MARCH_MIN_PX)) {
            int seg = tune.seg_rows > 0 ? std::max(32, tune.seg_rows / 32 * 32) : blur_march_seg_rows(a.w, a.h, nb, BLUR_MARCH_WGS);
            return launch_blur_march(a, bd, nb, halo, seg, s);
        }
    }
    if (mode == 0 && tile_h == 32 && a.dog_off < 0 && halo <= 16 && blur_is_small(a.w, a.h, nb)) return launch_blur_small(a, bd, nb, halo, s);
    /* the 27-tap level of a large plane: 512 lanes, two per 4x4 block -- every lane filters half the rows of the
     * throughput shape and the tile keeps its LDS footprint (24.9 instead of 27.1 us per 3840 x 2160 launch; 1024 lanes
     * with two or four per block: 27.6 / 28.3 us) */
    if (mode == 0 && tile_h == 64 && a.dog_off < 0 && halo > 10 && halo <= 13) {
        hipLaunchKernelGGL((k_blur_tile64_lp<13, 512, 2>), dim3(a.tiles_x * a.tiles_y, nb), dim3(512), 0, s, a, bd);
        return hipGetLastError();
    }
    constexpr int nt64 = 512;
    /* 512 lanes per 64-row tile halve the serial work per wave at the same LDS footprint (measured
     * -12 % per launch); the 27-tap instance needs ~150 VGPRs for its vertical window and is better
     * off with 256 lanes */
    if (tile_h == 64 && nt64 == 512 && halo <= 10) {
        switch (mode) {
        case 0: return launch_blur_mode<0, 64, 512>(a, bd, nb, halo, s);
        case 1: return launch_blur_mode<1, 64, 512>(a, bd, nb, halo, s);
        case 2: return launch_blur_mode<2, 64, 512>(a, bd, nb, halo, s);
        }
    } else if (tile_h == 64) {
        switch (mode) {
        case 0: return launch_blur_mode<0, 64, 256>(a, bd, nb, halo, s);
        case 1: return launch_blur_mode<1, 64, 256>(a, bd, nb, halo, s);
        case 2: return launch_blur_mode<2, 64, 256>(a, bd, nb, halo, s);
        }
    } else if (tile_h == 32) {
        switch (mode) {
        case 0: return launch_blur_mode<0, 32, 256>(a, bd, nb, halo, s);
        case 1: return launch_blur_mode<1, 32, 256>(a, bd, nb, halo, s);
        case 2: return launch_blur_mode<2, 32, 256>(a, bd, nb, halo, s);
        }
    }
    return hipErrorInvalidValue;
}


}  // namespace popsift_hip
