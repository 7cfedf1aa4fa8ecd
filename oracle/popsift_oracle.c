/*
 * popsift_oracle.c -- CPU restatement of the PopSift (10183308/popsift) SIFT
 * extraction path, used as the parity oracle and as the reported host-core
 * baseline.  TEST INFRASTRUCTURE ONLY (see popsift_oracle.h).
 *
 * PARITY UNPINNED: no golden vectors exist in the reference tree and the CUDA
 * reference cannot be compiled in this image (needs nvcc + Boost); this file
 * follows the reference sources cited at each function.
 *
 * Known, unpinnable differences from a CUDA run of the reference (documented in
 * DESIGN.md "Parity notes"):
 *  - CUDA fast intrinsics (__expf, __sincosf, __fdividef) are replaced by the
 *    IEEE libm functions; nvcc's implicit mul+add contraction is reproduced
 *    only where the source makes the operand pairing unambiguous (the blur
 *    accumulations); elsewhere no contraction is used (-ffp-contract=off).
 *  - shared-memory atomicAdd order in the orientation histogram
 *    (s_orientation.cu:136) is replaced by raster order.
 *  - extrema are emitted in (level, y, x) raster order instead of atomicAdd
 *    arrival order (s_extrema.cu:22-44).
 *  - the last descriptor's racy double normalisation is not
 *    reproduced.
 *
 * Build: gcc -O2 -std=gnu11 -ffp-contract=off -mfma -fopenmp (see Makefile)
 */
#define _GNU_SOURCE
#include "popsift_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GA POPSIFT_HIP_GAUSS_ALIGN
#define MAXL POPSIFT_HIP_MAX_LEVELS
#define MAXO POPSIFT_HIP_MAX_OCTAVES
#define ORI_NBINS 36
#define ORI_WINFACTOR 1.5f
#define DESC_MAGNIFY 3.0f

/* sift_constants.h:22-29: M_PI / M_PI2 are *float* device constants */
static const float F_PI = 3.14159265358979323846f;
static const float F_PI2 = 2.0f * 3.14159265358979323846f;

typedef struct {
    int    w, h;
    float* data[MAXL + 3];
    float* dog[MAXL + 3];
} oct_t;

typedef struct {
    float xpos, ypos;
    int   lpos;
    float sigma;
    int   octave;
    int   cell;
    int   num_ori;
    int   idx_ori;
    float orientation[4];
} ext_t;

struct oracle_ctx {
    popsift_hip_params p;
    int   levels; /* max(2, p.levels), popsift.cpp:71 */
    int   L;      /* levels + 3 Gaussian planes, sift_pyramid.cu:112 */
    float filter[MAXL * GA];
    int   span[MAXL];
    float gsigma[MAXL];
    /* ConstInfo, sift_constants.cu:22-31 */
    float sigma0, sigma_k, edge_limit, threshold;
    int   max_extrema, norm_multi;
    int   threads;

    int   in_w, in_h;
    int   n_oct;
    int   frozen_octaves; /* popsift.cpp:107-111: octave count frozen after the first image */
    oct_t oct[MAXO];
    size_t plane_cap[MAXO];

    ext_t* ext;
    int    ext_cap, ext_total;
    int    ext_ct[MAXO];
    int    ori_total;
    float* desc;     /* normalised */
    float* desc_raw; /* before normalisation */
    int    desc_cap;
};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ tables */

/* gauss_filter.cu:303-328 (vlFeatSpan / openCVSpan) */
static int span_for(int gauss_mode, float sigma)
{
    if (gauss_mode == POPSIFT_HIP_GAUSS_OPENCV_COMPUTE) {
        int span = (int)roundf(2.0f * 4.0f * sigma + 1.0f) | 1;
        span >>= 1;
        span += 1;
        return imin(span, GA - 1);
    }
    return imin((int)(ceilf(4.0f * sigma) + 1.0f), GA - 1);
}

/* gauss_filter.cu:163-181 (inc.sigma) + :340-372 (computeBlurTable) */
static void init_tables(oracle_ctx* c)
{
    const popsift_hip_params* p = &c->p;
    const float sigma0 = p->sigma;
    const int   levels = c->levels;
    const float initial_blur =
        p->assume_initial_blur ? p->initial_blur * powf(2.0f, p->upscale_factor) : 0.0f;

    memset(c->filter, 0, sizeof(c->filter));
    memset(c->gsigma, 0, sizeof(c->gsigma));
    c->gsigma[0] = p->assume_initial_blur
                       ? sqrtf(fabsf(sigma0 * sigma0 - initial_blur * initial_blur))
                       : sigma0;
    for (int lvl = 1; lvl < c->L; lvl++) {
        const float sigmaP = sigma0 * powf(2.0f, (float)(lvl - 1) / (float)levels);
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        c->gsigma[lvl] = sqrtf(sigmaS * sigmaS - sigmaP * sigmaP);
    }
    for (int level = 0; level < MAXL; level++)
        c->span[level] = imin(span_for(p->gauss_mode, c->gsigma[level]), GA - 1);
    for (int level = 0; level < MAXL; level++) {
        const float sig = c->gsigma[level];
        const int   spn = c->span[level];
        float*      f = &c->filter[level * GA];
        double      sum = 1.0;
        f[0] = 1.0f;
        for (int x = 1; x < spn; x++) {
            const float val = (float)exp(-0.5 * (pow((double)x / sig, 2.0)));
            f[x] = val;
            sum += 2.0f * val;
        }
        for (int x = 0; x < spn; x++) f[x] = (float)(f[x] / sum);
        for (int x = spn; x < GA; x++) f[x] = 0.0f;
    }
    /* sift_constants.cu:22-31; threshold = Config::getPeakThreshold, sift_conf.cu:275-278 */
    c->sigma0 = sigma0;
    c->sigma_k = powf(2.0f, 1.0f / levels);
    c->edge_limit = p->edge_limit;
    c->threshold = p->threshold * 0.5f * 255.0f / levels;
    c->max_extrema = p->max_extrema;
    c->norm_multi = p->norm_multi;
}

oracle_ctx* oracle_create(const popsift_hip_params* p)
{
    if (!p) return NULL;
    if (p->sigma > 2.0f) return NULL;                 /* gauss_filter.cu:131-137 */
    if (p->levels > MAXL - 3) return NULL;            /* levels+3 planes must fit 12 table rows */
    if (p->gauss_mode != POPSIFT_HIP_GAUSS_VLFEAT_COMPUTE &&
        p->gauss_mode != POPSIFT_HIP_GAUSS_OPENCV_COMPUTE)
        return NULL;
    if (p->desc_mode < POPSIFT_HIP_DESC_LOOP || p->desc_mode > POPSIFT_HIP_DESC_NOTILE)
        return NULL;
    oracle_ctx* c = (oracle_ctx*)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->p = *p;
    c->levels = imax(2, p->levels);
    c->L = c->levels + 3;
    c->threads = 1;
    c->frozen_octaves = p->octaves;
    init_tables(c);
    return c;
}

void oracle_destroy(oracle_ctx* c)
{
    if (!c) return;
    for (int o = 0; o < MAXO; o++) {
        for (int l = 0; l < MAXL + 3; l++) {
            free(c->oct[o].data[l]);
            free(c->oct[o].dog[l]);
        }
    }
    free(c->ext);
    free(c->desc);
    free(c->desc_raw);
    free(c);
}

void oracle_set_threads(oracle_ctx* c, int threads) { c->threads = threads < 1 ? 1 : threads; }

int oracle_get_gauss_table(const oracle_ctx* c, float* filter, int* span, float* sigma, int* n_levels)
{
    if (!c) return -1;
    if (filter) memcpy(filter, c->filter, sizeof(float) * (size_t)c->L * GA);
    if (span) memcpy(span, c->span, sizeof(int) * (size_t)c->L);
    if (sigma) memcpy(sigma, c->gsigma, sizeof(float) * (size_t)c->L);
    if (n_levels) *n_levels = c->L;
    return 0;
}

/* popsift.cpp:89-120 */
static void plan(const oracle_ctx* c, int w, int h, int octaves_cfg, int* n_oct, int* bw, int* bh)
{
    const float scaleFactor = 1.0f / powf(2.0f, -c->p.upscale_factor);
    int oct = octaves_cfg;
    if (oct < 0)
        oct = imax((int)(floorf(logf((float)imin(w, h)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
    if (oct > MAXO) oct = MAXO;
    *n_oct = oct;
    *bw = (int)ceilf(w * scaleFactor);
    *bh = (int)ceilf(h * scaleFactor);
}

int oracle_plan(const oracle_ctx* c, int w, int h, int* n_octaves, int* base_w, int* base_h)
{
    if (!c || w <= 0 || h <= 0) return -1;
    int n, bw, bh;
    plan(c, w, h, c->p.octaves, &n, &bw, &bh);
    if (n_octaves) *n_octaves = n;
    if (base_w) *base_w = bw;
    if (base_h) *base_h = bh;
    return 0;
}

static int alloc_planes(oracle_ctx* c, int w, int h)
{
    int n, bw, bh;
    plan(c, w, h, c->frozen_octaves, &n, &bw, &bh);
    c->frozen_octaves = n; /* popsift.cpp:111 */
    c->n_oct = n;
    c->in_w = w;
    c->in_h = h;
    int ow = bw, oh = bh;
    for (int o = 0; o < n; o++) {
        oct_t* oc = &c->oct[o];
        oc->w = ow;
        oc->h = oh;
        size_t px = (size_t)ow * oh;
        if (px > c->plane_cap[o]) {
            for (int l = 0; l < c->L; l++) {
                free(oc->data[l]);
                oc->data[l] = (float*)malloc(px * sizeof(float));
                if (!oc->data[l]) return -1;
            }
            for (int l = 0; l < c->L - 1; l++) {
                free(oc->dog[l]);
                oc->dog[l] = (float*)malloc(px * sizeof(float));
                if (!oc->dog[l]) return -1;
            }
            c->plane_cap[o] = px;
        }
        /* sift_pyramid.cu:132-133 */
        ow = (int)ceilf(ow / 2.0f);
        oh = (int)ceilf(oh / 2.0f);
    }
    return 0;
}

/* ---------------------------------------------------------------- pyramid */

typedef struct {
    const uint8_t* u8;
    const float*   f32;
    int            w, h, pitch;
} src_img;

/* texel fetch with cudaAddressModeClamp; u8 is read as normalised float
 * (cudaReadModeNormalizedFloat, s_image.cu:149) */
static inline float texel(const src_img* s, int x, int y)
{
    x = clampi(x, 0, s->w - 1);
    y = clampi(y, 0, s->h - 1);
    if (s->u8) return (float)s->u8[(size_t)y * s->pitch + x] / 255.0f;
    return s->f32[(size_t)y * s->pitch + x];
}

/* One axis of a CUDA linear-filter fetch at normalised coordinate r:
 * xB = r*N - 0.5, i = floor(xB), alpha = frac(xB) in 1.8 fixed point. */
static inline void lin_coord(float r, int n, int* i0, float* alpha)
{
    const float xb = r * (float)n - 0.5f;
    const float fl = floorf(xb);
    float       a = xb - fl;
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    *i0 = (int)fl;
    *alpha = a;
}

/* s_pyramid_build_ra.cu:17-55 (normalizedSource::horiz) + launch params
 * s_pyramid_build.cu:96-126.  Writes the horizontally blurred, x255 plane. */
static void l0_horiz(const oracle_ctx* c, const src_img* s, float* intm, int dw, int dh, float shift)
{
    const int    span = c->span[0];
    const float* filter = &c->filter[0];
#pragma omp parallel for schedule(static) num_threads(c->threads) if (c->threads > 1)
    for (int y = 0; y < dh; y++) {
        const float read_y = ((float)y + shift) / (float)dh;
        int         iy;
        float       b;
        lin_coord(read_y, s->h, &iy, &b);
        /* upscaled row U(x, y) for x in [-(span-1), dw+span-1) */
        const int pad = span;
        float*    U = (float*)malloc(sizeof(float) * (size_t)(dw + 2 * pad));
        for (int xi = -pad; xi < dw + pad; xi++) {
            const float read_x = ((float)xi + shift) / (float)dw;
            int         ix;
            float       a;
            lin_coord(read_x, s->w, &ix, &a);
            const float t00 = texel(s, ix, iy), t10 = texel(s, ix + 1, iy);
            const float t01 = texel(s, ix, iy + 1), t11 = texel(s, ix + 1, iy + 1);
            const float top = (1.0f - a) * t00 + a * t10;
            const float bot = (1.0f - a) * t01 + a * t11;
            U[xi + pad] = (1.0f - b) * top + b * bot;
        }
        for (int x = 0; x < dw; x++) {
            float out = 0.0f;
            for (int offset = span - 1; offset > 0; offset--) { /* filter[span] == 0 */
                const float g = filter[offset];
                const float v1 = U[x - offset + pad];
                const float v2 = U[x + offset + pad];
                out = fmaf(v1 + v2, g, out);
            }
            out = fmaf(U[x + pad], filter[0], out);
            intm[(size_t)y * dw + x] = out * 255.0f;
        }
        free(U);
    }
}

/* s_pyramid_build_aa.cu:17-52 (absoluteSource::horiz): centre tap first, then
 * offsets span-1 .. 1, clamp addressing. */
static void horiz(const oracle_ctx* c, const float* src, float* dst, int w, int h, int level)
{
    const int    span = c->span[level];
    const float* filter = &c->filter[level * GA];
#pragma omp parallel for schedule(static) num_threads(c->threads) if (c->threads > 1)
    for (int y = 0; y < h; y++) {
        const float* row = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float out = row[x] * filter[0];
            for (int offset = span - 1; offset > 0; offset--) {
                const float D = row[clampi(x - offset, 0, w - 1)];
                const float E = row[clampi(x + offset, 0, w - 1)];
                out = fmaf(D + E, filter[offset], out);
            }
            dst[(size_t)y * w + x] = out;
        }
    }
}

/* s_pyramid_build_aa.cu:55-91 (absoluteSource::vert): outermost tap first,
 * upper then lower sample as two separate FMAs, centre last. */
static void vert(const oracle_ctx* c, const float* src, float* dst, int w, int h, int level)
{
    const int    span = c->span[level];
    const float* filter = &c->filter[level * GA];
#pragma omp parallel for schedule(static) num_threads(c->threads) if (c->threads > 1)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float out = 0.0f;
            for (int offset = span - 1; offset > 0; offset--) {
                const float g = filter[offset];
                out = fmaf(src[(size_t)clampi(y - offset, 0, h - 1) * w + x], g, out);
                out = fmaf(src[(size_t)clampi(y + offset, 0, h - 1) * w + x], g, out);
            }
            out = fmaf(src[(size_t)y * w + x], filter[0], out);
            dst[(size_t)y * w + x] = out;
        }
    }
}

/* Pyramid::build_pyramid default branch, s_pyramid_build.cu:549-588 */
static int build_pyramid(oracle_ctx* c, const src_img* s)
{
    if (alloc_planes(c, s->w, s->h)) return -1;
    const int mode = c->p.sift_mode;
    size_t    maxpx = (size_t)c->oct[0].w * c->oct[0].h;
    float*    intm = (float*)malloc(maxpx * sizeof(float));
    if (!intm) return -1;
    for (int o = 0; o < c->n_oct; o++) {
        oct_t* oc = &c->oct[o];
        for (int level = 0; level < c->L; level++) {
            if (level == 0) {
                if (o == 0) {
                    /* s_pyramid_build.cu:109-114 */
                    float shift = 0.5f;
                    if (mode == POPSIFT_HIP_SIFT_POPSIFT || mode == POPSIFT_HIP_SIFT_VLFEAT)
                        shift = 0.5f * powf(2.0f, c->p.upscale_factor - 0);
                    l0_horiz(c, s, intm, oc->w, oc->h, shift);
                    vert(c, intm, oc->data[0], oc->w, oc->h, 0);
                } else {
                    /* get_by_2_pick_every_second, s_pyramid_build.cu:50-71; PREV_LEVEL 3 */
                    const oct_t* pv = &c->oct[o - 1];
                    const float* src = pv->data[c->L - 3];
                    for (int y = 0; y < oc->h; y++)
                        for (int x = 0; x < oc->w; x++)
                            oc->data[0][(size_t)y * oc->w + x] =
                                src[(size_t)imin(y << 1, pv->h - 1) * pv->w + imin(x << 1, pv->w - 1)];
                }
            } else {
                horiz(c, oc->data[level - 1], intm, oc->w, oc->h, level);
                vert(c, intm, oc->data[level], oc->w, oc->h, level);
            }
        }
        /* make_dog, s_pyramid_build.cu:74-92 */
        for (int l = 0; l < c->L - 1; l++) {
            const size_t n = (size_t)oc->w * oc->h;
            const float *a = oc->data[l], *b = oc->data[l + 1];
            float*       d = oc->dog[l];
            for (size_t i = 0; i < n; i++) d[i] = b[i] - a[i];
        }
    }
    free(intm);
    return 0;
}

/* ---------------------------------------------------------------- extrema */

/* s_solve.h:24-85 */
static int solve3(float i[3][3], float b[3])
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

    if (det == 0) return 0;

    float rsd = 1.0f / det; /* __frcp_rn */

    i[0][0] = det0 * rsd;
    i[1][0] = det1 * rsd;
    i[2][0] = det2 * rsd;
    i[1][1] = det3 * rsd;
    i[1][2] = det4 * rsd;
    i[2][2] = det5 * rsd;
    i[0][1] = i[1][0];
    i[0][2] = i[2][0];
    i[2][1] = i[1][2];

    float vout[3];
    vout[0] = vout[1] = vout[2] = 0;
    for (int y = 0; y < 3; y++) {
        vout[y] += (i[y][0] * b[0]);
        vout[y] += (i[y][1] * b[1]);
        vout[y] += (i[y][2] * b[2]);
    }
    b[0] = vout[0];
    b[1] = vout[1];
    b[2] = vout[2];
    return 1;
}

int oracle_solve3(float A[9], float b[3])
{
    float m[3][3];
    memcpy(m, A, sizeof(m));
    int ok = solve3(m, b);
    memcpy(A, m, sizeof(m));
    return ok;
}

/* point-texture read of the layered DoG array: clamp in x, y (sift_octave.cu
 * tex_desc.addressMode) and clamp of the layer index */
static inline float dogv(const oracle_ctx* c, const oct_t* oc, int x, int y, int z)
{
    x = clampi(x, 0, oc->w - 1);
    y = clampi(y, 0, oc->h - 1);
    z = clampi(z, 0, c->L - 2);
    return oc->dog[z][(size_t)y * oc->w + x];
}

/* s_extrema.cu:56-120: strict max or strict min over the 26 neighbours */
static int is_extremum(const oracle_ctx* c, const oct_t* oc, int x, int y, int z)
{
    const float val = dogv(c, oc, x, y, z);
    int         gt = 1, lt = 1;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy && !dz) continue;
                const float f = dogv(c, oc, x + dx, y + dy, z + dz);
                gt &= (val > f);
                lt &= (val < f);
            }
    return gt || lt;
}

static inline int f2i_sat(float f)
{
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* s_extrema.cu:300-504 find_extrema_in_dog_sub + ModeFunctions :145-298 */
static int find_one(const oracle_ctx* c, const oct_t* oc, int octave, int x, int y, int level, ext_t* ec)
{
    const int mode = c->p.sift_mode;
    const int width = oc->w, height = oc->h;
    const int maxlevel = c->L - 1; /* find_extrema passes _levels-1, s_extrema.cu:608 */

    if (mode == POPSIFT_HIP_SIFT_OPENCV) {
        if (x < 5 || y < 5 || x >= width - 5 || y >= height - 5) return 0;
    }
    const float val = dogv(c, oc, x, y, level);

    /* first_contrast_ok */
    if (mode == POPSIFT_HIP_SIFT_OPENCV) {
        if (!(fabsf(val) >= floorf(c->threshold))) return 0;
    } else if (mode == POPSIFT_HIP_SIFT_VLFEAT) {
        if (!(fabsf(val) >= 0.8f * 2.0f * c->threshold)) return 0;
    } else {
        if (!(fabsf(val) >= 1.6f * c->threshold)) return 0;
    }
    if (!is_extremum(c, oc, x, y, level)) return 0;

    float     Dx = 0, Dy = 0, Dz = 0, DDx = 0, DDy = 0, DDz = 0, DXx = 0, DXy = 0, DXz = 0;
    float     d[3] = {0, 0, 0};
    const float v = val;
    int       nx = x, ny = y, nz = level;
    int       iter = 0;
    const int MAX_ITERATIONS = 5;

#define R(dx, dy, dz) dogv(c, oc, nx + (dx), ny + (dy), nz + (dz))
    do {
        iter++;
        const float x2y1z1 = R(1, 0, 0), x0y1z1 = R(-1, 0, 0);
        const float x1y2z1 = R(0, 1, 0), x1y0z1 = R(0, -1, 0);
        const float x1y1z2 = R(0, 0, 1), x1y1z0 = R(0, 0, -1);
        Dx = scalbnf(x2y1z1 - x0y1z1, -1);
        Dy = scalbnf(x1y2z1 - x1y0z1, -1);
        Dz = scalbnf(x1y1z2 - x1y1z0, -1);

        const float x1y1z1 = R(0, 0, 0);
        DDx = x2y1z1 + x0y1z1 - scalbnf(x1y1z1, 1);
        DDy = x1y2z1 + x1y0z1 - scalbnf(x1y1z1, 1);
        DDz = x1y1z2 + x1y1z0 - scalbnf(x1y1z1, 1);

        const float x0y0z1 = R(-1, -1, 0), x0y1z0 = R(-1, 0, -1), x0y1z2 = R(-1, 0, 1);
        const float x0y2z1 = R(-1, 1, 0), x1y0z0 = R(0, -1, -1), x1y0z2 = R(0, -1, 1);
        const float x1y2z0 = R(0, 1, -1), x1y2z2 = R(0, 1, 1), x2y0z1 = R(1, -1, 0);
        const float x2y1z0 = R(1, 0, -1), x2y1z2 = R(1, 0, 1), x2y2z1 = R(1, 1, 0);
        DXx = scalbnf(x2y2z1 + x0y0z1 - x0y2z1 - x2y0z1, -2);
        DXy = scalbnf(x2y1z2 + x0y1z0 - x0y1z2 - x2y1z0, -2);
        DXz = scalbnf(x1y2z2 + x1y0z0 - x1y2z0 - x1y0z2, -2);

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
            d[0] = d[1] = d[2] = 0;
            break;
        }
        d[0] = b[0];
        d[1] = b[1];
        d[2] = b[2];

        const int last_it = (iter == MAX_ITERATIONS);
        int       retval;
        if (mode == POPSIFT_HIP_SIFT_OPENCV) {
            const float tx = fabsf(d[0]), ty = fabsf(d[1]), tz = fabsf(d[2]);
            if (tx < 0.5f && ty < 0.5f && tz < 0.5f) {
                retval = 1;
            } else {
                nx = f2i_sat((float)nx + roundf(d[0]));
                ny = f2i_sat((float)ny + roundf(d[1]));
                nz = f2i_sat((float)nz + roundf(d[2]));
                retval = (nx < 5 || nx >= width - 5 || ny < 5 || ny >= height - 5 || nz < 1 ||
                          nz > maxlevel - 2)
                             ? -1
                             : 0;
            }
        } else if (mode == POPSIFT_HIP_SIFT_VLFEAT) {
            if (last_it) {
                retval = 0;
            } else {
                const float tx = ((d[0] >= 0.6f && nx < width - 2) ? 1.0f : 0.0f) +
                                 ((d[0] <= -0.6f && nx > 1) ? -1.0f : 0.0f);
                const float ty = ((d[1] >= 0.6f && ny < height - 2) ? 1.0f : 0.0f) +
                                 ((d[1] <= -0.6f && ny > 1) ? -1.0f : 0.0f);
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
                const int tx = ((d[0] >= 0.6f && nx < width - 2) ? 1 : 0) +
                               ((d[0] <= -0.6f && nx > 1) ? -1 : 0);
                const int ty = ((d[1] >= 0.6f && ny < height - 2) ? 1 : 0) +
                               ((d[1] <= -0.6f && ny > 1) ? -1 : 0);
                const int tz = ((d[2] >= 0.6f && nz < maxlevel - 1) ? 1 : 0) +
                               ((d[2] <= -0.6f && nz > 1) ? -1 : 0);
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
        if (retval == -1) return 0;
        if (retval == 1) break;
    } while (iter < MAX_ITERATIONS);
#undef R

    if (iter >= MAX_ITERATIONS && mode == POPSIFT_HIP_SIFT_OPENCV) return 0;

    if (mode == POPSIFT_HIP_SIFT_POPSIFT || mode == POPSIFT_HIP_SIFT_VLFEAT) {
        if (d[0] >= 1.5f || d[1] >= 1.5f || d[2] >= 1.5f) return 0;
    }

    const float xn = nx + d[0];
    const float yn = ny + d[1];
    const float sn = nz + d[2];

    if (mode != POPSIFT_HIP_SIFT_OPENCV) {
        if (xn < 0.0f || xn > width - 1.0f || yn < 0.0f || yn > height - 1.0f || sn < 0.0f ||
            sn > maxlevel)
            return 0;
    }

    const float contr = v + scalbnf(Dx * d[0] + Dy * d[1] + Dz * d[2], -1);
    const float tr = DDx + DDy;
    const float det = DDx * DDy - DXx * DXx;
    const float edgeval = tr * tr / det;

    if (det <= 0.0f) return 0;
    if (fabsf(contr) < scalbnf(c->threshold, 1)) return 0;
    if (edgeval >= (c->edge_limit + 1.0f) * (c->edge_limit + 1.0f) / c->edge_limit) return 0;

    const float wdiv = (float)width / c->p.filter_grid_size;  /* sift_octave.cu:39-40 */
    const float hdiv = (float)height / c->p.filter_grid_size;
    ec->xpos = xn;
    ec->ypos = yn;
    ec->lpos = (int)roundf(sn);
    ec->sigma = c->sigma0 * powf(c->sigma_k, sn);
    ec->octave = octave;
    ec->cell = (int)(floorf(yn / hdiv) * c->p.filter_grid_size + floorf(xn / wdiv));
    ec->num_ori = 0;
    ec->idx_ori = 0;
    return 1;
}

static int push_ext(oracle_ctx* c, const ext_t* e)
{
    if (c->ext_total == c->ext_cap) {
        int    ncap = c->ext_cap ? c->ext_cap * 2 : 4096;
        ext_t* n = (ext_t*)realloc(c->ext, sizeof(ext_t) * (size_t)ncap);
        if (!n) return -1;
        c->ext = n;
        c->ext_cap = ncap;
    }
    c->ext[c->ext_total++] = *e;
    return 0;
}

/* Pyramid::find_extrema, s_extrema.cu:565-644: x,y in [1, ...], level in [1, levels];
 * threads past the border read clamped texels and fail the strict test. */
static int find_extrema(oracle_ctx* c)
{
    c->ext_total = 0;
    for (int o = 0; o < c->n_oct; o++) {
        const oct_t* oc = &c->oct[o];
        int          ct = 0;
        for (int level = 1; level <= c->levels; level++) {
            /* rows are independent: gather per row, append in raster order */
            int     nrows = oc->h - 1;
            ext_t** rowbuf = (ext_t**)calloc((size_t)imax(nrows, 1), sizeof(ext_t*));
            int*    rowct = (int*)calloc((size_t)imax(nrows, 1), sizeof(int));
#pragma omp parallel for schedule(dynamic, 8) num_threads(c->threads) if (c->threads > 1)
            for (int y = 1; y < oc->h - 1; y++) {
                int cap = 0;
                for (int x = 1; x < oc->w - 1; x++) {
                    ext_t e;
                    if (find_one(c, oc, o, x, y, level, &e)) {
                        if (rowct[y] == cap) {
                            cap = cap ? cap * 2 : 8;
                            rowbuf[y] = (ext_t*)realloc(rowbuf[y], sizeof(ext_t) * (size_t)cap);
                        }
                        rowbuf[y][rowct[y]++] = e;
                    }
                }
            }
            for (int y = 1; y < oc->h - 1; y++) {
                for (int k = 0; k < rowct[y]; k++) {
                    if (ct < c->max_extrema) { /* s_extrema.cu:541,558 */
                        if (push_ext(c, &rowbuf[y][k])) return -1;
                        ct++;
                    }
                }
                free(rowbuf[y]);
            }
            free(rowbuf);
            free(rowct);
        }
        c->ext_ct[o] = ct;
    }
    for (int o = c->n_oct; o < MAXO; o++) c->ext_ct[o] = 0;
    return 0;
}

/* ------------------------------------------------------------- orientation */

/* s_gradiant.h:55-69 (texture variant; callers stay inside [1,w-2]x[1,h-2]) */
static inline void get_gradiant(float* grad, float* theta, int x, int y, const float* pl, int w, int h)
{
    const int   xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
    const int   ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1);
    const float dx = pl[(size_t)y * w + xp] - pl[(size_t)y * w + xm];
    const float dy = pl[(size_t)yp * w + x] - pl[(size_t)ym * w + x];
    *grad = hypotf(dx, dy);
    *theta = atan2f(dy, dx);
}

/* s_orientation.cu:60-242 ori_par */
/* Warp32::shiftit (warp_bitonic_sort.h:58-70) for all 32 lanes at once: every lane reads its partner's value and index
 * as they were BEFORE the step (the shuffles of a warp execute together) */
static void warp32_shiftit(const float* arr, int* idx, int shift, int direction, int increasing)
{
    int nxt[32];
    for (int t = 0; t < 32; t++) {
        const int   o = t ^ (1 << shift);
        const float my_val = arr[idx[t]];
        const float other_val = arr[idx[o]];
        const int   reverse = (t & (1 << direction)) != 0;
        const int   id_less = (t & (1 << shift)) == 0;
        const int   my_more = id_less ? (my_val > other_val) : (my_val < other_val);
        const int   must_swap = !(my_more ^ reverse ^ increasing);
        nxt[t] = must_swap ? idx[o] : idx[t];
    }
    for (int t = 0; t < 32; t++) idx[t] = nxt[t];
}

/* Warp32::sort64 (warp_bitonic_sort.h:35-56): x[t], y[t] = the int2 of lane t */
static void warp32_sort64(const float* arr, int* x, int* y)
{
    for (int outer = 0; outer < 5; outer++)
        for (int inner = outer; inner >= 0; inner--) {
            warp32_shiftit(arr, x, inner, outer + 1, 0);
            warp32_shiftit(arr, y, inner, outer + 1, 1);
        }
    for (int t = 0; t < 32; t++)
        if (arr[x[t]] < arr[y[t]]) {
            const int m = y[t];
            y[t] = x[t];
            x[t] = m;
        }
    for (int outer = 0; outer < 5; outer++)
        for (int inner = outer; inner >= 0; inner--) {
            warp32_shiftit(arr, x, inner, outer + 1, 0);
            warp32_shiftit(arr, y, inner, outer + 1, 0);
        }
}

/* unit-test entry: the 64 indices after sort64 (x of lanes 0 .. 31, then y of lanes 0 .. 31) */
void oracle_warp32_sort64(const float* yval64, int* out64)
{
    int x[32], y[32];
    for (int t = 0; t < 32; t++) {
        x[t] = t;
        y[t] = t + 32;
    }
    warp32_sort64(yval64, x, y);
    for (int t = 0; t < 32; t++) {
        out64[t] = x[t];
        out64[32 + t] = y[t];
    }
}

static void orientation_one(const oracle_ctx* c, ext_t* e)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    w = oc->w, h = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    float        hist[ORI_NBINS], sm_hist[ORI_NBINS];
    for (int i = 0; i < ORI_NBINS; i++) hist[i] = 0.0f;

    const float x = e->xpos, y = e->ypos, sig = e->sigma;
    const float sigw = ORI_WINFACTOR * sig;
    const int   rad = (int)roundf(3.0f * sigw);
    const float factor = -0.5f / (sigw * sigw);
    const int   sq_thres = rad * rad;

    const int xmin = imax(1, (int)roundf(x) - rad);
    const int xmax = imin(w - 2, (int)roundf(x) + rad);
    const int ymin = imax(1, (int)roundf(y) - rad);
    const int ymax = imin(h - 2, (int)roundf(y) + rad);
    const int wx = xmax - xmin + 1;
    const int hy = ymax - ymin + 1;
    const int loops = wx * hy;

    for (int i = 0; i < loops; i++) {
        const int yy = i / wx + ymin;
        const int xx = i % wx + xmin;
        float     grad, theta;
        get_gradiant(&grad, &theta, xx, yy, layer, w, h);
        const float dx = xx - x;
        const float dy = yy - y;
        const int   sq_dist = (int)(dx * dx + dy * dy);
        if (sq_dist <= sq_thres) {
            const float weight = grad * expf(sq_dist * factor);
            int         bidx = (int)roundf((float)ORI_NBINS * (theta + F_PI) / F_PI2);
            bidx = (bidx == ORI_NBINS) ? 0 : bidx;
            if (bidx >= 0 && bidx < ORI_NBINS) hist[bidx] += weight;
        }
    }

    /* WITH_VLFEAT_SMOOTHING, s_orientation.cu:142-160 */
    for (int i = 0; i < 3; i++) {
        for (int bin = 0; bin < ORI_NBINS; bin++) {
            const int prev = bin == 0 ? ORI_NBINS - 1 : bin - 1;
            const int next = bin == ORI_NBINS - 1 ? 0 : bin + 1;
            sm_hist[bin] = (hist[prev] + hist[bin] + hist[next]) / 3.0f;
        }
        for (int bin = 0; bin < ORI_NBINS; bin++) {
            const int prev = bin == 0 ? ORI_NBINS - 1 : bin - 1;
            const int next = bin == ORI_NBINS - 1 ? 0 : bin + 1;
            hist[bin] = (sm_hist[prev] + sm_hist[bin] + sm_hist[next]) / 3.0f;
        }
    }
    for (int bin = 0; bin < ORI_NBINS; bin++) sm_hist[bin] = hist[bin];

    float refined_angle[64], yval[64];
    for (int bin = 0; bin < 64; bin++) {
        const int prev = bin == 0 ? ORI_NBINS - 1 : bin - 1;
        const int next = bin == ORI_NBINS - 1 ? 0 : bin + 1;
        int       predicate =
            (bin < ORI_NBINS) && (sm_hist[bin] > fmaxf(sm_hist[prev], sm_hist[next]));
        const float num =
            predicate ? 3.0f * sm_hist[prev] - 4.0f * sm_hist[bin] + 1.0f * sm_hist[next] : 0.0f;
        const float denB =
            predicate ? 2.0f * (sm_hist[prev] - 2.0f * sm_hist[bin] + sm_hist[next]) : 1.0f;
        const float newbin = num / denB;
        predicate = (predicate && newbin >= 0.0f && newbin <= 2.0f);
        refined_angle[bin] = predicate ? prev + newbin : -1;
        yval[bin] = predicate ? -(num * num) / (4.0f * denB) + sm_hist[prev] : -INFINITY;
    }

    /* BitonicSort::Warp32<float>::sort64 (warp_bitonic_sort.h:35-78), restated literally: lane t of 32 ends with
     * best_index.x = the index of the t-th largest yval; threads 0 .. 3 use theirs (s_orientation.cu:207-231) */
    int bx[32], by[32];
    for (int t = 0; t < 32; t++) {
        bx[t] = t;
        by[t] = t + 32;
    }
    warp32_sort64(yval, bx, by);
    const int* best = bx;
    const float yval_ref = 0.8f * yval[best[0]];
    int         angles = 0;
    for (int k = 0; k < 4; k++) {
        const float best_val = yval[best[k]];
        if (best_val >= yval_ref) {
            float chosen_bin = refined_angle[best[k]];
            if (chosen_bin >= ORI_NBINS) chosen_bin -= ORI_NBINS;
            e->orientation[angles++] = fmaf(F_PI2 * chosen_bin, 1.0f / ORI_NBINS, -F_PI);
        }
    }
    for (int k = angles; k < 4; k++) e->orientation[k] = 0.0f;
    e->num_ori = angles;
}

/* ------------------------------------------------------------- descriptor */

/* cos / sin of the keypoint orientation (the reference: __sincosf, a fast intrinsic of a few ulp).  Evaluated in
 * double and rounded once, i.e. correctly rounded floats up to a ~1e-8 chance of double rounding: the HIP kernels
 * do the same, so both sides rotate the sampling frame by the SAME two floats -- with the single-precision libm
 * functions host and device differed by an ulp now and then, which the pixel-snapping grid descriptor amplifies. */
static inline void sincos_cr(float ang, float* s, float* c)
{
    *s = (float)sin((double)ang);
    *c = (float)cos((double)ang);
}

/* __fmaf_ru / __fmul_ru (round towards +inf), computed exactly without touching
 * the FP environment: the product of two floats is exact in double; TwoSum
 * gives the exact error of the double addition. */
static inline float fma_up(float a, float b, float c)
{
    const double p = (double)a * (double)b;
    const double s = p + (double)c;
    const double bb = s - p;
    const double err = (p - (s - bb)) + ((double)c - bb);
    float        r = (float)s;
    if ((double)r < s || ((double)r == s && err > 0.0)) r = nextafterf(r, INFINITY);
    return r;
}
static inline float mul_up(float a, float b)
{
    const double p = (double)a * (double)b;
    float        r = (float)p;
    if ((double)r < p) r = nextafterf(r, INFINITY);
    return r;
}

/* s_desc_loop.cu:19-138 ext_desc_loop_sub, block (32,4,4): lane = threadIdx.x */
static void descriptor_one(const oracle_ctx* c, const ext_t* e, float ang, float* features)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    width = oc->w, height = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    const float  x = e->xpos, y = e->ypos, sig = e->sigma;
    const float  SBP = fabsf(DESC_MAGNIFY * sig);
    const float  M_4RPI = 4.0f / F_PI;

    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;

    float cos_t, sin_t;
    sincos_cr(ang, &sin_t, &cos_t);
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
    const float crsbp = cos_t / SBP, srsbp = sin_t / SBP;

    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++) {
            const int   tile = ((iy << 2) + ix) << 3;
            const float offx = ix - 1.5f, offy = iy - 1.5f;
            const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
            const float pty = fmaf(csbp, offy, fmaf(ssbp, offx, y));
            const float bsz = fabsf(csbp) + fabsf(ssbp);
            const int   xmin = imax(1, (int)floorf(ptx - bsz));
            const int   ymin = imax(1, (int)floorf(pty - bsz));
            const int   xmax = imin(width - 2, (int)floorf(ptx + bsz));
            const int   ymax = imin(height - 2, (int)floorf(pty + bsz));
            const int   wx = xmax - xmin + 1;
            const int   hy = ymax - ymin + 1;
            const int   loops = wx * hy;

            float dpt[32][9];
            memset(dpt, 0, sizeof(dpt));
            for (int lane = 0; lane < 32; lane++) {
                for (int i = lane; i < loops; i += 32) {
                    const int   ii = i / wx + ymin;
                    const int   jj = i % wx + xmin;
                    const float dx = jj - ptx, dy = ii - pty;
                    const float nx = fmaf(crsbp, dx, srsbp * dy);
                    const float ny = fmaf(crsbp, dy, -srsbp * dx);
                    const float nnx = fabsf(nx), nny = fabsf(ny);
                    if (nnx < 1.0f && nny < 1.0f) {
                        float mod, th;
                        get_gradiant(&mod, &th, jj, ii, layer, width, height);
                        const float dnx = nx + offx, dny = ny + offy;
                        const float ww = expf(-scalbnf(dnx * dnx + dny * dny, -3));
                        const float wgt = ww * (1.0f - nnx) * (1.0f - nny) * mod;

                        th -= ang;
                        th += (th < 0.0f ? F_PI2 : 0.0f);
                        th -= (th >= F_PI2 ? F_PI2 : 0.0f);

                        const float tth = mul_up(th, M_4RPI);
                        const int   fo0 = (int)floorf(tth);
                        const float do0 = tth - fo0;
                        const float wgt1 = 1.0f - do0;
                        const float wgt2 = do0;
                        int         fo = fo0 % 8;
                        if (fo < 0) fo = 0; /* unreachable for finite input */
                        dpt[lane][fo] = fma_up(wgt1, wgt, dpt[lane][fo]);
                        dpt[lane][fo + 1] = fma_up(wgt2, wgt, dpt[lane][fo + 1]);
                    }
                }
                dpt[lane][0] += dpt[lane][8];
            }
            /* shuffle_down 16, 8, 4, 2, 1 tree; result of lane 0 */
            for (int b = 0; b < 8; b++) {
                float v[32];
                for (int l = 0; l < 32; l++) v[l] = dpt[l][b];
                for (int s = 16; s >= 1; s >>= 1)
                    for (int l = 0; l < s; l++) v[l] += v[l + s];
                features[tile + b] = v[0];
            }
        }
}

/* point-texture read with clamp addressing (sift_octave.cu:233-235) */
static inline float texv(const float* pl, int w, int h, int x, int y)
{
    return pl[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

/* s_desc_grid.cu:19-123 ext_desc_grid_sub, block (16,4,4): 16 lanes (xd) per cell, yd = 0..15.
 * Every cell samples a fixed 16 x 16 grid of points of its own rotated unit square, snapped to
 * the nearest pixel; the gradient comes from the int-coordinate texture overload of
 * get_gradiant (s_gradiant.h:55-69, clamp addressing, no [1,w-2] clipping). */
static void descriptor_grid_one(const oracle_ctx* c, const ext_t* e, float ang, float* features)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    width = oc->w, height = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    const float  x = e->xpos, y = e->ypos, sig = e->sigma;
    const float  SBP = fabsf(DESC_MAGNIFY * sig);
    const float  M_4RPI = 4.0f / F_PI;

    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;

    float cos_t, sin_t;
    sincos_cr(ang, &sin_t, &cos_t);
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;

    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++) {
            const int   tile = ((iy << 2) + ix) << 3;
            const float offx = ix - 1.5f, offy = iy - 1.5f;
            const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
            const float pty = fmaf(csbp, offy, fmaf(ssbp, offx, y));
            const float ldx = -cos_t + sin_t, ldy = -cos_t - sin_t;       /* lft_dn */
            const float rsx = cos_t / 8.0f, rsy = sin_t / 8.0f;           /* rgt_stp */
            const float usx = -sin_t / 8.0f, usy = cos_t / 8.0f;          /* up__stp */
            float       dpt[16][9];
            memset(dpt, 0, sizeof(dpt));
            for (int xd = 0; xd < 16; xd++) {
                for (int yd = 0; yd < 16; yd++) {
                    /* pixo = lft_dn + (xd+0.5)*rgt_stp + (yd+0.5)*up__stp (left to right) */
                    float pixox = ldx + (xd + 0.5f) * rsx + (yd + 0.5f) * usx;
                    float pixoy = ldy + (xd + 0.5f) * rsy + (yd + 0.5f) * usy;
                    float pixx = pixox * SBP, pixy = pixoy * SBP;
                    pixx = roundf(ptx + pixx) - ptx;
                    pixy = roundf(pty + pixy) - pty;
                    pixox = pixx / SBP;
                    pixoy = pixy / SBP;
                    /* get_gradiant( mod, th, (pt+pix).x, (pt+pix).y, ... ): float -> int truncation */
                    const int   gx = (int)(ptx + pixx), gy = (int)(pty + pixy);
                    const float dxv = texv(layer, width, height, gx + 1, gy) - texv(layer, width, height, gx - 1, gy);
                    const float dyv = texv(layer, width, height, gx, gy + 1) - texv(layer, width, height, gx, gy - 1);
                    const float mod = hypotf(dxv, dyv);
                    float       th = atan2f(dyv, dxv);
                    const float npx = fmaf(cos_t, pixox, sin_t * pixoy);
                    const float npy = fmaf(cos_t, pixoy, -sin_t * pixox);
                    const float dnx = npx + offx, dny = npy + offy;
                    const float ww = expf(-scalbnf(dnx * dnx + dny * dny, -3));
                    const float wx_ = 1.0f - fabsf(npx), wy_ = 1.0f - fabsf(npy);
                    if (wx_ < 0.0f || wy_ < 0.0f) continue;
                    const float wgt = ww * wx_ * wy_ * mod;
                    th -= ang;
                    th += (th < 0.0f ? F_PI2 : 0.0f);
                    th -= (th >= F_PI2 ? F_PI2 : 0.0f);
                    const float tth = mul_up(th, M_4RPI);
                    const int   fo0 = (int)floorf(tth);
                    const float do0 = tth - fo0;
                    int         fo = fo0 % 8;
                    if (fo < 0) fo = 0;
                    dpt[xd][fo] = fma_up(1.0f - do0, wgt, dpt[xd][fo]);
                    dpt[xd][fo + 1] = fma_up(do0, wgt, dpt[xd][fo + 1]);
                }
                dpt[xd][0] += dpt[xd][8];
            }
            /* shuffle_down 8, 4, 2, 1 within 16 lanes; lane 0 */
            for (int b = 0; b < 8; b++) {
                float v[16];
                for (int l = 0; l < 16; l++) v[l] = dpt[l][b];
                for (int s = 8; s >= 1; s >>= 1)
                    for (int l = 0; l < s; l++) v[l] += v[l + s];
                features[tile + b] = v[0];
            }
        }
}

/* linear-filter texture read of a plane at pixel-centre coordinates (readTex adds the +0.5 that
 * CUDA's unnormalised linear filter subtracts again, common/assist.h:66-81): bilinear between
 * the four surrounding pixels, clamp addressing, 1.8 fixed-point weights */
static inline float tex_linear(const float* pl, int w, int h, float x, float y)
{
    const float fx = floorf(x), fy = floorf(y);
    float       a = x - fx, b = y - fy;
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int   i = (int)fx, j = (int)fy;
    const float t00 = texv(pl, w, h, i, j), t10 = texv(pl, w, h, i + 1, j);
    const float t01 = texv(pl, w, h, i, j + 1), t11 = texv(pl, w, h, i + 1, j + 1);
    const float top = (1.0f - a) * t00 + a * t10;
    const float bot = (1.0f - a) * t01 + a * t11;
    return (1.0f - b) * top + b * bot;
}

/* s_desc_notile.cu:28-99 ext_desc_notile_sub, block (32,4,1), with the tables of
 * sift_constants.cu:33-47 (desc_gauss[40][40], desc_tile[16]) and the rotated, interpolated
 * gradient of s_gradiant.h:71-87.  Cell (cellx, out_y) is owned by the 8 lanes tx = 8*cellx+in_x;
 * each lane visits 2 x 16 of the cell's 16 x 16 sample points; lanes are summed with
 * shuffle_down 4, 2, 1. */
static void descriptor_notile_one(const oracle_ctx* c, const ext_t* e, float ang, float* features)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    width = oc->w, height = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    const float  x = e->xpos, y = e->ypos;
    const float  SBP = fabsf(DESC_MAGNIFY * e->sigma);
    const float  M_4RPI = 4.0f / F_PI;
    const float  stepbase = -2.5f + 1.0f / 16.0f;

    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (e->sigma == 0) return;
    float cos_t, sin_t;
    sincos_cr(ang, &sin_t, &cos_t);

    float desc_tile[16];
    for (int i = 0; i < 16; i++) desc_tile[i] = 1.0f - fabsf(-1.0f + 1.0f / 16.0f + i * 1.0f / 8.0f);
    const float dn_step = 1.0f / 8.0f, dn_base = 0.5f * dn_step - 20.0f * dn_step;

    for (int out_y = 0; out_y < 4; out_y++)
        for (int cellx = 0; cellx < 4; cellx++) {
            float dpt[8][8];
            memset(dpt, 0, sizeof(dpt));
            for (int in_x = 0; in_x < 8; in_x++) {
                const int tx = 8 * cellx + in_x;
                for (int xoff = 0; xoff < 2; xoff++) {
                    const int xd = (xoff << 3) + in_x;
                    const int newx = (xoff << 3) + tx;
                    for (int yoff = 0; yoff < 2; yoff++)
                        for (int in_y = 0; in_y < 8; in_y++) {
                            const int   yd = (yoff << 3) + in_y;
                            const int   newy = (out_y << 3) + yd;
                            const float wgt = desc_tile[xd] * desc_tile[yd];
                            const float stepx = stepbase + scalbnf((float)newx, -3);
                            const float stepy = stepbase + scalbnf((float)newy, -3);
                            const float ptx = cos_t * stepx + -sin_t * stepy;
                            const float pty = cos_t * stepy + sin_t * stepx;
                            const float px = x + ptx * SBP, py = y + pty * SBP;
                            const float dxv = tex_linear(layer, width, height, px + cos_t, py + sin_t) -
                                              tex_linear(layer, width, height, px - cos_t, py - sin_t);
                            const float dyv = tex_linear(layer, width, height, px - sin_t, py + cos_t) -
                                              tex_linear(layer, width, height, px + sin_t, py - cos_t);
                            const float mod = hypotf(dxv, dyv);
                            float       th = atan2f(dyv, dxv);
                            th += (th < 0.0f ? F_PI2 : 0.0f);
                            const float tth = th * M_4RPI;
                            const int   fo = (int)floorf(th * M_4RPI);
                            const float do0 = tth - fo;
                            const int   fo0 = fo & 7, fo1 = (fo0 + 1) & 7;
                            const float dnx = dn_base + newx * dn_step, dny = dn_base + newy * dn_step;
                            const float ww = expf(-scalbnf(dnx * dnx + dny * dny, -3)) * mod; /* desc_gauss */
                            dpt[in_x][fo0] += (wgt * ((1.0f - do0) * ww));
                            dpt[in_x][fo1] += (wgt * (do0 * ww));
                        }
                }
            }
            for (int b = 0; b < 8; b++) {
                float v[8];
                for (int l = 0; l < 8; l++) v[l] = dpt[l][b];
                for (int s = 4; s >= 1; s >>= 1)
                    for (int l = 0; l < s; l++) v[l] += v[l + s];
                features[((out_y << 2) + cellx) * 8 + b] = v[0];
            }
        }
}

/* rotated, interpolated gradient of s_gradiant.h:71-87 at a real position */
static inline void get_gradiant_rot(float* grad, float* theta, float px, float py, float cos_t, float sin_t,
                                    const float* layer, int width, int height)
{
    const float dxv = tex_linear(layer, width, height, px + cos_t, py + sin_t) -
                      tex_linear(layer, width, height, px - cos_t, py - sin_t);
    const float dyv = tex_linear(layer, width, height, px - sin_t, py + cos_t) -
                      tex_linear(layer, width, height, px + sin_t, py - cos_t);
    *grad = hypotf(dxv, dyv);
    *theta = atan2f(dyv, dxv);
}

/* s_desc_igrid.cu:20-83 ext_desc_igrid_sub, block (16,16,1): the 16 lanes xd of cell (ix, iy) each walk
 * 16 rows yd of the cell's 16 x 16 sample points (the same 40 x 40 point lattice as notile, cell by cell);
 * lanes are summed with shuffle_xor 1, 2, 4, 8. */
static void descriptor_igrid_one(const oracle_ctx* c, const ext_t* e, float ang, float* features)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    width = oc->w, height = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    const float  x = e->xpos, y = e->ypos;
    const float  SBP = fabsf(DESC_MAGNIFY * e->sigma);
    const float  M_4RPI = 4.0f / F_PI;

    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (e->sigma == 0) return;
    float cos_t, sin_t;
    sincos_cr(ang, &sin_t, &cos_t);

    float desc_tile[16];
    for (int i = 0; i < 16; i++) desc_tile[i] = 1.0f - fabsf(-1.0f + 1.0f / 16.0f + i * 1.0f / 8.0f);
    const float dn_step = 1.0f / 8.0f, dn_base = 0.5f * dn_step - 20.0f * dn_step;

    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++) {
            float dpt[16][8];
            memset(dpt, 0, sizeof(dpt));
            for (int xd = 0; xd < 16; xd++)
                for (int yd = 0; yd < 16; yd++) {
                    const float stepx = ix - 2.5f + 1.0f / 16.0f + xd / 8.0f;
                    const float stepy = iy - 2.5f + 1.0f / 16.0f + yd / 8.0f;
                    const float ptx = cos_t * stepx + -sin_t * stepy;
                    const float pty = cos_t * stepy + sin_t * stepx;
                    float       mod, th;
                    get_gradiant_rot(&mod, &th, x + ptx * SBP, y + pty * SBP, cos_t, sin_t, layer, width, height);
                    th += (th < 0.0f ? F_PI2 : 0.0f);
                    th -= (th >= F_PI2 ? F_PI2 : 0.0f);
                    const int   gx = ix * 8 + xd, gy = iy * 8 + yd; /* desc_gauss[gy][gx], sift_constants.cu:33-41 */
                    const float dnx = dn_base + gx * dn_step, dny = dn_base + gy * dn_step;
                    const float ww = expf(-scalbnf(dnx * dnx + dny * dny, -3));
                    const float wgt = ww * desc_tile[xd] * desc_tile[yd] * mod;
                    const float tth = mul_up(th, M_4RPI);
                    const int   fo = (int)floorf(tth);
                    const float do0 = tth - fo;
                    dpt[xd][(fo + 1) & 7] = dpt[xd][(fo + 1) & 7] + wgt * do0;
                    dpt[xd][fo & 7] = dpt[xd][fo & 7] + wgt * (1.0f - do0);
                }
            for (int b = 0; b < 8; b++) {
                float v[16];
                for (int l = 0; l < 16; l++) v[l] = dpt[l][b];
                for (int s = 1; s <= 8; s <<= 1) { /* shuffle_xor butterflies: every lane ends with the total */
                    float t[16];
                    for (int l = 0; l < 16; l++) t[l] = v[l] + v[l ^ s];
                    memcpy(v, t, sizeof(v));
                }
                features[(((iy << 2) + ix) << 3) + b] = v[0];
            }
        }
}

/* s_desc_iloop.cu:18-133 ext_desc_iloop_sub, block (32,1,16): cell (ix, iy) samples a fixed 32 x 32 lattice
 * over the bounding box of its rotated two-cell square (lane j = column, 32 rows i), keeps the points with
 * |n| < 1, interpolated rotated gradient, weights as in the loop descriptor; shuffle_down 16..1. */
static void descriptor_iloop_one(const oracle_ctx* c, const ext_t* e, float ang, float* features)
{
    const oct_t* oc = &c->oct[e->octave];
    const int    width = oc->w, height = oc->h;
    const float* layer = oc->data[clampi(e->lpos, 0, c->L - 1)];
    const float  x = e->xpos, y = e->ypos;
    const float  SBP = fabsf(DESC_MAGNIFY * e->sigma);
    const float  M_4RPI = 4.0f / F_PI;

    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;
    float cos_t, sin_t;
    sincos_cr(ang, &sin_t, &cos_t);
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
    const float bsz = fabsf(cos_t) + fabsf(sin_t);

    for (int iy = 0; iy < 4; iy++)
        for (int ix = 0; ix < 4; ix++) {
            const float offx = ix - 1.5f, offy = iy - 1.5f;
            const float ptx = fmaf(csbp, offx, -ssbp * offy);
            const float pty = fmaf(csbp, offy, ssbp * offx);
            float       dpt[32][9];
            memset(dpt, 0, sizeof(dpt));
            for (int j = 0; j < 32; j++) {
                for (int i = 0; i < 32; i++) {
                    const float dx = (-bsz + j * bsz / 16.0f);
                    const float dy = (-bsz + i * bsz / 16.0f);
                    const float nx = fmaf(cos_t, dx, sin_t * dy);
                    const float ny = fmaf(cos_t, dy, -sin_t * dx);
                    const float nnx = fabsf(nx), nny = fabsf(ny);
                    if (nnx < 1.0f && nny < 1.0f) {
                        const float jj = x + ptx + dx * SBP;
                        const float ii = y + pty + dy * SBP;
                        float       mod, th;
                        get_gradiant_rot(&mod, &th, jj, ii, cos_t, sin_t, layer, width, height);
                        const float dnx = nx + offx, dny = ny + offy;
                        const float ww = expf(-scalbnf(dnx * dnx + dny * dny, -3));
                        const float wgt = ww * (1.0f - nnx) * (1.0f - nny) * mod;
                        th += (th < 0.0f ? F_PI2 : 0.0f);
                        th -= (th >= F_PI2 ? F_PI2 : 0.0f);
                        const float tth = mul_up(th, M_4RPI);
                        const int   fo0 = (int)floorf(tth);
                        const float do0 = tth - fo0;
                        int         fo = fo0 % 8;
                        if (fo < 0) fo = 0; /* unreachable for finite input */
                        dpt[j][fo] = fma_up(1.0f - do0, wgt, dpt[j][fo]);
                        dpt[j][fo + 1] = fma_up(do0, wgt, dpt[j][fo + 1]);
                    }
                }
                dpt[j][0] += dpt[j][8];
            }
            for (int b = 0; b < 8; b++) {
                float v[32];
                for (int l = 0; l < 32; l++) v[l] = dpt[l][b];
                for (int s = 16; s >= 1; s >>= 1)
                    for (int l = 0; l < s; l++) v[l] += v[l + s];
                features[(((iy << 2) + ix) << 3) + b] = v[0];
            }
        }
}

/* s_desc_norm_rs.h:44-79 / s_desc_norm_l2.h:87-134 (32 lanes x float4, tree sums) */
static float tree_sum32(const float* lane)
{
    float v[32];
    memcpy(v, lane, sizeof(v));
    for (int s = 16; s >= 1; s >>= 1)
        for (int l = 0; l < s; l++) v[l] += v[l + s];
    return v[0];
}

void oracle_normalize(float* d, int norm_mode, int norm_multi)
{
    float lane[32];
    if (norm_mode == POPSIFT_HIP_NORM_ROOTSIFT) {
        for (int l = 0; l < 32; l++) lane[l] = d[4 * l] + d[4 * l + 1] + d[4 * l + 2] + d[4 * l + 3];
        const float sum = tree_sum32(lane);
        for (int i = 0; i < 128; i++) d[i] = scalbnf(sqrtf(d[i] / sum), norm_multi);
    } else {
        for (int l = 0; l < 32; l++)
            lane[l] = d[4 * l] * d[4 * l] + d[4 * l + 1] * d[4 * l + 1] + d[4 * l + 2] * d[4 * l + 2] +
                      d[4 * l + 3] * d[4 * l + 3];
        float norm = sqrtf(tree_sum32(lane));
        for (int i = 0; i < 128; i++) d[i] = fminf(d[i], 0.2f * norm);
        for (int l = 0; l < 32; l++)
            lane[l] = d[4 * l] * d[4 * l] + d[4 * l + 1] * d[4 * l + 1] + d[4 * l + 2] * d[4 * l + 2] +
                      d[4 * l + 3] * d[4 * l + 3];
        norm = 1.0f / sqrtf(tree_sum32(lane)); /* __frsqrt_rn */
        norm = scalbnf(norm, norm_multi);
        for (int i = 0; i < 128; i++) d[i] = d[i] * norm;
    }
}

/* ---------------------------------------------------------------- grid filter */

/*
 * Pyramid::extrema_filter_grid, s_filtergrid.cu:109-322 (hook: s_orientation.cu:362-367).
 * Thins the initial extrema of ALL octaves to roughly filter_max_extrema by capping the number kept
 * in each of the grid_size x grid_size image cells:
 *   1. key every extremum by (cell, scale = sigma * 2^octave)           (:56-70)
 *   2. order them by cell, inside a cell by scale descending / ascending, or -- RandomScale --
 *      leave the cell's members in their original (octave, index) order    (:159-196)
 *   3. counts of the NON-EMPTY cells in cell order, padded with zeros to n = grid_size^2 entries
 *      (reduce_by_key compacts, the device_vector is zero-initialised)     (:199-202)
 *   4. host arithmetic on the ascending-sorted counts c[0..n-1]: sumup[i] = c[i]*(n-1-i) + sum(c[0..i]);
 *      ct = #{i : sumup[i] > max}; tailaverage = mean of the ct largest counts;
 *      newlimit = ceil(tailaverage - (ext_total - max) / ct)  -- the quotient is an INTEGER division
 *      (:247-256); every count is clamped to newlimit
 *   5. in every cell the members beyond its clamped count are dropped       (:266-280)
 *   6. the survivors keep their relative order inside their octave           (:286-314)
 * Thrust's merge sort is stable, so members of a cell with equal scale stay in original order; the
 * same is stated here with an index tie-break.  The original order itself is atomicAdd arrival order in
 * the reference and raster order here, so RandomScale agrees with the reference only in the NUMBER
 * kept per cell, not in the members.
 */
typedef struct {
    int   cell;
    float scale;
    int   idx;
} fkey_t;

static int g_filter_mode; /* comparator context (the oracle filters one context at a time) */

static int fkey_cmp(const void* a, const void* b)
{
    const fkey_t* l = (const fkey_t*)a;
    const fkey_t* r = (const fkey_t*)b;
    if (l->cell != r->cell) return l->cell < r->cell ? -1 : 1;
    if (g_filter_mode == POPSIFT_HIP_FILTER_LARGEST_FIRST && l->scale != r->scale) return l->scale > r->scale ? -1 : 1;
    if (g_filter_mode == POPSIFT_HIP_FILTER_SMALLEST_FIRST && l->scale != r->scale) return l->scale < r->scale ? -1 : 1;
    return l->idx < r->idx ? -1 : (l->idx > r->idx ? 1 : 0);
}

static int int_cmp(const void* a, const void* b)
{
    const int l = *(const int*)a, r = *(const int*)b;
    return l < r ? -1 : (l > r ? 1 : 0);
}

/* keep[i] = 1 if extremum i survives.  Returns the per-cell limit ("newlimit"), or -1 on error. */
int oracle_filter_grid_keys(const int* cell, const float* scale, int n_ext, int grid_size, int filter_max,
                            int mode, unsigned char* keep)
{
    const int n = grid_size * grid_size;
    if (n_ext <= 0 || n <= 0) return -1;
    fkey_t* k = (fkey_t*)malloc(sizeof(fkey_t) * (size_t)n_ext);
    int*    counts = (int*)calloc((size_t)(n > n_ext ? n : n_ext) + 1, sizeof(int));
    int*    start = (int*)calloc((size_t)n_ext + 1, sizeof(int));
    int*    sorted = (int*)calloc((size_t)n, sizeof(int));
    if (!k || !counts || !start || !sorted) return -1;
    for (int i = 0; i < n_ext; i++) {
        k[i].cell = cell[i];
        k[i].scale = scale[i];
        k[i].idx = i;
    }
    g_filter_mode = mode;
    qsort(k, (size_t)n_ext, sizeof(fkey_t), fkey_cmp);

    /* run lengths of equal cell values, in order (reduce_by_key) */
    int groups = 0;
    for (int i = 0; i < n_ext; i++) {
        if (i == 0 || k[i].cell != k[i - 1].cell) {
            start[groups] = i;
            counts[groups++] = 0;
        }
        counts[groups - 1]++;
    }
    start[groups] = n_ext;
    /* the reference's count vector has exactly n entries; more distinct cell values than that would
     * overrun it there -- they cannot occur for positions inside the image */
    for (int i = 0; i < n; i++) sorted[i] = i < groups ? counts[i] : 0;
    qsort(sorted, (size_t)n, sizeof(int), int_cmp);

    int ct = 0, prefix = 0;
    for (int i = 0; i < n; i++) {
        prefix += sorted[i];
        const int sumup = sorted[i] * (n - 1 - i) + prefix;
        if (sumup > filter_max) ct++;
    }
    int newlimit;
    if (ct == 0) {
        newlimit = 0x7fffffff; /* cannot happen when the caller's 10 % test passed; keep everything */
    } else {
        int tail = 0;
        for (int i = n - ct; i < n; i++) tail += sorted[i];
        const float tailaverage = (float)tail / ct;
        newlimit = (int)ceilf(tailaverage - (float)((n_ext - filter_max) / ct));
    }
    memset(keep, 0, (size_t)n_ext);
    for (int g = 0; g < groups; g++) {
        const int kept = counts[g] < newlimit ? counts[g] : newlimit;
        for (int j = 0; j < kept; j++) keep[k[start[g] + j].idx] = 1;
    }
    free(k);
    free(counts);
    free(start);
    free(sorted);
    return newlimit;
}

/* the hook of Pyramid::orientation, s_orientation.cu:353-367 */
static int filter_grid(oracle_ctx* c)
{
    const int fmax = c->p.filter_max_extrema;
    const int n = c->ext_total;
    if (!(fmax > 0 && (int)(fmax * 1.1) < n)) return 0;
    int*           cell = (int*)malloc(sizeof(int) * (size_t)n);
    float*         scale = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* keep = (unsigned char*)malloc((size_t)n);
    if (!cell || !scale || !keep) return -1;
    for (int i = 0; i < n; i++) {
        cell[i] = c->ext[i].cell;
        scale[i] = c->ext[i].sigma * powf(2.0f, (float)c->ext[i].octave); /* s_filtergrid.cu:68 */
    }
    const int lim = oracle_filter_grid_keys(cell, scale, n, c->p.filter_grid_size, fmax, c->p.filter_sorting, keep);
    if (lim < 0) return -1;
    int out = 0;
    for (int o = 0; o < MAXO; o++) c->ext_ct[o] = 0;
    for (int i = 0; i < n; i++) {
        if (keep[i]) {
            c->ext_ct[c->ext[i].octave]++;
            c->ext[out++] = c->ext[i];
        }
    }
    c->ext_total = out;
    free(cell);
    free(scale);
    free(keep);
    return 0;
}

/* ------------------------------------------------------------------- matching */

/* l2_in_t0, features.cu:157-176: lane t of a 32-thread block squares and sums float4 number t of the
 * difference (nvcc contracts x*x + y*y + z*z + w*w into an FMA chain), then shuffle_down(16,8,4,2,1)
 * adds the 32 lane values; lane 0 holds the result. */
static float l2_in_t0(const float* l, const float* r)
{
    float lane[32];
    for (int t = 0; t < 32; t++) {
        const float x = l[4 * t] - r[4 * t], y = l[4 * t + 1] - r[4 * t + 1];
        const float z = l[4 * t + 2] - r[4 * t + 2], w = l[4 * t + 3] - r[4 * t + 3];
        lane[t] = fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x)));
    }
    for (int s = 16; s >= 1; s >>= 1)
        for (int t = 0; t < s; t++) lane[t] = lane[t] + lane[t + s]; /* only the lanes that reach lane 0 */
    return lane[0];
}

/* compute_distance, features.cu:177-221: out[i] = {best, second, accept, d_best, d_second} */
void oracle_match(const float* l, int l_len, const float* r, int r_len, popsift_hip_match* out, int threads)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads < 1 ? 1 : threads) if (threads > 1)
    for (int idx = 0; idx < l_len; idx++) {
        float match_1st_val = INFINITY, match_2nd_val = INFINITY;
        int   match_1st_idx = 0, match_2nd_idx = 0;
        for (int i = 0; i < r_len; i++) {
            const float res = l2_in_t0(l + 128 * (size_t)idx, r + 128 * (size_t)i);
            if (res < match_1st_val) {
                match_2nd_val = match_1st_val;
                match_2nd_idx = match_1st_idx;
                match_1st_val = res;
                match_1st_idx = i;
            } else if (res < match_2nd_val) {
                match_2nd_val = res;
                match_2nd_idx = i;
            }
        }
        out[idx].best = match_1st_idx;
        out[idx].second = match_2nd_idx;
        out[idx].accept = (match_1st_val / match_2nd_val < 0.8f) ? 1 : 0;
        out[idx].dist_best = match_1st_val;
        out[idx].dist_second = match_2nd_val;
    }
}

/* ------------------------------------------------------------------ driver */

static void all_descriptors(oracle_ctx* c)
{
    const int n = c->ext_total;
#pragma omp parallel for schedule(dynamic, 16) num_threads(c->threads) if (c->threads > 1)
    for (int i = 0; i < n; i++) {
        const ext_t* e = &c->ext[i];
        for (int k = 0; k < e->num_ori; k++) {
            float* raw = c->desc_raw + 128 * (size_t)(e->idx_ori + k);
            float* out = c->desc + 128 * (size_t)(e->idx_ori + k);
            if (c->p.desc_mode == POPSIFT_HIP_DESC_GRID)
                descriptor_grid_one(c, e, e->orientation[k], raw);
            else if (c->p.desc_mode == POPSIFT_HIP_DESC_IGRID)
                descriptor_igrid_one(c, e, e->orientation[k], raw);
            else if (c->p.desc_mode == POPSIFT_HIP_DESC_ILOOP)
                descriptor_iloop_one(c, e, e->orientation[k], raw);
            else if (c->p.desc_mode == POPSIFT_HIP_DESC_NOTILE)
                descriptor_notile_one(c, e, e->orientation[k], raw);
            else
                descriptor_one(c, e, e->orientation[k], raw);
            memcpy(out, raw, 128 * sizeof(float));
            oracle_normalize(out, c->p.norm_mode, c->norm_multi);
        }
    }
}

static int keypoint_stages(oracle_ctx* c)
{
    if (c->n_oct <= 0) return -1;
    if (find_extrema(c)) return -1;
    if (filter_grid(c)) return -1;
    const int n = c->ext_total;
#pragma omp parallel for schedule(dynamic, 16) num_threads(c->threads) if (c->threads > 1)
    for (int i = 0; i < n; i++) orientation_one(c, &c->ext[i]);

    /* ori_prefix_sum, s_orientation.cu:303-345 */
    int total = 0;
    for (int i = 0; i < n; i++) {
        c->ext[i].idx_ori = total;
        total += c->ext[i].num_ori;
    }
    c->ori_total = total;
    if (total > c->desc_cap) {
        free(c->desc);
        free(c->desc_raw);
        c->desc_cap = total + 1024;
        c->desc = (float*)malloc(sizeof(float) * 128 * (size_t)c->desc_cap);
        c->desc_raw = (float*)malloc(sizeof(float) * 128 * (size_t)c->desc_cap);
        if (!c->desc || !c->desc_raw) return -1;
    }
    all_descriptors(c);
    return 0;
}

/* Test hook: the descriptors again, in OTHER frames -- orientations `ori` (4 floats per extremum, in the order of
 * oracle_fetch; NULL: the oracle's own) moved by `ori_ulps` units in the last place, scales `sigma` (one float per extremum,
 * in OCTAVE units like ext_t::sigma; NULL: the oracle's own) moved by `sigma_ulps`.  Extrema, their number of
 * orientations and the order of the descriptors stay as they are.  tests/ use it to take orientation and scale out of a
 * descriptor comparison (the HIP path's values go in: its angles differ from the oracle's in the last bits like two runs
 * of the reference do, its sigma by the device's powf) and to measure how far an ulp of either moves a descriptor. */
int oracle_redo_descriptors(oracle_ctx* c, const float* ori, int ori_ulps, const float* sigma, int sigma_ulps)
{
    if (!c || c->n_oct <= 0) return -1;
    if (c->ext_total == 0 || c->ori_total == 0) return 0; /* nothing to recompute */
    if (!c->desc) return -1;
    for (int i = 0; i < c->ext_total; i++) {
        for (int k = 0; k < c->ext[i].num_ori; k++) {
            float a = ori ? ori[4 * (size_t)i + k] : c->ext[i].orientation[k];
            for (int u = 0; u < (ori_ulps < 0 ? -ori_ulps : ori_ulps); u++) a = nextafterf(a, ori_ulps < 0 ? -INFINITY : INFINITY);
            c->ext[i].orientation[k] = a;
        }
        float sg = sigma ? sigma[i] : c->ext[i].sigma;
        for (int u = 0; u < (sigma_ulps < 0 ? -sigma_ulps : sigma_ulps); u++) sg = nextafterf(sg, sigma_ulps < 0 ? -INFINITY : INFINITY);
        c->ext[i].sigma = sg;
    }
    all_descriptors(c);
    return 0;
}

static int run(oracle_ctx* c, const src_img* s, int keypoints)
{
    if (!c || s->w <= 0 || s->h <= 0 || s->pitch < s->w) return -1;
    if (build_pyramid(c, s)) return -1;
    if (keypoints) return keypoint_stages(c);
    return 0;
}

int oracle_run_u8(oracle_ctx* c, const uint8_t* img, int w, int h, int pitch)
{
    src_img s = {img, NULL, w, h, pitch};
    return run(c, &s, 1);
}
int oracle_run_f32(oracle_ctx* c, const float* img, int w, int h, int pitch)
{
    src_img s = {NULL, img, w, h, pitch};
    return run(c, &s, 1);
}
int oracle_build_pyramid_u8(oracle_ctx* c, const uint8_t* img, int w, int h, int pitch)
{
    src_img s = {img, NULL, w, h, pitch};
    return run(c, &s, 0);
}
int oracle_build_pyramid_f32(oracle_ctx* c, const float* img, int w, int h, int pitch)
{
    src_img s = {NULL, img, w, h, pitch};
    return run(c, &s, 0);
}
int oracle_run_keypoint_stages(oracle_ctx* c) { return keypoint_stages(c); }

int oracle_num_octaves(const oracle_ctx* c) { return c ? c->n_oct : -1; }
int oracle_octave_dims(const oracle_ctx* c, int octave, int* w, int* h)
{
    if (!c || octave < 0 || octave >= c->n_oct) return -1;
    if (w) *w = c->oct[octave].w;
    if (h) *h = c->oct[octave].h;
    return 0;
}
float* oracle_plane_mut(oracle_ctx* c, int octave, int kind, int level)
{
    if (!c || octave < 0 || octave >= c->n_oct || level < 0) return NULL;
    if (kind == 0) return level < c->L ? c->oct[octave].data[level] : NULL;
    if (kind == 1) return level < c->L - 1 ? c->oct[octave].dog[level] : NULL;
    return NULL;
}
const float* oracle_plane(const oracle_ctx* c, int octave, int kind, int level)
{
    return oracle_plane_mut((oracle_ctx*)c, octave, kind, level);
}

int oracle_counts(const oracle_ctx* c, int* n_features, int* n_descriptors)
{
    if (!c) return -1;
    if (n_features) *n_features = c->ext_total;
    if (n_descriptors) *n_descriptors = c->ori_total;
    return 0;
}
int oracle_ext_count(const oracle_ctx* c, int octave)
{
    if (!c || octave < 0 || octave >= MAXO) return -1;
    return c->ext_ct[octave];
}

/* prep_features, sift_pyramid.cu:249-279: note int up_fac (truncation of the float) */
int oracle_fetch(const oracle_ctx* c, popsift_hip_feature* feats, float* desc)
{
    if (!c) return -1;
    const int up_fac = (int)c->p.upscale_factor;
    for (int i = 0; i < c->ext_total; i++) {
        const ext_t*         e = &c->ext[i];
        popsift_hip_feature* f = &feats[i];
        const float          sc = powf(2.0f, (float)(e->octave - up_fac));
        f->debug_octave = e->octave;
        f->xpos = e->xpos * sc;
        f->ypos = e->ypos * sc;
        f->sigma = e->sigma * sc;
        f->num_ori = e->num_ori;
        int k;
        for (k = 0; k < e->num_ori; k++) {
            f->desc_idx[k] = e->idx_ori + k;
            f->orientation[k] = e->orientation[k];
        }
        for (; k < 4; k++) {
            f->desc_idx[k] = -1;
            f->orientation[k] = 0;
        }
    }
    if (desc && c->ori_total) memcpy(desc, c->desc, sizeof(float) * 128 * (size_t)c->ori_total);
    return 0;
}

int oracle_fetch_raw_desc(const oracle_ctx* c, float* desc)
{
    if (!c) return -1;
    if (desc && c->ori_total) memcpy(desc, c->desc_raw, sizeof(float) * 128 * (size_t)c->ori_total);
    return 0;
}

int oracle_fetch_extrema(const oracle_ctx* c, popsift_hip_extremum* out)
{
    if (!c) return -1;
    for (int i = 0; i < c->ext_total; i++) {
        out[i].xpos = c->ext[i].xpos;
        out[i].ypos = c->ext[i].ypos;
        out[i].lpos = c->ext[i].lpos;
        out[i].sigma = c->ext[i].sigma;
        out[i].octave = c->ext[i].octave;
        out[i].cell = c->ext[i].cell;
    }
    return 0;
}
