/*
 * Sanitizer driver for the CPU oracle (test infrastructure): the whole pipeline -- every SiftMode, both Gauss modes,
 * every descriptor mode, both normalisations, the grid filter, float input, odd sizes, the matcher -- on synthetic
 * images, compiled together with popsift_oracle.c under -fsanitize=address,undefined (make -C oracle san).
 * Exits non-zero on a sanitizer report (-fno-sanitize-recover) or an inconsistent result.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../popsift_oracle.h"

static uint32_t lcg(uint32_t* s)
{
    *s = *s * 1664525u + 1013904223u;
    return *s >> 8;
}

static void make_image(uint8_t* img, int w, int h, uint32_t seed)
{
    uint32_t s = seed;
    for (int i = 0; i < w * h; i++) img[i] = (uint8_t)(96 + lcg(&s) % 64);
    for (int k = 0; k < (w * h) / 400 + 1; k++) { /* blobs */
        const int   cx = (int)(lcg(&s) % (uint32_t)w), cy = (int)(lcg(&s) % (uint32_t)h);
        const float sd = 1.5f + (float)(lcg(&s) % 80) / 10.0f, amp = (lcg(&s) & 1) ? 70.0f : -70.0f;
        const int   r = (int)(3.0f * sd);
        for (int y = cy - r; y <= cy + r; y++)
            for (int x = cx - r; x <= cx + r; x++) {
                if (x < 0 || y < 0 || x >= w || y >= h) continue;
                const float d2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                float       v = (float)img[y * w + x] + amp * expf(-0.5f * d2 / (sd * sd));
                img[y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
}

static int run_case(const char* name, popsift_hip_params p, int w, int h, int as_float, int threads)
{
    uint8_t* img = (uint8_t*)malloc((size_t)w * h);
    float*   fimg = (float*)malloc(sizeof(float) * (size_t)w * h);
    make_image(img, w, h, 12345u + (uint32_t)(w * 31 + h));
    for (int i = 0; i < w * h; i++) fimg[i] = (float)img[i] / 256.0f;
    oracle_ctx* c = oracle_create(&p);
    if (!c) return 1;
    oracle_set_threads(c, threads);
    const int rc = as_float ? oracle_run_f32(c, fimg, w, h, w) : oracle_run_u8(c, img, w, h, w);
    int       nf = 0, nd = 0, bad = rc != 0;
    if (!bad) {
        oracle_counts(c, &nf, &nd);
        popsift_hip_feature* f = (popsift_hip_feature*)calloc((size_t)nf + 1, sizeof(*f));
        float*               d = (float*)calloc((size_t)nd * 128 + 1, sizeof(float));
        oracle_fetch(c, f, d);
        int sum = 0;
        for (int i = 0; i < nf; i++) {
            sum += f[i].num_ori;
            if (f[i].num_ori < 1 || f[i].num_ori > 4) bad = 1;
        }
        if (sum != nd) bad = 1;
        for (int i = 0; i < nd * 128; i++)
            if (!(d[i] >= 0.0f)) bad = 1; /* also catches NaN */
        if (nd >= 2) { /* the matcher on the descriptors against themselves */
            popsift_hip_match* m = (popsift_hip_match*)calloc((size_t)nd, sizeof(*m));
            oracle_match(d, nd, d, nd, m, threads);
            for (int i = 0; i < nd; i++)
                if (m[i].best < 0 || m[i].best >= nd) bad = 1;
            free(m);
        }
        free(f);
        free(d);
    }
    printf("%-28s %4dx%-4d features %5d descriptors %5d %s\n", name, w, h, nf, nd, bad ? "FAILED" : "ok");
    oracle_destroy(c);
    free(img);
    free(fimg);
    return bad;
}

int main(void)
{
    popsift_hip_params d;
    memset(&d, 0, sizeof(d));
    d.octaves = -1;
    d.levels = 3;
    d.sigma = 1.6f;
    d.edge_limit = 10.0f;
    d.threshold = 0.04f;
    d.upscale_factor = 1.0f;
    d.max_extrema = 100000;
    d.assume_initial_blur = 1;
    d.initial_blur = 0.5f;
    d.filter_grid_size = 2;
    d.filter_max_extrema = -1;
    int bad = 0;
    popsift_hip_params p = d;
    bad += run_case("default", p, 160, 120, 0, 1);
    bad += run_case("default, 4 threads", p, 161, 97, 0, 4);
    p = d; p.sift_mode = POPSIFT_HIP_SIFT_OPENCV; p.gauss_mode = POPSIFT_HIP_GAUSS_OPENCV_COMPUTE; p.norm_mode = POPSIFT_HIP_NORM_CLASSIC;
    bad += run_case("opencv, classic norm", p, 133, 101, 0, 1);
    p = d; p.sift_mode = POPSIFT_HIP_SIFT_VLFEAT; p.octaves = 3; p.levels = 5;
    bad += run_case("vlfeat, 5 levels", p, 120, 90, 1, 2);
    for (int dm = POPSIFT_HIP_DESC_ILOOP; dm <= POPSIFT_HIP_DESC_NOTILE; dm++) {
        p = d; p.desc_mode = dm; p.upscale_factor = 0.0f;
        char name[64];
        snprintf(name, sizeof(name), "desc mode %d, no upscale", dm);
        bad += run_case(name, p, 96, 80, 0, 2);
    }
    p = d; p.upscale_factor = -1.0f; p.norm_multi = 9;
    bad += run_case("downscale, norm_multi 9", p, 200, 150, 0, 1);
    p = d; p.filter_max_extrema = 20; p.filter_grid_size = 3; p.filter_sorting = POPSIFT_HIP_FILTER_LARGEST_FIRST;
    bad += run_case("grid filter", p, 160, 120, 0, 2);
    p = d; p.max_extrema = 50;
    bad += run_case("max_extrema 50", p, 160, 120, 0, 1);
    bad += run_case("tiny", d, 9, 9, 0, 1);
    bad += run_case("one row too small", d, 17, 3, 0, 1);
    printf("%s\n", bad ? "oracle_san: FAILED" : "oracle_san: ok");
    return bad ? 1 : 0;
}
