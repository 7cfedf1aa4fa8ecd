/*
 * popsift_oracle.h -- CPU restatement of the PopSift extraction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under popsift_amd/ or include/ may call,
 * link or import this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do (as the checker / reported baseline, never as the product).
 *
 * PARITY UNPINNED: the reference (CUDA + Boost) cannot be built in this image
 * and ships no golden vectors in-tree (its regression tarball is a download,
 * testScripts/downloadOxfordDataset.sh.in:4-9).  This file follows the
 * reference source line by line (citations per function in popsift_oracle.c)
 * and is pinned only by the known-answer tests of SURVEY.md section 8(c).
 */
#ifndef POPSIFT_ORACLE_H
#define POPSIFT_ORACLE_H

#include "../include/popsift_hip.h" /* shared POD structs only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

oracle_ctx* oracle_create(const popsift_hip_params* p);
void        oracle_destroy(oracle_ctx* c);
/* threads <= 1: scalar single-thread; > 1: OpenMP over rows / keypoints */
void        oracle_set_threads(oracle_ctx* c, int threads);

/* gauss_filter.cu:127-372 (inc table only; dd[0] == inc[0]) */
int oracle_get_gauss_table(const oracle_ctx* c, float* filter, int* span, float* sigma, int* n_levels);
/* popsift.cpp:89-120 */
int oracle_plan(const oracle_ctx* c, int w, int h, int* n_octaves, int* base_w, int* base_h);

/* full pipeline: pyramid, DoG, extrema, orientation, descriptors, normalise */
int oracle_run_u8(oracle_ctx* c, const uint8_t* img, int w, int h, int pitch);
int oracle_run_f32(oracle_ctx* c, const float* img, int w, int h, int pitch);
/* stage control */
int oracle_build_pyramid_u8(oracle_ctx* c, const uint8_t* img, int w, int h, int pitch);
int oracle_build_pyramid_f32(oracle_ctx* c, const float* img, int w, int h, int pitch);
int oracle_run_keypoint_stages(oracle_ctx* c); /* on the planes currently held */

int          oracle_num_octaves(const oracle_ctx* c);
int          oracle_octave_dims(const oracle_ctx* c, int octave, int* w, int* h);
/* kind 0 = Gaussian plane (levels+3 of them), 1 = DoG plane (levels+2) */
const float* oracle_plane(const oracle_ctx* c, int octave, int kind, int level);
float*       oracle_plane_mut(oracle_ctx* c, int octave, int kind, int level);

int oracle_counts(const oracle_ctx* c, int* n_features, int* n_descriptors);
int oracle_ext_count(const oracle_ctx* c, int octave);
int oracle_fetch(const oracle_ctx* c, popsift_hip_feature* feats, float* desc);
int oracle_fetch_extrema(const oracle_ctx* c, popsift_hip_extremum* out);
/* un-normalised 128-bin histograms, same order as the descriptors */
int oracle_fetch_raw_desc(const oracle_ctx* c, float* desc);

/* test hook: recompute all descriptors in the frames of the orientations `ori` (4 per extremum, order of oracle_fetch;
 * NULL = the oracle's own) and scales `sigma` (1 per extremum, octave units; NULL = own), moved by so many units in the
 * last place */
int oracle_redo_descriptors(oracle_ctx* c, const float* ori, int ori_ulps, const float* sigma, int sigma_ulps);

/* isolated helpers for unit tests */
void  oracle_warp32_sort64(const float* yval64, int* out64);  /* common/warp_bitonic_sort.h:35-78, all 64 indices */
int   oracle_solve3(float A[9], float b[3]);                 /* s_solve.h:24-85 */
void  oracle_normalize(float* d128, int norm_mode, int norm_multi); /* s_desc_norm_*.h */
/* brute-force 2-NN of every left descriptor among the right ones (features.cu:157-221) */
void  oracle_match(const float* l, int l_len, const float* r, int r_len, popsift_hip_match* out, int threads);
/* grid filter on bare keys (s_filtergrid.cu:109-322): keep[i] = 1 for survivors; returns the per-cell limit */
int   oracle_filter_grid_keys(const int* cell, const float* scale, int n_ext, int grid_size, int filter_max,
                              int mode, unsigned char* keep);

#ifdef __cplusplus
}
#endif
#endif
