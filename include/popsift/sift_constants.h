/*
 * popsift/sift_constants.h -- public compile-time constants of the drop-in API.
 * Replaces the public part of the reference's sift_constants.h:34-52 (the
 * __constant__ ConstInfo table of that file lives inside libpopsift_hip here).
 */
#pragma once

#define GAUSS_ALIGN 32
#define GAUSS_LEVELS 12

#define ORI_NBINS 36
#define ORI_WINFACTOR 1.5F

#define DESC_BINS 8
#define DESC_MAGNIFY 3.0f

/* VLFeat convention: at most 4 orientations per extremum (Lowe: 3) */
#define ORIENTATION_MAX_COUNT 4
