/* devfeatures.h -- device-resident result set (internal; the C ABI sees an opaque pointer). */
#pragma once
#include "sift_types.h"

/* popsift::Feature as laid out on an LP64 host (features.h:22-34): 72 bytes, the four
 * Descriptor* point into `desc` of the same set */
struct DevFeature {
    int    debug_octave;
    float  xpos, ypos, sigma;
    int    num_ori;
    float  orientation[POPSIFT_HIP_ORI_MAX];
    int    pad;
    float* desc[POPSIFT_HIP_ORI_MAX];
};
static_assert(sizeof(DevFeature) == 72, "popsift::Feature layout");

struct popsift_hip_devfeatures {
    int         device = 0;
    int         n_feat = 0, n_desc = 0;
    DevFeature* d_feat = nullptr; /* n_feat      */
    float*      d_desc = nullptr; /* n_desc * 128 */
    int*        d_rev = nullptr;  /* n_desc: descriptor -> feature (feat_to_ext_map) */
    /* scratch of popsift_hip_match_sets with this set on the left, kept between calls */
    void*  m_stream = nullptr;  /* hipStream_t */
    void*  m_partial = nullptr;
    size_t m_partial_cap = 0;
    void*  m_out = nullptr;     /* popsift_hip_match[n_desc] */
    void*  m_host = nullptr;    /* pinned staging of the same size */
    int*   m_redo = nullptr;    /* [0] = count, [1 ..] = rows the screening pass left to the exact kernel */
    float* d_norm = nullptr;    /* |x|^2 per descriptor, computed on first use by a match */
    float* m_rnorm = nullptr;   /* norms of the right set of the current call */
    size_t m_rnorm_cap = 0;
};
