/*
 * popsift_hip.h -- thin C ABI between the C++ host layer (libpopsift) and the
 * HIP/CDNA4 SIFT extraction kernels (libpopsift_hip.so).
 *
 * The reference (10183308/popsift) has no FFI: its boundary is the C++ class API
 * (popsift.h:40-167).  BASELINE.json's north_star asks for "the C++ host ...
 * calling HIP through a thin C-ABI"; SURVEY.md section 8(b) lists the entry points.
 * Each function below cites the reference member function(s) it replaces.
 *
 * Conventions: every function returns 0 on success and a negative
 * popsift_hip_status on failure; no C++ types, no exceptions, no torch types.
 * A context is owned by one host thread at a time; distinct contexts (also on
 * the same device) are fully independent (the reference's process-global
 * counters, sift_pyramid.cu:38-49, do not exist here).
 */
#ifndef POPSIFT_HIP_H
#define POPSIFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POPSIFT_HIP_MAX_OCTAVES 20 /* sift_conf.h:13 MAX_OCTAVES */
#define POPSIFT_HIP_MAX_LEVELS 12  /* sift_constants.h:35 GAUSS_LEVELS */
#define POPSIFT_HIP_MAX_BATCH 16   /* images of one size a context extracts per submit (popsift_hip_submit_batch) */
#define POPSIFT_HIP_GAUSS_ALIGN 32 /* sift_constants.h:34 GAUSS_ALIGN */
#define POPSIFT_HIP_ORI_MAX 4      /* sift_constants.h:51 ORIENTATION_MAX_COUNT */

typedef enum popsift_hip_status {
    POPSIFT_HIP_OK = 0,
    POPSIFT_HIP_ERR_INVALID = -1,     /* bad argument / unsupported mode        */
    POPSIFT_HIP_ERR_DEVICE = -2,      /* a HIP runtime call failed              */
    POPSIFT_HIP_ERR_NO_DEVICE = -3,   /* no usable GPU                          */
    POPSIFT_HIP_ERR_OOM = -4,         /* host or device allocation failed       */
    POPSIFT_HIP_ERR_STATE = -5,       /* call sequence error (e.g. fetch first) */
    POPSIFT_HIP_ERR_TOO_SMALL = -6    /* caller buffer too small                */
} popsift_hip_status;

/* enum values follow the declaration order in sift_conf.h:32-72 */
enum { POPSIFT_HIP_SIFT_POPSIFT = 0, POPSIFT_HIP_SIFT_OPENCV = 1, POPSIFT_HIP_SIFT_VLFEAT = 2 };
enum { POPSIFT_HIP_GAUSS_VLFEAT_COMPUTE = 0, POPSIFT_HIP_GAUSS_VLFEAT_RELATIVE = 1,
       POPSIFT_HIP_GAUSS_VLFEAT_RELATIVE_ALL = 2, POPSIFT_HIP_GAUSS_OPENCV_COMPUTE = 3,
       POPSIFT_HIP_GAUSS_FIXED9 = 4, POPSIFT_HIP_GAUSS_FIXED15 = 5 };
enum { POPSIFT_HIP_DESC_LOOP = 0, POPSIFT_HIP_DESC_ILOOP = 1, POPSIFT_HIP_DESC_GRID = 2,
       POPSIFT_HIP_DESC_IGRID = 3, POPSIFT_HIP_DESC_NOTILE = 4 };
enum { POPSIFT_HIP_NORM_ROOTSIFT = 0, POPSIFT_HIP_NORM_CLASSIC = 1 };
/* Config::GridFilterMode (sift_conf.h:68-72): which extrema of an over-full cell survive */
enum { POPSIFT_HIP_FILTER_RANDOM = 0, POPSIFT_HIP_FILTER_LARGEST_FIRST = 1, POPSIFT_HIP_FILTER_SMALLEST_FIRST = 2 };

/* Flattened popsift::Config (sift_conf.h:28-310, defaults sift_conf.cu:17-39). */
typedef struct popsift_hip_params {
    int32_t octaves;             /* -1 = auto (popsift.cpp:107-111)            */
    int32_t levels;              /* DoG search levels, clamped to >= 2         */
    float   sigma;               /* 1.6                                        */
    float   edge_limit;          /* 10                                         */
    float   threshold;           /* Config::_threshold, 0.04                   */
    float   upscale_factor;      /* +1 => input stretched by 2                 */
    int32_t sift_mode;           /* POPSIFT_HIP_SIFT_*                         */
    int32_t gauss_mode;          /* POPSIFT_HIP_GAUSS_*                        */
    int32_t desc_mode;           /* POPSIFT_HIP_DESC_*                         */
    int32_t norm_mode;           /* POPSIFT_HIP_NORM_*                         */
    int32_t norm_multi;          /* descriptor *= 2^norm_multi                 */
    int32_t max_extrema;         /* per octave, 100000                         */
    int32_t assume_initial_blur; /* 1                                          */
    float   initial_blur;        /* 0.5                                        */
    int32_t filter_grid_size;    /* 2: the grid filter works on size x size cells */
    int32_t filter_max_extrema;  /* -1 = grid filter off (s_orientation.cu:362)  */
    int32_t filter_sorting;      /* POPSIFT_HIP_FILTER_*                         */
    int32_t store_dog;           /* 0 (default): DoG planes are formed on the fly by their consumers (bit-identical);
                                  * 1: stored as the reference does (s_pyramid_build.cu:74-92), for stage tests     */
    int32_t reserved[2];
} popsift_hip_params;

/* POD mirror of popsift::Feature (features.h:22-34): the four Descriptor*
 * become indices into the descriptor array (-1 = unused slot). */
typedef struct popsift_hip_feature {
    int32_t debug_octave;
    float   xpos;
    float   ypos;
    float   sigma;
    int32_t num_ori;
    float   orientation[POPSIFT_HIP_ORI_MAX];
    int32_t desc_idx[POPSIFT_HIP_ORI_MAX];
} popsift_hip_feature;

/* Pre-orientation extremum (sift_extremum.h:24-33 InitialExtremum), for stage tests. */
typedef struct popsift_hip_extremum {
    float   xpos;
    float   ypos;
    int32_t lpos;
    float   sigma;
    int32_t octave;
    int32_t cell;
} popsift_hip_extremum;

/* Per-image timing / counter report (device side, HIP events). */
typedef struct popsift_hip_report {
    int32_t num_octaves;
    int32_t base_w, base_h;
    int32_t ext_ct[POPSIFT_HIP_MAX_OCTAVES];
    int32_t ori_ct[POPSIFT_HIP_MAX_OCTAVES];
    int32_t ext_total, ori_total;
    float   ms_device;      /* first kernel -> last kernel of the image          */
    float   ms_blur;        /* sum of blur-level kernel durations (profile mode) */
    int32_t blur_launches;  /* number of blur-level launches (profile mode)      */
    double  blur_alg_bytes; /* algorithmic bytes of those launches               */
    double  pyramid_pixels; /* sum over octaves of w*h                           */
    /* the same three for the 64-row-tile instantiation only (k_blur_tile<HALO,0,64>:
     * the level launches of the large octaves, the dominant kernel of the pipeline) */
    double  big_alg_bytes;
    float   ms_big;
    int32_t big_launches;
    /* profile mode 2: device time of the stages of one image, HIP events on the context's stream between the
     * launches (POPSIFT_HIP_STAGE_*); 0 when not collected */
    float   ms_stage[8];
} popsift_hip_report;
enum { POPSIFT_HIP_STAGE_PYRAMID = 0,     /* every k_blur_tile launch                              */
       POPSIFT_HIP_STAGE_DETECT = 1,      /* k_detect (both passes)                                */
       POPSIFT_HIP_STAGE_REFINE = 2,      /* k_refine (+ the grid filter when enabled)             */
       POPSIFT_HIP_STAGE_ORIENTATION = 3, /* k_orientation                                         */
       POPSIFT_HIP_STAGE_SCAN = 4,        /* k_scan_local + k_scan_apply                           */
       POPSIFT_HIP_STAGE_DESCRIPTOR = 5,  /* the descriptor kernel                                 */
       POPSIFT_HIP_STAGE_COUNT = 6 };

typedef struct popsift_hip_ctx popsift_hip_ctx;

/* Fills p with the defaults of popsift::Config::Config() (sift_conf.cu:17-39). */
void popsift_hip_default_params(popsift_hip_params* p);

/* Library / build identification string (static storage). */
const char* popsift_hip_version(void);
const char* popsift_hip_strerror(int status);
/* Message of the last failure on this context (static per-context storage). */
const char* popsift_hip_last_error(const popsift_hip_ctx* ctx);

/* Replaces common/device_prop.cu:23-47 (device enumeration). */
int popsift_hip_device_count(int* count);

/* What device_prop_t::print shows (common/device_prop.cu:39-70), as a POD. */
typedef struct popsift_hip_device_info {
    char     name[256];
    int32_t  arch_major, arch_minor; /* "compute capability" slot: gfx major / minor */
    uint64_t total_mem;              /* bytes of HBM */
    uint64_t lds_per_block;          /* "per-block shared mem" */
    int32_t  wave_size;              /* 64 on CDNA */
    int32_t  max_threads_per_block;
    int32_t  max_threads_per_cu;
    int32_t  max_block[3];
    int32_t  max_grid[3];
    int32_t  cu_count;               /* "number of SM(x)s" */
    int32_t  concurrent_kernels;
    int32_t  can_map_host;
    int32_t  unified_addressing;
} popsift_hip_device_info;
int popsift_hip_get_device_info(int device, popsift_hip_device_info* out);
/* NUMA node the GPU hangs off (from its PCI address, /sys/bus/pci/devices/<id>/numa_node), -1 when the host does not
 * say.  The C++ layer binds each worker thread to that node's CPUs, so that the staging copy of the image and the
 * pinned result blocks the worker allocates are node-local; no counterpart in the single-GPU
 * reference. */
int popsift_hip_device_numa_node(int device, int* node);

/* Replaces PopSift::configure (popsift.cpp:63-87: init_filter + init_constants)
 * and Pyramid::Pyramid (sift_pyramid.cu:108-165); buffers are sized lazily on
 * the first image and only grow (contrast sift_octave.cu:55-89). */
int popsift_hip_ctx_create(int device, const popsift_hip_params* p, popsift_hip_ctx** out);
/* Replaces PopSift::uninit / Pyramid::~Pyramid (popsift.cpp:122-137). */
int popsift_hip_ctx_destroy(popsift_hip_ctx* ctx);

/* Gauss tables as uploaded to the device (gauss_filter.cu:127-257), for KATs:
 * filter[(levels+3) * 32], span[levels+3], sigma[levels+3]. */
int popsift_hip_get_gauss_table(const popsift_hip_ctx* ctx, float* filter, int* span, float* sigma,
                                int* n_levels);

/* Replaces Image::load + Pyramid::step1 + step2 (s_image.cu:71-79,
 * sift_pyramid.cu:226-239): upload one host image and enqueue the whole
 * extraction asynchronously on the context's stream.  pitch in elements.
 * u8 values are 0..255, f32 values are [0,1) (popsift.h:108-116). */
int popsift_hip_submit_u8(popsift_hip_ctx* ctx, const uint8_t* img, int w, int h, int pitch);
int popsift_hip_submit_f32(popsift_hip_ctx* ctx, const float* img, int w, int h, int pitch);
/* Same for an image in page-locked host memory (popsift_hip_host_alloc): it is uploaded straight from there, without
 * the staging copy Image::load makes (s_image.cu:71-79), so the caller keeps it valid and unchanged until
 * popsift_hip_wait has returned for this image.  PopSift::enqueue's one copy of the caller's image lands in such a
 * block (popsift.cpp:245-247 makes that copy too, and the upload thread a second one). */
int popsift_hip_submit_pinned_u8(popsift_hip_ctx* ctx, const uint8_t* img, int w, int h, int pitch);
int popsift_hip_submit_pinned_f32(popsift_hip_ctx* ctx, const float* img, int w, int h, int pitch);
/* Same, image already resident in this device's memory (bench "inputs in HBM"). */
int popsift_hip_submit_dev_u8(popsift_hip_ctx* ctx, const void* d_img, int w, int h, int pitch);
int popsift_hip_submit_dev_f32(popsift_hip_ctx* ctx, const void* d_img, int w, int h, int pitch);

/*
 * Several images of ONE size per submit (round 3).  The reference's demo enqueues all its images and then collects all
 * the results (src/application/main.cpp:304-326) while its pipeline still extracts them one by one
 * (popsift.cpp:139-213); here up to POPSIFT_HIP_MAX_BATCH images go through the per-image launch sequence TOGETHER --
 * every kernel is launched once for the whole batch, image index in blockIdx.y -- so the latency-bound launches of the
 * small octaves, refinement and the scans are paid once per batch.  Every image has its own planes, lists and result
 * slabs: its results are bit-identical to a submit of its own, whatever it is batched with
 * (tests/test_gpu_batch.py).  kind says where the images lie and what they hold; pointers, w, h, pitch as for the
 * single-image calls above, which are batches of one.  The slots of a context are allocated on first use and kept.
 */
enum { POPSIFT_HIP_IMG_HOST_U8 = 0, POPSIFT_HIP_IMG_HOST_F32 = 1, POPSIFT_HIP_IMG_DEV_U8 = 2, POPSIFT_HIP_IMG_DEV_F32 = 3,
       POPSIFT_HIP_IMG_PINNED_U8 = 4, POPSIFT_HIP_IMG_PINNED_F32 = 5 };
int popsift_hip_submit_batch(popsift_hip_ctx* ctx, const void* const* imgs, int n, int kind, int w, int h, int pitch);
/* blocks until the batch is finished; n_features / n_descriptors: arrays of at least *n_images (= the n submitted) */
int popsift_hip_wait_batch(popsift_hip_ctx* ctx, int* n_images, int* n_features, int* n_descriptors);
/* results of image k of the finished batch (popsift_hip_fetch / popsift_hip_results_dev are k = 0) */
int popsift_hip_fetch_item(popsift_hip_ctx* ctx, int k, popsift_hip_feature* feats, size_t feats_cap, float* desc,
                           size_t desc_cap);
int popsift_hip_results_dev_item(popsift_hip_ctx* ctx, int k, const void** d_feats, const void** d_desc);
/* popsift_hip_fetch_begin (below) for image k of the finished batch: call it for every image whose results are wanted,
 * submit the next batch, then ONE popsift_hip_fetch_end waits for all the downloads */
int popsift_hip_fetch_begin_item(popsift_hip_ctx* ctx, int k, popsift_hip_feature* feats, size_t feats_cap, float* desc,
                                 size_t desc_cap);

/* Replaces the counter read-back of Pyramid::get_descriptors
 * (sift_pyramid.cu:281-294): blocks until the submitted image is finished and
 * returns the feature / descriptor counts. */
int popsift_hip_wait(popsift_hip_ctx* ctx, int* n_features, int* n_descriptors);
/* Replaces prep_features + the two D2H copies (sift_pyramid.cu:249-321).
 * feats: n_features entries; desc: n_descriptors * 128 floats. */
int popsift_hip_fetch(popsift_hip_ctx* ctx, popsift_hip_feature* feats, size_t feats_cap,
                      float* desc, size_t desc_cap);
/* The same download split in two, so that one context overlaps it with the kernels of its next image
 * (the reference serialises them: Pyramid::get_descriptors blocks on its two copies, sift_pyramid.cu:296-321,
 * before PopSift::extractDownloadLoop takes the next job, popsift.cpp:187-213).
 * fetch_begin: requires a finished image (it waits like popsift_hip_wait), starts the two copies on the context's
 * copy stream and returns at once; the context switches to its second result slab, so the next submit / wait may
 * follow immediately.  feats / desc should be pinned (popsift_hip_host_alloc) -- a pageable target makes the copy
 * synchronous -- and must stay untouched until fetch_end returns.
 * fetch_end: blocks until that download has landed.  ERR_STATE without a pending download.
 * After fetch_begin the image's results are no longer in the context: fetch, fetch_begin, results_dev and
 * clone_results return ERR_STATE until another image has been submitted.  A second fetch_begin (for the next image)
 * first waits for the pending download; so does ctx_destroy. */
int popsift_hip_fetch_begin(popsift_hip_ctx* ctx, popsift_hip_feature* feats, size_t feats_cap,
                            float* desc, size_t desc_cap);
int popsift_hip_fetch_end(popsift_hip_ctx* ctx);
/* Device-resident results (FeaturesDev analogue, features.h:98-118): pointers
 * stay valid until the next submit on this context. */
int popsift_hip_results_dev(popsift_hip_ctx* ctx, const void** d_feats, const void** d_desc);

/* Pinned (page-locked) host memory for result buffers: a D2H copy into it runs at PCIe speed
 * without staging.  Replaces FeaturesHost::pin/unpin (cudaHostRegister per image,
 * features.cu:84-109).  Returns NULL on failure / when no GPU runtime is usable. */
void* popsift_hip_host_alloc(size_t bytes);
void  popsift_hip_host_free(void* p);

/* ---- MatchingMode ------------------------------------------------------------
 * Device-resident copy of the last image's results: replaces Pyramid::clone_device_descriptors
 * (sift_pyramid.cu:323-361) and FeaturesDev (features.h:98-118).  The set lives on the context's
 * GPU and is independent of the context afterwards. */
typedef struct popsift_hip_devfeatures popsift_hip_devfeatures;
int popsift_hip_clone_results(popsift_hip_ctx* ctx, popsift_hip_devfeatures** out);
int popsift_hip_devfeatures_free(popsift_hip_devfeatures* f);
int popsift_hip_devfeatures_info(const popsift_hip_devfeatures* f, int* device, int* n_features, int* n_descriptors);
/* Device pointers: features in the 72-byte layout of popsift::Feature (features.h:22-34) whose
 * desc[] point into the descriptor array, descriptors (128 floats each), descriptor -> feature map. */
int popsift_hip_devfeatures_ptrs(const popsift_hip_devfeatures* f, void** d_features, void** d_descriptors,
                                 void** d_reverse_map);
/* An empty (zero-filled) set of the given sizes: FeaturesDev::reset (features.cu:148-160). */
int popsift_hip_devfeatures_alloc(int device, int n_features, int n_descriptors, popsift_hip_devfeatures** out);
/* A set from caller-supplied HOST descriptors (n * 128 floats), for matching without extraction. */
int popsift_hip_devfeatures_from_host(int device, const float* desc, int n_descriptors, popsift_hip_devfeatures** out);
/* Host copies (tests, printing): desc n_descriptors*128 floats, rev n_descriptors ints; either may be NULL. */
int popsift_hip_devfeatures_download(const popsift_hip_devfeatures* f, float* desc, int32_t* rev);

/* One row of the reference's match_matrix (int3, features.cu:178-220) plus the two squared distances. */
typedef struct popsift_hip_match {
    int32_t best;        /* index of the nearest right descriptor        */
    int32_t second;      /* index of the second nearest                  */
    int32_t accept;      /* dist_best / dist_second < 0.8                */
    float   dist_best;   /* squared L2 distances                         */
    float   dist_second;
} popsift_hip_match;
/* Replaces FeaturesDev::match / compute_distance (features.cu:157-300): brute-force 2-NN of every
 * descriptor of `l` among the descriptors of `r`; out has l's descriptor count entries (host memory).
 * Sets on different GPUs are allowed (the right set is copied to the left set's GPU). */
int popsift_hip_match_sets(const popsift_hip_devfeatures* l, const popsift_hip_devfeatures* r, popsift_hip_match* out);
/* Which kernels popsift_hip_match_sets uses, process-wide (tests compare the paths; results are identical):
 * AUTO (default): matrix-core screening + exact re-rank from 4e6 pairs on, the exact brute-force kernel below;
 * EXACT: the exact kernel only; SCREEN: screening for every size. */
enum { POPSIFT_HIP_MATCH_AUTO = 0, POPSIFT_HIP_MATCH_EXACT = 1, POPSIFT_HIP_MATCH_SCREEN = 2 };
int popsift_hip_match_set_path(int path);

int popsift_hip_get_report(const popsift_hip_ctx* ctx, popsift_hip_report* rep);
/* profile != 0: bracket every blur-level launch with HIP events (serialises the
 * octave streams; used by bench.py for the roofline object only). */
int popsift_hip_set_profile(popsift_hip_ctx* ctx, int profile); /* 0 off, 1 blur launches, 2 stages */

/* Debug / parity hooks (replace Octave::download_and_save_array,
 * sift_octave.cu:110-187).  kind: 0 = Gaussian plane, 1 = DoG plane. */
int popsift_hip_octave_dims(const popsift_hip_ctx* ctx, int octave, int* w, int* h);
int popsift_hip_download_plane(popsift_hip_ctx* ctx, int octave, int kind, int level, float* out);
/* Overwrite a plane (stage isolation in tests), then re-run later stages.  kind = 1 needs params.store_dog = 1
 * (POPSIFT_HIP_ERR_STATE otherwise: the consumers form DoG values from the Gaussian planes). */
int popsift_hip_upload_plane(popsift_hip_ctx* ctx, int octave, int kind, int level, const float* in);
/* Initial extrema of the last image, all octaves, in device compaction order. */
int popsift_hip_download_extrema(popsift_hip_ctx* ctx, popsift_hip_extremum* out, size_t cap, int* n);
/* Re-run extrema + orientation + descriptors on the planes currently in memory. */
int popsift_hip_rerun_keypoint_stages(popsift_hip_ctx* ctx);
/* Test switches of one context (no environment variables are read by this library).  Set them before the first
 * submit: DET_QCAP = candidate-queue entries the fast detection pass may use (small values force strips into the
 * slow pass); CAND_CAP / OHIST_CAP = initial capacity of the candidate buffer / of the orientation-histogram buffer
 * (small values exercise the grow-and-rerun path of popsift_hip_wait); FAIL_ALLOC = n: the n-th device allocation
 * of this context from now on fails with POPSIFT_HIP_ERR_OOM (0 = off); DESC_ROWS = patch rows the loop descriptor
 * walks per pass (4 .. 128, default 128: small values make ordinary patches take the several passes that otherwise only
 * patches of more than 128 rows take -- sigma0 near 2 at the coarsest level; results do not depend on it); PYR_ORDER = 0:
 * level 1 of octave 1 after ALL levels of octave 0 (the default), 1: right behind the level that writes its source
 * plane (results do not depend on it; tools/pyr_order.sh times both); KP_WAVES = waves per image in the launches of
 * the orientation and descriptor kernels (a multiple of 32; default 8 per wave slot of the device; results do not
 * depend on it; tools/kp_waves_sweep.sh: 8192 .. 131072 within 1.5 %); BLUR_PATH = which kernels build the pyramid's
 * plane-to-plane levels: 0 by plane size (default), 1 the one-tile-per-workgroup kernels only, 2 the strip-march kernels
 * wherever they apply, also on small planes (results do not depend on it: every path is bit-identical; the tests run
 * their small images through 2); BLUR_SEG = rows per segment of the march kernels (a multiple of 32; 0 = chosen from the
 * plane and the batch); PYR_TAIL = 0: the smallest octaves -- from the first whose plane fits one workgroup's LDS -- are
 * built by one launch (default), 1: by level launches like the others (results do not depend on it). */
enum { POPSIFT_HIP_DEBUG_DET_QCAP = 1, POPSIFT_HIP_DEBUG_CAND_CAP = 2, POPSIFT_HIP_DEBUG_OHIST_CAP = 3,
       POPSIFT_HIP_DEBUG_FAIL_ALLOC = 4, POPSIFT_HIP_DEBUG_DESC_ROWS = 5, POPSIFT_HIP_DEBUG_PYR_ORDER = 6,
       POPSIFT_HIP_DEBUG_KP_WAVES = 7, POPSIFT_HIP_DEBUG_BLUR_PATH = 8, POPSIFT_HIP_DEBUG_BLUR_SEG = 9,
       POPSIFT_HIP_DEBUG_PYR_TAIL = 10 };
int popsift_hip_debug_set(popsift_hip_ctx* ctx, int what, int value);

#ifdef __cplusplus
}
#endif
#endif /* POPSIFT_HIP_H */
