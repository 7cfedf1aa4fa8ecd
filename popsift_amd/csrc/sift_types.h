/*
 * sift_types.h -- device/host shared plain structs of libpopsift_hip.
 * Data layout in HBM (see DESIGN.md "Data layout"):
 *   - every Gaussian / DoG plane is a row-major float plane, row pitch padded
 *     to 64 floats (256 B), all planes of a context carved from one arena;
 *   - extrema / features / descriptors are flat arrays sized by max_extrema.
 */
#pragma once
#include <stdint.h>

#include <hip/hip_vector_types.h> /* int2, float2 */

#include "../../include/popsift_hip.h"

#define PS_MAX_OCT POPSIFT_HIP_MAX_OCTAVES
#define PS_MAX_PLANES (POPSIFT_HIP_MAX_LEVELS + 3)
#define PS_GA POPSIFT_HIP_GAUSS_ALIGN
#define PS_ORI_NBINS 36

/* One octave: planes are at base + level * plane_stride (floats). */
struct OctDesc {
    float*  data;         /* L Gaussian planes   */
    float*  dog;          /* L-1 DoG planes      */
    int64_t data_off;     /* the same, as float offsets from the arena base: kernels add them to the */
    int64_t dog_off;      /* arena kernel argument so that plane reads are global_load, not flat_load */
    int64_t plane_stride; /* floats per plane = pitch * h */
    int     w, h, pitch;
    int     tile_begin;   /* first wave-sized work unit of this octave in the detection launch */
};

struct PyrDesc {
    int     n_oct;
    int     levels; /* DoG search levels */
    int     L;      /* levels + 3        */
    int     total_tiles;
    int     dog_fly; /* 1: DoG planes are not stored; detection / refinement subtract the Gaussian planes they load */
    int     pad_;
    OctDesc o[PS_MAX_OCT];
};

/* sift_constants.h:56-67 ConstInfo (scalar part) + mode switches */
struct SiftConsts {
    float sigma0, sigma_k, edge_limit, threshold;
    int   max_extrema, norm_multi, norm_mode, sift_mode;
    int   grid_size;
    int   up_fac_int; /* prep_features(Descriptor*, int up_fac): truncated, sift_pyramid.cu:250 */
    int   desc_mode;  /* POPSIFT_HIP_DESC_* */
    int   det_qcap;   /* candidate queue entries the fast detection pass may use (tests shrink it) */
    int   filter_max;  /* Config::getFilterMaxExtrema(), <= 0: grid filter off */
    int   filter_mode; /* POPSIFT_HIP_FILTER_* */
    int   desc_rows;   /* patch rows k_descriptor walks per pass (tests shrink it) */
};

/* sift_extremum.h:24-33 InitialExtremum (without the grid-filter bookkeeping) */
struct InitExt {
    float xpos, ypos;
    int   lpos;
    float sigma;
    int   cell;
};

/* sift_extremum.h:40-51 Extremum */
struct Ext {
    float xpos, ypos;
    int   lpos;
    float sigma;
    int   octave;
    int   num_ori;
    int   idx_ori;
    float orientation[POPSIFT_HIP_ORI_MAX];
};

/* Everything about one (extremum, orientation) that the loop-descriptor kernel would otherwise derive, wave-uniformly and
 * once per descriptor, from the extremum, its octave and its rotation: written by k_scan_apply with one LANE per extremum
 * (same expressions, same values), read by k_descriptor as three 16-byte loads.  48 bytes. */
struct DescRec {
    float        x, y;         /* position in the octave */
    float        crsbp, srsbp; /* cos, sin of the orientation / (DESC_MAGNIFY * sigma): pixel -> cell units */
    float        ang_bins;     /* orientation in descriptor bins (pi / 4) */
    float        fscale;       /* 2^fbits: fixed-point scale of the histogram */
    unsigned int xymin, xymax; /* patch bounding box, two signed 16-bit halves each (x low, y high) */
    unsigned int off_lo, off_hi; /* offset of the Gaussian plane in the arena, in floats (64 bits) */
    unsigned int misc;         /* pitch (16 bits) | fbits << 16 | valid << 24 */
    unsigned int pad;
};

static_assert(sizeof(DescRec) == 48, "three 16-byte loads");

/* sift_pyramid.h:22-37 ExtremaCounters, device resident */
struct Counters {
    int ext_ct[PS_MAX_OCT];
    int ori_ct[PS_MAX_OCT];
    int ext_ps[PS_MAX_OCT];
    int ori_ps[PS_MAX_OCT];
    int ext_total;
    int ori_total;
    int pad[2]; /* [0] unused, [1] strips left to the slow detection pass */
    /* The detection kernel appends candidates to DET_SUBQ sub-queues, each with its own counter and its own slice of
     * the candidate buffer: a returning atomicAdd on ONE address saturates near 90/us on MI355X, which cost the kernel
     * 29 of its 106 us (one flush per strip, 5456 strips).  A sub-queue is a REGION of the image (an 8 x 8 grid over
     * every octave): refinement takes the sub-queues one after the other and appends a batch of survivors at a time,
     * so the extrema lists -- and with them the work of the orientation and descriptor kernels -- come out grouped by
     * region without a sorting pass.  Every counter sits on a cache line of its own. */
    struct {
        int n;
        int pad[31];
    } qcnt[64];
};
#define DET_SUBQ 64

/*
 * A context extracts up to PS_MAX_BATCH images of one size per submit (round 3: popsift_hip_submit_batch; a plain submit
 * is a batch of one).  Every image of the batch has a SLOT: its own arena, counters, lists and result slabs -- exactly
 * what a single-image context owned -- and every kernel of the per-image sequence is launched ONCE for the whole batch
 * with the image index in blockIdx.y, picking its slot from this table.  The table is a KERNEL ARGUMENT, passed by value
 * (2 KB of the 4 KB kernarg segment; a wave-uniform index, so the entries come in by scalar loads): pointers read from
 * the kernarg segment are known to be global, whereas pointers loaded from a table in device memory are generic to the
 * compiler and every access through them becomes a flat_load / flat_store (tried: 9871 of them in pyramid.hip).
 * So the 21 latency-bound launches of the small octaves, refinement, the scans and the slow detection pass are paid
 * once per batch instead of once per image, and the results of an image do not depend on what it is batched with.
 * Capacities (candidates, histograms, descriptors) are the same for all slots and stay kernel arguments.
 */
#define PS_MAX_BATCH POPSIFT_HIP_MAX_BATCH
struct FilterState;
struct Slot {
    float*               arena;
    Counters*            ct;
    const void*          input; /* the image (device memory), u8 or f32 */
    int2*                cand;
    int*                 ovf;
    InitExt*             iext;
    InitExt*             iext2; /* grid filter output (filter enabled only) */
    FilterState*         fstate;
    int*                 fhist;
    float*               ohist;
    Ext*                 ext;
    int*                 partial;
    int*                 map;
    float2*              rot;
    DescRec*             drec;
    popsift_hip_feature* feats;
    float*               desc;
};
struct BatchDesc {
    Slot s[PS_MAX_BATCH];
};

/* grid filter working set (filter.hip), device resident */
#define FILTER_MAX_CELLS 4096 /* grid_size <= 64 */
struct FilterState {
    int                active;   /* the reference's 10 % test passed: the filter really thins */
    int                newlimit; /* per-cell cap derived from the counts */
    int                pad[2];
    int                new_ct[PS_MAX_OCT];
    int                cell_count[FILTER_MAX_CELLS];
    int                cell_limit[FILTER_MAX_CELLS];
    int                remaining[FILTER_MAX_CELLS]; /* radix select: members still to cover */
    int                cell_mode[FILTER_MAX_CELLS]; /* 0 select, 1 keep all, 2 keep none */
    unsigned long long prefix[FILTER_MAX_CELLS];    /* radix select: digits chosen so far */
};
