/* kernels.h -- host-callable launchers of the HIP kernels (internal). */
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sift_types.h"

namespace popsift_hip {

struct Taps {
    float g[PS_GA];
};

/* One level launch; the planes are given as float offsets from the arena of the image's slot (sift_types.h, Slot): the
 * same arguments serve every image of a batch, the kernel adds its slot's arena (blockIdx.y). */
struct BlurArgs {
    int64_t src_off; /* plane l-1 (MODE 0)                     */
    int64_t dst_off; /* plane l                                */
    int64_t dog_off; /* DoG plane l-1 (MODE 0), < 0: not stored */
    int     w, h, pitch;
    int     tiles_x, tiles_y;
    /* MODE 1 (u8) / 2 (f32): the input image is Slot::input */
    int     in_w, in_h, in_pitch;
    float   shift;
    int     fast2x; /* level 0 only: the plane is the input stretched by exactly 2 with source coordinate X / 2 (2: u8 rows on 4-byte boundaries) */
    Taps    taps;
    /* level L-3 only: level 0 of the next octave = every second pixel of this plane (get_by_2_pick_every_second,
     * s_pyramid_build.cu:50-71), written by the same launch; < 0 otherwise */
    int64_t next0_off;
    int     next_pitch;
    /* level 0 only (the first launch of an image): words of the slot's Counters this launch clears, so that the image needs
     * no separate fill launch before detection; 0 otherwise */
    int     zero_words;
    /* march kernels (blur_march.hip) only: rows per segment, a multiple of 32; tiles_y = segments per strip */
    int     seg_rows;
};

/* how launch_blur picks the kernel of a plane-to-plane level launch (popsift_hip_debug_set BLUR_PATH / BLUR_SEG) */
struct BlurTune {
    int path;     /* 0: by plane size, 1: tile kernels only (pyramid.hip), 2: march kernels wherever they apply */
    int seg_rows; /* 0: chosen from the plane and the batch, else rows per segment (rounded to a multiple of 32) */
};

int        blur_tile_w();
int        blur_tile_h(int w, int h); /* 32 or 64 rows, by plane size */
hipError_t launch_blur(const BlurArgs& a, const BatchDesc& bd, int nb, int mode, int span, int tile_h, hipStream_t s,
                       BlurTune tune = BlurTune{0, 0});
/* blur_march.hip: level l >= 1 of a large plane, strips of 128 columns marched down 32 rows a step */
bool       blur_march_supported(const BlurArgs& a, int halo);
int        blur_march_seg_rows(int w, int h, int nb, int want_wgs);
hipError_t launch_blur_march(BlurArgs a, const BatchDesc& bd, int nb, int halo, int seg_rows, hipStream_t s);
/* two plane-to-plane level launches with 32-row tiles in one (small octaves) */
hipError_t launch_blur_duo(const BlurArgs& a, int span_a, const BlurArgs& b, int span_b, const BatchDesc& bd, int nb, hipStream_t s);

/* pyr_tail.hip: every octave from `first_oct` on in one launch, one workgroup per image (planes in LDS) */
#define PYR_TAIL_PAD 16             /* LDS border = the largest halo the tail takes */
#define PYR_TAIL_PLANE_FLOATS 17408 /* one LDS plane buffer incl. its border (two of them + the next octave's level 0: 154 KB) */
#define PYR_TAIL_MAX_L 10           /* Gaussian planes per octave (levels + 3) */
#define PYR_TAIL_MAX_PX 4096        /* pixels of the first plane the tail takes */
struct TailArgs {
    int     n_oct, first_oct, L;
    int     halo[PYR_TAIL_MAX_L];                     /* span - 1 of level l, rounded up to an instantiated HALO */
    float   g[PYR_TAIL_MAX_L][PYR_TAIL_PAD + 1];      /* taps of level l, zero beyond its span */
    int     w[PS_MAX_OCT], h[PS_MAX_OCT], pitch[PS_MAX_OCT];
    int64_t data_off[PS_MAX_OCT], plane_stride[PS_MAX_OCT]; /* floats from the arena base / between planes */
};
bool       pyr_tail_fits(int w, int h);
hipError_t launch_pyr_tail(const TailArgs& a, const BatchDesc& bd, int nb, hipStream_t s);

/* extrema.hip */
hipError_t launch_dog_plane(float* dog, const float* upper, const float* lower, size_t n, hipStream_t s); /* debug / test downloads */
int        extrema_units(int w, int h); /* wave-sized work units of the detection kernel */
/* Every launcher below takes the slot table `bd` (passed to the kernels by value) and the number of images nb = gridDim.y. */
/* mid: event recorded between detection and refinement (stage timing), or null */
hipError_t launch_extrema(const PyrDesc& pd, const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc, int cand_cap,
                          bool filtered, hipStream_t s, hipEvent_t mid);

/* keypoint.hip */
/* ohist: 36 floats per extremum (the raw orientation histogram), hist_cap extrema */
hipError_t launch_orientation(const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc, bool filtered, int hist_cap,
                              int blocks, hipStream_t s);
int        scan_chunk(); /* extrema per k_scan_apply workgroup */
int        scan_partials_per_chunk(); /* partial sums k_scan_local leaves per scan_chunk() extrema */
hipError_t launch_scan(const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc, bool filtered, int hist_cap,
                       int n_chunks, int desc_cap, hipStream_t s);
hipError_t launch_descriptors(const PyrDesc* d_pd, const BatchDesc& bd, int nb, const SiftConsts& sc, int desc_cap, int blocks,
                              hipStream_t s);

/* filter.hip: grid filter between refinement and orientation (s_filtergrid.cu:109-322) */
bool       filter_supported(int n_oct, int max_extrema, int grid_size);
size_t     filter_hist_bytes(int grid_size);
hipError_t launch_filter(int n_oct, const SiftConsts& sc, const BatchDesc& bd, int nb, hipStream_t s);

/* match.hip: brute-force 2-NN (features.cu:157-300) */
int        match_splits(int l_len, int r_len);
size_t     match_partial_bytes(int l_len, int n_split);
hipError_t launch_match(const float* ldesc, int l_len, const float* rdesc, int r_len, int n_split, void* partial,
                        popsift_hip_match* out, const int* rows, const int* n_rows, int first_row, int row_end,
                        hipStream_t s);
/* match_mfma.hip: matrix-core screening + exact re-rank of the survivors */
int        screen_splits(int l_len, int r_len);
size_t     screen_partial_bytes(int l_len, int n_split);
hipError_t launch_norms(const float* desc, int n, float* out, hipStream_t s);
hipError_t launch_match_screen(const float* ldesc, int l_len, const float* rdesc, int r_len, const float* lnorm,
                               const float* rnorm, int n_split, void* partial, popsift_hip_match* out, int* redo_list,
                               int* redo_count, hipStream_t s);
/* Feature records (72-byte popsift::Feature layout) with device descriptor pointers for a cloned set */
hipError_t launch_clone_features(const popsift_hip_feature* feats, int n_feat, float* desc_base, void* out, hipStream_t s);

}  // namespace popsift_hip
