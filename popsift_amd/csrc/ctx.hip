/*
 * ctx.hip -- C-ABI implementation of libpopsift_hip (include/popsift_hip.h):
 * per-GPU extraction context, Gauss / constant tables, HBM arena and the
 * per-image launch sequence.
 *
 * Replaces the host side of the reference's L3 layer:
 *   init_filter / init_constants      gauss_filter.cu:127-257, sift_constants.cu:22-53
 *   Pyramid::Pyramid / Octave::alloc  sift_pyramid.cu:108-165, sift_octave.cu:33-53
 *   Pyramid::step1 / build_pyramid    sift_pyramid.cu:226-230, s_pyramid_build.cu:460-596
 *   Pyramid::step2                    sift_pyramid.cu:232-239
 *   Pyramid::get_descriptors          sift_pyramid.cu:281-321
 * with no process-global state: every context owns its stream, arena and
 * counters, so any number of contexts can run per process / per GPU.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "devfeatures.h"
#include "kernels.h"
#include "sift_types.h"
#include "trace.h"

using namespace popsift_hip;

namespace {

constexpr int PITCH_ALIGN = 64; /* floats: rows start on 256 B */
constexpr int PROFILE_REPS = 4; /* launches per event pair in profile mode */

struct HostTables {
    float filter[POPSIFT_HIP_MAX_LEVELS * PS_GA];
    int   span[POPSIFT_HIP_MAX_LEVELS];
    float sigma[POPSIFT_HIP_MAX_LEVELS];
};

struct EventPair {
    hipEvent_t a, b;
    double     bytes;
    bool       big; /* level launch (MODE 0) with 64-row tiles */
};

}  // namespace

/* Device memory of ONE image of a batch (sift_types.h, Slot): what a single-image context owned.  Grow-only. */
struct ImageSlot {
    void*   d_input = nullptr;
    size_t  input_cap = 0;
    void*   h_input = nullptr; /* pinned staging: submit() copies the caller's image before returning */
    size_t  h_input_cap = 0;
    float*  d_arena = nullptr;
    size_t  arena_cap = 0; /* floats */
    InitExt* d_iext = nullptr;
    InitExt* d_iext2 = nullptr;      /* grid filter output (filter enabled only) */
    FilterState* d_fstate = nullptr;
    int*     d_fhist = nullptr;
    Ext*     d_ext = nullptr;
    float*   d_ohist = nullptr; /* raw orientation histograms, 36 floats per extremum (k_orientation -> k_scan_local) */
    size_t   ohist_cap = 0;     /* extrema */
    popsift_hip_feature* d_feats = nullptr;
    size_t   iext_cap = 0, iext2_cap = 0, extrec_cap = 0, feats_cap = 0;
    int*     d_map = nullptr;
    float2*  d_rot = nullptr; /* (cos, sin) of every descriptor's orientation, correctly rounded (k_scan_apply) */
    DescRec* d_drec = nullptr; /* per-descriptor constants of the loop descriptor (k_scan_apply -> k_descriptor) */
    float*   d_desc = nullptr;
    int      desc_cap = 0;
    int2*    d_cand = nullptr;
    int      cand_cap = 0;
    int*     d_partial = nullptr; /* one partial sum per scan chunk */
    size_t   partial_cap = 0;
    int*     d_ovf = nullptr;     /* detection strips handed to the slow pass */
    size_t   ovf_cap = 0;
    bool     sized = false;       /* the buffers fit the context's current geometry */
    /* second result slab (popsift_hip_fetch_begin_item): the download of this image reads one slab on copy_stream while the
     * kernels of the next batch write the other.  Invariant: alt caps <= the current slab's; fetch_begin equalises and swaps */
    popsift_hip_feature* alt_feats = nullptr;
    size_t   alt_feats_cap = 0;
    float*   alt_desc = nullptr;
    int      alt_desc_cap = 0;
    bool     moved = false; /* the finished image's results went to fetch_begin: the current slab is stale */
};

struct popsift_hip_ctx {
    int                device = 0;
    popsift_hip_params p{};
    int                levels = 3, L = 6;
    HostTables         tab{};
    SiftConsts         sc{};
    hipStream_t        stream = nullptr;
    hipEvent_t         ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t         ev_stage[POPSIFT_HIP_STAGE_COUNT + 1] = {}; /* profile mode 2: boundaries of the stages */

    /* image geometry (one for all images of a batch) */
    int  in_w = 0, in_h = 0;
    int  frozen_octaves = -1;
    bool have_image = false, finished = false;
    bool batch_ok = false; /* the last submit enqueued everything it had to: there are (or will be) results to wait for */
    int  nb = 1; /* images of the submitted batch */

    /* one slot per image of a batch; a plain submit uses slot 0 */
    ImageSlot slot[PS_MAX_BATCH];
    BatchDesc bd{}; /* the kernels' view of the slots in use (a kernel argument, passed by value) */
    PyrDesc   pd{};
    int       kp_waves = 65536; /* launch size of the keypoint kernels in waves (8 per wave slot of the device) */
    int       det_qcap = 1 << 30;  /* popsift_hip_debug_set hooks, see popsift_hip.h */
    int       desc_rows = 1 << 30;
    int       pyr_order = 0;
    BlurTune  blur_tune{0, 0}; /* BLUR_PATH / BLUR_SEG debug switches */
    int       pyr_tail = 0;    /* PYR_TAIL: 0 the smallest octaves in one launch where they fit, 1 level launches only */
    int       cand_cap_init = 1 << 20;
    bool      cand_cap_user = false;
    int       ohist_cap_init = 0;
    size_t    ext_cap = 0; /* entries every one of d_iext/d_ext/d_feats(/d_iext2) of the sized slots holds */
    /* capacities the kernels are told: the smallest over the slots of the batch (refresh_caps) */
    int       cand_cap = 0, desc_cap = 0;
    size_t    ohist_cap = 0;
    hipStream_t copy_stream = nullptr;
    bool     copy_pending = false;  /* fetch_begin downloads have not been waited for */
    unsigned long submit_seq = 0, copy_seq = 0; /* the batch now in the context / the batch whose downloads are pending */
    Counters* d_ct = nullptr; /* PS_MAX_BATCH counter blocks, one per slot */
    Counters* h_ct = nullptr; /* pinned mirror */
    PyrDesc*  d_pd = nullptr; /* device copy of pd (kernels index octaves dynamically) */
    PyrDesc*  h_pd = nullptr; /* pinned staging for d_pd */
    int       n_feat[PS_MAX_BATCH] = {}, n_desc[PS_MAX_BATCH] = {}; /* results of the finished batch */

    /* profiling */
    int                    fail_alloc_in = 0; /* test hook (popsift_hip_debug_fail_alloc): the n-th device allocation from now fails */
    int                    profile = 0;
    std::vector<EventPair> blur_events;
    size_t                 blur_events_used = 0;
    popsift_hip_report     rep{};

    char err[256] = {0};
};

namespace {

int fail(popsift_hip_ctx* c, int code, const char* fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIP_TRY(c, call)                                                                              \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return fail((c), e__ == hipErrorOutOfMemory ? POPSIFT_HIP_ERR_OOM : POPSIFT_HIP_ERR_DEVICE, \
                        "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));          \
    } while (0)

/* POP_SYNC_CHK (common/debug_macros.h:25-29): in a SYNC_CHECK build every launch of the per-image sequence is followed
 * by a stream synchronisation and an error check, so a faulting kernel is named where it was launched */
#ifdef POPSIFT_SYNC_CHECK
#define SYNC_CHK(c, what)                                                                                     \
    do {                                                                                                      \
        hipError_t e__ = hipStreamSynchronize((c)->stream);                                                   \
        if (e__ == hipSuccess) e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) {                                                                              \
            fprintf(stderr, "%s:%d sync check after %s: %s\n", __FILE__, __LINE__, what, hipGetErrorString(e__)); \
            return fail((c), POPSIFT_HIP_ERR_DEVICE, "sync check after %s: %s", what, hipGetErrorString(e__)); \
        }                                                                                                     \
    } while (0)
#else
#define SYNC_CHK(c, what) \
    do {                  \
    } while (0)
#endif

/* GaussInfo::getSpan, gauss_filter.cu:274-328 */
int span_for(int gauss_mode, float sigma)
{
    if (gauss_mode == POPSIFT_HIP_GAUSS_OPENCV_COMPUTE) {
        int span = int(roundf(2.0f * 4.0f * sigma + 1.0f)) | 1;
        span >>= 1;
        span += 1;
        return std::min<int>(span, PS_GA - 1);
    }
    return std::min<int>(ceilf(4.0f * sigma) + 1, PS_GA - 1);
}

/* init_filter (inc table; dd[0] is identical to inc[0]) gauss_filter.cu:163-181,340-372
 * and init_constants sift_constants.cu:22-31 */
void init_tables(popsift_hip_ctx* c)
{
    const popsift_hip_params& p = c->p;
    const float sigma0 = p.sigma;
    const int   levels = c->levels;
    const float initial_blur = p.assume_initial_blur ? p.initial_blur * powf(2.0f, p.upscale_factor) : 0.0f;
    HostTables& t = c->tab;
    memset(&t, 0, sizeof(t));
    t.sigma[0] = p.assume_initial_blur ? sqrtf(fabsf(sigma0 * sigma0 - initial_blur * initial_blur)) : sigma0;
    for (int lvl = 1; lvl < c->L; lvl++) {
        const float sigmaP = sigma0 * powf(2.0f, (float)(lvl - 1) / (float)levels);
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        t.sigma[lvl] = sqrtf(sigmaS * sigmaS - sigmaP * sigmaP);
    }
    for (int level = 0; level < POPSIFT_HIP_MAX_LEVELS; level++) {
        t.span[level] = std::min(span_for(p.gauss_mode, t.sigma[level]), PS_GA - 1);
        const float sig = t.sigma[level];
        const int   spn = t.span[level];
        float*      f = &t.filter[level * PS_GA];
        double      sum = 1.0;
        f[0] = 1.0f;
        for (int x = 1; x < spn; x++) {
            const float val = (float)exp(-0.5 * (pow(double(x) / sig, 2.0)));
            f[x] = val;
            sum += 2.0f * val;
        }
        for (int x = 0; x < spn; x++) f[x] = (float)(f[x] / sum);
        for (int x = spn; x < PS_GA; x++) f[x] = 0.0f;
    }
    SiftConsts& sc = c->sc;
    sc.sigma0 = sigma0;
    sc.sigma_k = powf(2.0f, 1.0f / levels);
    sc.edge_limit = p.edge_limit;
    sc.threshold = p.threshold * 0.5f * 255.0f / levels; /* Config::getPeakThreshold, sift_conf.cu:275-278 */
    sc.max_extrema = p.max_extrema;
    sc.norm_multi = p.norm_multi;
    sc.norm_mode = p.norm_mode;
    sc.sift_mode = p.sift_mode;
    sc.grid_size = p.filter_grid_size > 0 ? p.filter_grid_size : 1;
    sc.up_fac_int = (int)p.upscale_factor;
    sc.desc_mode = p.desc_mode;
    sc.filter_max = p.filter_max_extrema;
    sc.filter_mode = p.filter_sorting;
    sc.det_qcap = c->det_qcap;
    sc.desc_rows = c->desc_rows;
}

/* PopSift::private_init, popsift.cpp:89-120 */
void plan_dims(const popsift_hip_ctx* c, int w, int h, int octaves_cfg, int* n_oct, int* bw, int* bh)
{
    const float scaleFactor = 1.0f / powf(2.0f, -c->p.upscale_factor);
    int         oct = octaves_cfg;
    if (oct < 0) oct = std::max(int(floorf(logf((float)std::min(w, h)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
    oct = std::min(oct, PS_MAX_OCT);
    *n_oct = oct;
    *bw = (int)ceilf(w * scaleFactor);
    *bh = (int)ceilf(h * scaleFactor);
}

/* every device allocation of a context goes through here, so that tests can make the n-th one fail */
hipError_t ctx_malloc(popsift_hip_ctx* c, void** p, size_t bytes)
{
    if (c->fail_alloc_in > 0 && --c->fail_alloc_in == 0) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(p, bytes);
}

template <typename T>
int grow(popsift_hip_ctx* c, T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*ptr) HIP_TRY(c, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(c, ctx_malloc(c, (void**)ptr, need * sizeof(T)));
    *cap = need;
    return 0;
}

/* the capacities the kernels are told = the smallest over the slots of the batch, and the kernels' slot table */
void refresh_caps(popsift_hip_ctx* c)
{
    int    cand = 1 << 30, desc = 1 << 30;
    size_t hist = (size_t)1 << 40;
    for (int k = 0; k < c->nb; k++) {
        const ImageSlot& s = c->slot[k];
        cand = std::min(cand, s.cand_cap);
        desc = std::min(desc, s.desc_cap);
        hist = std::min(hist, s.ohist_cap);
        Slot& v = c->bd.s[k];
        v.arena = s.d_arena;
        v.ct = c->d_ct + k;
        v.cand = s.d_cand;
        v.ovf = s.d_ovf;
        v.iext = s.d_iext;
        v.iext2 = s.d_iext2;
        v.fstate = s.d_fstate;
        v.fhist = s.d_fhist;
        v.ohist = s.d_ohist;
        v.ext = s.d_ext;
        v.partial = s.d_partial;
        v.map = s.d_map;
        v.rot = s.d_rot;
        v.drec = s.d_drec;
        v.feats = s.d_feats;
        v.desc = s.d_desc;
    }
    c->cand_cap = cand;
    c->desc_cap = desc;
    c->ohist_cap = hist;
}

int slot_desc_cap(popsift_hip_ctx* c, ImageSlot& s, int need)
{
    if (need <= s.desc_cap) return 0;
    if (s.d_desc) HIP_TRY(c, hipFree(s.d_desc));
    if (s.d_map) HIP_TRY(c, hipFree(s.d_map));
    if (s.d_rot) HIP_TRY(c, hipFree(s.d_rot));
    if (s.d_drec) HIP_TRY(c, hipFree(s.d_drec));
    s.d_desc = nullptr;
    s.d_map = nullptr;
    s.d_rot = nullptr;
    s.d_drec = nullptr;
    s.desc_cap = 0;
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_desc, (size_t)need * 128 * sizeof(float)));
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_map, (size_t)need * sizeof(int)));
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_rot, (size_t)need * sizeof(float2)));
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_drec, (size_t)need * sizeof(DescRec)));
    s.desc_cap = need;
    return 0;
}

int slot_ohist_cap(popsift_hip_ctx* c, ImageSlot& s, size_t need)
{
    if (need <= s.ohist_cap) return 0;
    if (s.d_ohist) HIP_TRY(c, hipFree(s.d_ohist));
    s.d_ohist = nullptr;
    s.ohist_cap = 0;
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_ohist, need * PS_ORI_NBINS * sizeof(float)));
    s.ohist_cap = need;
    return 0;
}

int slot_cand_cap(popsift_hip_ctx* c, ImageSlot& s, int need)
{
    if (need <= s.cand_cap) return 0;
    if (s.d_cand) HIP_TRY(c, hipFree(s.d_cand));
    s.d_cand = nullptr;
    s.cand_cap = 0;
    HIP_TRY(c, ctx_malloc(c, (void**)&s.d_cand, (size_t)need * sizeof(int2)));
    s.cand_cap = need;
    return 0;
}

/* grow a list of every slot of the batch; the kernels' capacities follow whatever the outcome */
int ensure_desc_cap(popsift_hip_ctx* c, int need)
{
    int rc = 0;
    for (int k = 0; k < c->nb && !rc; k++) rc = slot_desc_cap(c, c->slot[k], need);
    refresh_caps(c);
    return rc;
}
int ensure_ohist_cap(popsift_hip_ctx* c, size_t need)
{
    int rc = 0;
    for (int k = 0; k < c->nb && !rc; k++) rc = slot_ohist_cap(c, c->slot[k], need);
    refresh_caps(c);
    return rc;
}
int ensure_cand_cap(popsift_hip_ctx* c, int need)
{
    int rc = 0;
    for (int k = 0; k < c->nb && !rc; k++) rc = slot_cand_cap(c, c->slot[k], need);
    refresh_caps(c);
    return rc;
}

/* Pyramid::Pyramid / resetDimensions: sizes for this image size and batch, grow-only buffers.
 * Failure-safe: the context forgets its geometry before anything is freed and commits the new one only after every
 * allocation and the upload of the device copy have been issued, so a submit after a failed one (ERR_OOM is a
 * recoverable status of this ABI) never finds sizes that describe buffers which no longer exist. */
int prepare_geometry(popsift_hip_ctx* c, int w, int h, int nb)
{
    const bool same = c->have_image && w == c->in_w && h == c->in_h && c->pd.n_oct > 0;
    PyrDesc    pd = c->pd;
    int        bw = c->rep.base_w, bh = c->rep.base_h;
    size_t     total = 0;
    if (!same) {
        int n_oct;
        plan_dims(c, w, h, c->frozen_octaves, &n_oct, &bw, &bh);
        if (bw < 1 || bh < 1) return fail(c, POPSIFT_HIP_ERR_INVALID, "scaled image is empty");
        /* candidates pack (x, y) into 16 bits each; the reference's Plane2D uses short dims too (plane_2d.h:257) */
        if (bw > 32767 || bh > 32767) return fail(c, POPSIFT_HIP_ERR_INVALID, "scaled image exceeds 32767 pixels per side");
        if (c->sc.filter_max > 0 && !filter_supported(n_oct, c->sc.max_extrema, c->sc.grid_size))
            return fail(c, POPSIFT_HIP_ERR_INVALID, "grid filter: grid size > 64 or octaves * max_extrema >= 2^25");
        c->frozen_octaves = n_oct; /* popsift.cpp:111: decided by the first image */

        c->in_w = c->in_h = 0;
        c->have_image = false;
        c->pd.n_oct = 0;
        c->ext_cap = 0;
        for (int k = 0; k < PS_MAX_BATCH; k++) c->slot[k].sized = false;

        memset(&pd, 0, sizeof(pd));
        pd.n_oct = n_oct;
        pd.levels = c->levels;
        pd.L = c->L;
        /* DoG planes are not stored (params.store_dog = 0): detection and refinement subtract the Gaussian planes they
         * load -- the same f32 subtraction make_dog does (s_pyramid_build.cu:74-92), so results are bit-identical, with a
         * third fewer bytes per level launch and an octave-0 working set that fits the last-level cache.  An arena then
         * holds L planes per octave plus ONE scratch plane (the size of octave 0's) into which the debug download forms a
         * DoG plane on demand; with store_dog = 1 it holds the reference's 2L-1 planes per octave.  Planes are described
         * by their float offsets from the arena base, the same for every image of a batch (OctDesc::data_off / dog_off;
         * the absolute pointers of OctDesc are not used). */
        pd.dog_fly = c->p.store_dog ? 0 : 1;
        int ow = bw, oh = bh, tiles = 0;
        for (int o = 0; o < n_oct; o++) {
            OctDesc& od = pd.o[o];
            od.w = ow;
            od.h = oh;
            od.pitch = (ow + PITCH_ALIGN - 1) / PITCH_ALIGN * PITCH_ALIGN;
            od.plane_stride = (int64_t)od.pitch * oh;
            od.tile_begin = tiles;
            tiles += extrema_units(ow, oh);
            od.data_off = (int64_t)total;
            total += (size_t)od.plane_stride * (size_t)c->L;
            if (!pd.dog_fly) {
                od.dog_off = (int64_t)total;
                total += (size_t)od.plane_stride * (size_t)(c->L - 1);
            }
            ow = (int)ceilf(ow / 2.0f); /* sift_pyramid.cu:132-133 */
            oh = (int)ceilf(oh / 2.0f);
        }
        if (pd.dog_fly) {
            for (int o = 0; o < n_oct; o++) pd.o[o].dog_off = (int64_t)total; /* download_plane(kind = 1) only */
            total += (size_t)pd.o[0].plane_stride;
        }
        pd.total_tiles = tiles;
    } else {
        const OctDesc& last = pd.o[pd.n_oct - 1];
        total = pd.dog_fly ? (size_t)pd.o[0].dog_off + (size_t)pd.o[0].plane_stride
                           : (size_t)last.dog_off + (size_t)last.plane_stride * (size_t)(c->L - 1);
    }
    const size_t need_ext = (size_t)pd.n_oct * (size_t)c->sc.max_extrema;
    for (int k = 0; k < nb; k++) {
        ImageSlot& s = c->slot[k];
        if (s.sized) continue;
        if (int rc = grow(c, &s.d_arena, &s.arena_cap, total)) return rc;
        /* each buffer keeps its own capacity: a failed grow leaves the others consistent */
        if (int rc = grow(c, &s.d_iext, &s.iext_cap, need_ext)) return rc;
        if (int rc = grow(c, &s.d_ext, &s.extrec_cap, need_ext)) return rc;
        if (int rc = grow(c, &s.d_feats, &s.feats_cap, need_ext)) return rc;
        if (c->sc.filter_max > 0) {
            if (int rc = grow(c, &s.d_iext2, &s.iext2_cap, need_ext)) return rc;
            if (!s.d_fstate) HIP_TRY(c, ctx_malloc(c, (void**)&s.d_fstate, sizeof(FilterState)));
            if (!s.d_fhist) HIP_TRY(c, ctx_malloc(c, (void**)&s.d_fhist, filter_hist_bytes(c->sc.grid_size)));
        }
        if (int rc = grow(c, &s.d_partial, &s.partial_cap, (need_ext / scan_chunk() + 2) * scan_partials_per_chunk())) return rc;
        /* sift_pyramid.cu:149: max(2*max_extrema, max_orientations) descriptors to start with */
        if (int rc = slot_desc_cap(c, s, std::max(std::max(2 * c->sc.max_extrema, c->sc.max_extrema + c->sc.max_extrema / 4), c->desc_cap)))
            return rc;
        /* candidates: 64 region slices of one buffer (extrema.hip); a default-sized buffer grows with the pyramid, so that
         * the first image of a 4K stream does not overflow a slice and re-run (a test's explicit CAND_CAP is taken as is) */
        int cand0 = std::max(c->cand_cap_init, DET_SUBQ);
        if (!c->cand_cap_user) {
            double px = 0;
            for (int o = 0; o < pd.n_oct; o++) px += (double)pd.o[o].w * pd.o[o].h;
            cand0 = std::max(cand0, (int)std::min(px / 8.0, 64.0 * 1024 * 1024));
        }
        if (int rc = slot_cand_cap(c, s, std::max(cand0, c->cand_cap))) return rc;
        /* orientation histograms: 2 * max_extrema extrema to start with (all octaves together seldom exceed one octave's
         * cap); finish() grows the buffer and re-runs the keypoint stages when an image has more */
        if (int rc = slot_ohist_cap(c, s, std::max(c->ohist_cap_init > 0 ? (size_t)c->ohist_cap_init
                                                                        : std::min(need_ext, (size_t)2 * c->sc.max_extrema),
                                                   c->slot[0].sized ? c->ohist_cap : (size_t)0)))
            return rc;
        if (int rc = grow(c, &s.d_ovf, &s.ovf_cap, (size_t)pd.total_tiles + 1)) return rc;
        s.sized = true;
    }
    c->ext_cap = need_ext;
    if (!same) {
        /* the stream is idle here (submit drains the previous batch first), so h_pd is free to reuse */
        *c->h_pd = pd;
        HIP_TRY(c, hipMemcpyAsync(c->d_pd, c->h_pd, sizeof(PyrDesc), hipMemcpyHostToDevice, c->stream));
        c->pd = pd;
        c->in_w = w;
        c->in_h = h;
        c->rep.num_octaves = pd.n_oct;
        c->rep.base_w = bw;
        c->rep.base_h = bh;
        double px = 0;
        for (int o = 0; o < pd.n_oct; o++) px += (double)pd.o[o].w * pd.o[o].h;
        c->rep.pyramid_pixels = px;
    }
    return 0;
}

int blur_launch(popsift_hip_ctx* c, const BlurArgs& a, int mode, int span, int tile_h, double alg_bytes)
{
    if (c->profile == 1) {
        if (c->blur_events_used == c->blur_events.size()) {
            EventPair ep;
            HIP_TRY(c, hipEventCreate(&ep.a));
            HIP_TRY(c, hipEventCreate(&ep.b));
            c->blur_events.push_back(ep);
        }
        EventPair& ep = c->blur_events[c->blur_events_used++];
        ep.bytes = alg_bytes * c->nb;
        ep.big = (mode == 0 && tile_h == 64);
        /* A level launch is idempotent (reads plane l-1, writes plane l and DoG l-1), so profile mode
         * brackets PROFILE_REPS back-to-back launches with one event pair: the event-to-kernel gap
         * (~4 us, as large as a small launch itself) is amortised instead of being billed per launch. */
        HIP_TRY(c, hipEventRecord(ep.a, c->stream));
        for (int rep = 0; rep < PROFILE_REPS; rep++) HIP_TRY(c, launch_blur(a, c->bd, c->nb, mode, span, tile_h, c->stream, c->blur_tune));
        HIP_TRY(c, hipEventRecord(ep.b, c->stream));
    } else {
        HIP_TRY(c, launch_blur(a, c->bd, c->nb, mode, span, tile_h, c->stream, c->blur_tune));
        SYNC_CHK(c, "k_blur_tile");
    }
    return 0;
}

/* arguments of the launch that produces plane `level` (>= 1) of octave o from plane level - 1 */
BlurArgs level_args(const popsift_hip_ctx* c, int o, int level)
{
    const PyrDesc& pd = c->pd;
    const OctDesc& od = pd.o[o];
    BlurArgs       a{};
    a.w = od.w;
    a.h = od.h;
    a.pitch = od.pitch;
    const int thd = blur_tile_h(od.w, od.h), twd = blur_tile_w();
    a.tiles_x = (od.w + twd - 1) / twd;
    a.tiles_y = (od.h + thd - 1) / thd;
    memcpy(a.taps.g, &c->tab.filter[level * PS_GA], sizeof(a.taps.g));
    a.dst_off = od.data_off + level * od.plane_stride;
    a.src_off = od.data_off + (level - 1) * od.plane_stride;
    a.dog_off = pd.dog_fly ? -1 : od.dog_off + (level - 1) * od.plane_stride;
    /* level L-3 also writes every second pixel as plane 0 of the next octave (get_by_2_pick_every_second) */
    a.next0_off = (level == pd.L - 3 && o + 1 < pd.n_oct) ? pd.o[o + 1].data_off : -1;
    a.next_pitch = (o + 1 < pd.n_oct) ? pd.o[o + 1].pitch : 0;
    return a;
}

/*
 * Pyramid::build_pyramid default branch (s_pyramid_build.cu:549-588), one stream, every launch for all images of the
 * batch (gridDim.y).  Launch order:
 *   octave 0: level 0 (from the input image), levels 1 .. L-1;
 *   octave o >= 1: levels 1 .. L-3 (level 0 came with level L-3 of octave o-1); the two last levels of octave o-1
 *   (they feed nothing but detection) ride along with levels 1 and 2 of octave o in ONE launch (k_blur_duo) when both
 *   octaves use 32-row tiles -- 3 instead of 5 dependent launches per small octave;
 *   finally the two last levels of the last octave.
 */
int enqueue_pyramid(popsift_hip_ctx* c, int is_f32, int pitch, bool aligned4)
{
    POPSIFT_RANGE("popsift_hip: pyramid");
    const PyrDesc& pd = c->pd;
    const int      L = pd.L;
    auto single = [&](int o, int level) -> int {
        const BlurArgs a = level_args(c, o, level);
        const double   px = (double)pd.o[o].w * pd.o[o].h;
        /* read plane l-1 once, write plane l (and DoG l-1 when it is stored) once: 8 (12) B / pixel */
        return blur_launch(c, a, 0, c->tab.span[level], blur_tile_h(pd.o[o].w, pd.o[o].h), (pd.dog_fly ? 8.0 : 12.0) * px);
    };
    {
        /* horiz_from_input_image, s_pyramid_build.cu:96-126 */
        const OctDesc& od = pd.o[0];
        BlurArgs       a{};
        a.w = od.w;
        a.h = od.h;
        a.pitch = od.pitch;
        const int thd = blur_tile_h(od.w, od.h), twd = blur_tile_w();
        a.tiles_x = (od.w + twd - 1) / twd;
        a.tiles_y = (od.h + thd - 1) / thd;
        memcpy(a.taps.g, &c->tab.filter[0], sizeof(a.taps.g));
        a.dst_off = od.data_off;
        a.src_off = 0;
        a.dog_off = -1;
        a.next0_off = -1;
        float shift = 0.5f;
        if (c->p.sift_mode == POPSIFT_HIP_SIFT_POPSIFT || c->p.sift_mode == POPSIFT_HIP_SIFT_VLFEAT)
            shift = 0.5f * powf(2.0f, c->p.upscale_factor - 0);
        a.in_w = c->in_w;
        a.in_h = c->in_h;
        a.in_pitch = pitch;
        a.shift = shift;
        /* weights of the linear upscale are exactly {0, 1/2}: k_blur_tile's copy / average path */
        a.fast2x = (c->p.upscale_factor == 1.0f && shift == 1.0f && od.w == 2 * c->in_w && od.h == 2 * c->in_h) ? 1 : 0;
        /* ... and u8 images whose rows all start on 4-byte boundaries: the texels are fetched as aligned dwords */
        if (a.fast2x && !is_f32 && aligned4) a.fast2x = 2;
        a.zero_words = (int)(sizeof(Counters) / sizeof(int)); /* this launch clears the images' counters (enqueue_keypoint_stages(c, true)) */
        const double bytes = (double)c->in_w * c->in_h * (is_f32 ? 4 : 1) + 4.0 * (double)od.w * od.h;
        if (int rc = blur_launch(c, a, is_f32 ? 2 : 1, c->tab.span[0], thd, bytes)) return rc;
    }
    /* Level 1 of octave 1 reads what level L-3 of octave 0 has just written (every second pixel, 1/4 of a plane): launched
     * right behind it, that plane still sits in the L2s; after levels L-2 and L-1 of octave 0 (2 x 66 MB through the
     * caches) it came from HBM, and the launch took 14 us instead of 8 (round 2's "octave-1 anomaly"). */
    /* The smallest octaves -- from the first one whose plane fits the LDS of one workgroup -- are built by ONE launch
     * (pyr_tail.hip) instead of three dependent launches per octave; `n_front` octaves take the level launches. */
    int n_front = pd.n_oct;
    TailArgs ta{};
    if (c->pyr_tail != 1 && c->profile != 1 && pd.dog_fly && L <= PYR_TAIL_MAX_L) {
        bool taps_ok = true;
        for (int l = 1; l < L; l++) taps_ok = taps_ok && c->tab.span[l] - 1 <= PYR_TAIL_PAD && c->tab.span[l] >= 2;
        int first = pd.n_oct;
        for (int o = pd.n_oct - 1; o >= 1 && taps_ok && pyr_tail_fits(pd.o[o].w, pd.o[o].h); o--) first = o;
        if (first < pd.n_oct) {
            n_front = first;
            ta.n_oct = pd.n_oct;
            ta.first_oct = first;
            ta.L = L;
            for (int l = 1; l < L; l++) {
                ta.halo[l] = c->tab.span[l] - 1;
                for (int k = 0; k <= PYR_TAIL_PAD; k++) ta.g[l][k] = k < c->tab.span[l] ? c->tab.filter[l * PS_GA + k] : 0.0f;
            }
            for (int o = 0; o < pd.n_oct; o++) {
                ta.w[o] = pd.o[o].w;
                ta.h[o] = pd.o[o].h;
                ta.pitch[o] = pd.o[o].pitch;
                ta.data_off[o] = pd.o[o].data_off;
                ta.plane_stride[o] = pd.o[o].plane_stride;
            }
        }
    }
    const bool early1 = c->pyr_order == 1 && n_front >= 2 && L - 3 >= 1;
    for (int level = 1; level < L; level++) {
        if (int rc = single(0, level)) return rc;
        if (early1 && level == L - 3)
            if (int rc = single(1, 1)) return rc;
    }
    for (int o = 1; o < n_front; o++) {
        /* per-launch profiling keeps one kernel per event pair */
        const bool pair = c->profile != 1 && o >= 2 && blur_tile_h(pd.o[o].w, pd.o[o].h) == 32 &&
                          blur_tile_h(pd.o[o - 1].w, pd.o[o - 1].h) == 32;
        for (int level = 1; level <= L - 3; level++) {
            if (early1 && o == 1 && level == 1) continue; /* launched behind level L-3 of octave 0 */
            const int trail = L - 3 + level; /* L-2, L-1 of the octave before */
            if (pair && level <= 2) {
                const BlurArgs a = level_args(c, o, level), b = level_args(c, o - 1, trail);
                HIP_TRY(c, launch_blur_duo(a, c->tab.span[level], b, c->tab.span[trail], c->bd, c->nb, c->stream));
                SYNC_CHK(c, "k_blur_duo");
            } else {
                if (int rc = single(o, level)) return rc;
                if (!pair && level <= 2 && o >= 2)
                    if (int rc = single(o - 1, trail)) return rc;
            }
        }
    }
    if (n_front < pd.n_oct) {
        /* reads level 0 of octave n_front, which level L-3 of octave n_front - 1 has just written */
        HIP_TRY(c, launch_pyr_tail(ta, c->bd, c->nb, c->stream));
        SYNC_CHK(c, "k_pyr_tail");
    }
    if (n_front >= 2)
        for (int level = L - 2; level < L; level++)
            if (int rc = single(n_front - 1, level)) return rc;
    return 0;
}

/* Pyramid::step2 + prep_features: extrema -> orientation -> scan -> descriptors -> features */
InitExt* final_iext(popsift_hip_ctx* c, int k = 0) { return c->sc.filter_max > 0 ? c->slot[k].d_iext2 : c->slot[k].d_iext; }

/* counters_cleared: the level-0 launch of this batch has zeroed the counters (submit); re-runs clear them here */
int enqueue_keypoint_stages(popsift_hip_ctx* c, bool counters_cleared = false)
{
    POPSIFT_RANGE("popsift_hip: keypoint stages");
    const bool stages = (c->profile == 2);
    const bool filtered = c->sc.filter_max > 0;
    auto       mark = [&](int k) -> hipError_t { return stages ? hipEventRecord(c->ev_stage[k], c->stream) : hipSuccess; };
    if (!counters_cleared) HIP_TRY(c, hipMemsetAsync(c->d_ct, 0, sizeof(Counters) * (size_t)c->nb, c->stream));
    HIP_TRY(c, mark(POPSIFT_HIP_STAGE_DETECT)); /* = end of the pyramid stage */
    HIP_TRY(c, launch_extrema(c->pd, c->d_pd, c->bd, c->nb, c->sc, c->cand_cap, filtered, c->stream,
                              stages ? c->ev_stage[POPSIFT_HIP_STAGE_REFINE] : nullptr));
    SYNC_CHK(c, "k_detect / k_refine");
    if (filtered) {
        /* Pyramid::orientation's filter hook (s_orientation.cu:353-367); the 10 % test is taken on the device */
        HIP_TRY(c, launch_filter(c->pd.n_oct, c->sc, c->bd, c->nb, c->stream));
        SYNC_CHK(c, "grid filter");
    }
    HIP_TRY(c, mark(POPSIFT_HIP_STAGE_ORIENTATION));
    HIP_TRY(c, launch_orientation(c->d_pd, c->bd, c->nb, c->sc, filtered, (int)c->ohist_cap, c->kp_waves, c->stream));
    SYNC_CHK(c, "k_orientation");
    HIP_TRY(c, mark(POPSIFT_HIP_STAGE_SCAN));
    const int n_chunks = (int)(((size_t)c->pd.n_oct * c->sc.max_extrema + scan_chunk() - 1) / scan_chunk());
    HIP_TRY(c, launch_scan(c->d_pd, c->bd, c->nb, c->sc, filtered, (int)c->ohist_cap, std::max(n_chunks, 1), c->desc_cap, c->stream));
    SYNC_CHK(c, "k_scan_local / k_scan_apply");
    HIP_TRY(c, mark(POPSIFT_HIP_STAGE_DESCRIPTOR));
    HIP_TRY(c, launch_descriptors(c->d_pd, c->bd, c->nb, c->sc, c->desc_cap, c->kp_waves, c->stream));
    SYNC_CHK(c, "descriptor kernel");
    HIP_TRY(c, mark(POPSIFT_HIP_STAGE_COUNT));
    HIP_TRY(c, hipMemcpyAsync(c->h_ct, c->d_ct, sizeof(Counters) * (size_t)c->nb, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

/* where the images of a batch lie: kind = POPSIFT_HIP_IMG_* */
int submit_common(popsift_hip_ctx* c, const void* const* imgs, int nb, int kind, int w, int h, int pitch)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    POPSIFT_RANGE("popsift_hip: submit");
    if (!imgs || nb < 1 || nb > PS_MAX_BATCH || w <= 0 || h <= 0 || pitch < w || kind < 0 || kind > POPSIFT_HIP_IMG_PINNED_F32)
        return fail(c, POPSIFT_HIP_ERR_INVALID, "bad image arguments");
    for (int k = 0; k < nb; k++)
        if (!imgs[k]) return fail(c, POPSIFT_HIP_ERR_INVALID, "bad image arguments");
    const int is_f32 = kind & 1, where = kind >> 1; /* 0 host, 1 device, 2 pinned host */
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->have_image && !c->finished) {
        /* one batch in flight per context: drain the previous one */
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    /* from here until every enqueue below has succeeded the context holds NO results: a submit that fails half-way (an
     * allocation for a new slot of a larger batch of the same size, a pinned staging buffer, a launch) must not leave
     * wait / fetch with the previous batch's counts for slots whose kernels never ran */
    c->batch_ok = false;
    c->finished = false;
    if (int rc = prepare_geometry(c, w, h, nb)) return rc;
    c->nb = nb;
    refresh_caps(c);
    const size_t esz = is_f32 ? 4 : 1;
    int          dpitch = pitch;
    bool         aligned4 = (pitch & 3) == 0;
    for (int k = 0; k < nb; k++) {
        ImageSlot& s = c->slot[k];
        if (where == 1) {
            c->bd.s[k].input = imgs[k];
            aligned4 = aligned4 && ((uintptr_t)imgs[k] & 3) == 0;
            continue;
        }
        size_t cap = s.input_cap;
        char*  buf = (char*)s.d_input;
        const int rc_in = grow(c, &buf, &cap, (size_t)w * h * esz);
        s.d_input = buf; /* also after a failed grow, which has freed the old buffer */
        s.input_cap = cap;
        if (rc_in) return rc_in;
        const size_t bytes = (size_t)w * h * esz;
        if (where == 2) {
            /* page-locked memory of the caller, valid until wait(): uploaded from where it lies */
            HIP_TRY(c, hipMemcpy2DAsync(s.d_input, (size_t)w * esz, imgs[k], (size_t)pitch * esz, (size_t)w * esz, (size_t)h,
                                        hipMemcpyHostToDevice, c->stream));
        } else {
            /* Like Image::load (s_image.cu:71-79) the caller's buffer is copied into pinned memory before
             * this call returns: the caller may free or reuse it immediately (popsift.cpp:245-247), and an
             * async copy straight from pageable memory would read it later. */
            if (bytes > s.h_input_cap) {
                if (s.h_input) HIP_TRY(c, hipHostFree(s.h_input));
                s.h_input = nullptr;
                s.h_input_cap = 0;
                HIP_TRY(c, hipHostMalloc(&s.h_input, bytes, hipHostMallocDefault));
                s.h_input_cap = bytes;
            }
            for (int y = 0; y < h; y++)
                memcpy((char*)s.h_input + (size_t)y * w * esz, (const char*)imgs[k] + (size_t)y * pitch * esz, (size_t)w * esz);
            HIP_TRY(c, hipMemcpyAsync(s.d_input, s.h_input, bytes, hipMemcpyHostToDevice, c->stream));
        }
        c->bd.s[k].input = s.d_input;
        dpitch = w;
    }
    if (where != 1) aligned4 = (dpitch & 3) == 0; /* hipMalloc'd buffers are aligned */
    c->blur_events_used = 0;
    HIP_TRY(c, hipEventRecord(c->ev_begin, c->stream));
    if (c->profile == 2) HIP_TRY(c, hipEventRecord(c->ev_stage[POPSIFT_HIP_STAGE_PYRAMID], c->stream));
    if (int rc = enqueue_pyramid(c, is_f32, dpitch, aligned4)) return rc;
    if (int rc = enqueue_keypoint_stages(c, true)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_end, c->stream));
    c->have_image = true;
    c->finished = false;
    c->batch_ok = true;
    c->submit_seq++;
    for (ImageSlot& sl : c->slot) sl.moved = false;
    return 0;
}

int finish(popsift_hip_ctx* c)
{
    if (!c->have_image || !c->batch_ok) return fail(c, POPSIFT_HIP_ERR_STATE, c->have_image ? "the last submit failed" : "no image submitted");
    if (c->finished) return 0;
    POPSIFT_RANGE("popsift_hip: wait");
    HIP_TRY(c, hipSetDevice(c->device));
    bool rerun = false, fits = false;
    for (int attempt = 0; attempt < 8; attempt++) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        /* the largest need over the images of the batch */
        int  ori_max = 0, qmax = 0;
        long ext_max = 0;
        for (int k = 0; k < c->nb; k++) {
            const Counters& h = c->h_ct[k];
            ori_max = std::max(ori_max, h.ori_total);
            for (int q = 0; q < DET_SUBQ; q++) qmax = std::max(qmax, h.qcnt[q].n);
            long ext_sum = 0; /* ext_total is written by the scan, which may have run on a clipped list: recount */
            for (int o = 0; o < c->pd.n_oct; o++) ext_sum += std::min(h.ext_ct[o], c->sc.max_extrema);
            ext_max = std::max(ext_max, ext_sum);
        }
        const bool desc_short = ori_max > c->desc_cap;
        const bool cand_short = qmax > c->cand_cap / DET_SUBQ;
        const bool hist_short = (size_t)ext_max > c->ohist_cap;
        if (!desc_short && !cand_short && !hist_short) {
            fits = true;
            break;
        }
        /* more candidates / descriptors than the buffers hold (the reference reallocates between
         * stages, sift_pyramid.cu:179-209): grow and redo the keypoint stages of this batch */
        if (desc_short)
            if (int rc = ensure_desc_cap(c, ori_max + ori_max / 8 + 1024)) return rc;
        if (cand_short)
            if (int rc = ensure_cand_cap(c, DET_SUBQ * (qmax + qmax / 8 + 64))) return rc;
        if (hist_short)
            if (int rc = ensure_ohist_cap(c, (size_t)ext_max + (size_t)ext_max / 8 + 1024)) return rc;
        if (int rc = enqueue_keypoint_stages(c)) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_end, c->stream));
        rerun = true;
    }
    /* every re-run sizes the buffers from the counts of the run before, so the second attempt fits unless the counts
     * themselves were clipped; eight rounds of growing without fitting is a defect, not a result to hand out clipped */
    if (!fits) return fail(c, POPSIFT_HIP_ERR_DEVICE, "the keypoint buffers still do not fit after 8 grow-and-rerun rounds");
    popsift_hip_report& r = c->rep;
    for (int k = 0; k < c->nb; k++) {
        c->n_feat[k] = c->h_ct[k].ext_total;
        c->n_desc[k] = std::min(c->h_ct[k].ori_total, c->desc_cap);
    }
    /* the report describes image 0 of the batch: per-octave descriptor counts from the octave start offsets the scan
     * left (dct.ori_ps / ori_ct) */
    Counters& h0 = c->h_ct[0];
    int       next = h0.ori_total;
    for (int o = PS_MAX_OCT - 1; o >= 0; o--) {
        if (h0.ext_ct[o] > 0) {
            h0.ori_ct[o] = next - h0.ori_ps[o];
            next = h0.ori_ps[o];
        } else {
            h0.ori_ct[o] = 0;
            h0.ori_ps[o] = next;
        }
    }
    for (int o = 0; o < PS_MAX_OCT; o++) {
        r.ext_ct[o] = h0.ext_ct[o];
        r.ori_ct[o] = h0.ori_ct[o];
    }
    r.ext_total = c->n_feat[0];
    r.ori_total = c->n_desc[0];
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) r.ms_device = ms;
    for (int k = 0; k < 8; k++) r.ms_stage[k] = 0.0f;
    /* after a re-run the stage events of the keypoint stages are the re-run's and the pyramid's event pair spans the
     * first attempt as well: no stage times then */
    if (c->profile == 2 && !rerun)
        for (int k = 0; k < POPSIFT_HIP_STAGE_COUNT; k++) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, c->ev_stage[k], c->ev_stage[k + 1]) == hipSuccess) r.ms_stage[k] = t;
        }
    r.ms_blur = 0.0f;
    r.blur_launches = 0;
    r.blur_alg_bytes = 0.0;
    r.ms_big = 0.0f;
    r.big_launches = 0;
    r.big_alg_bytes = 0.0;
    for (size_t i = 0; i < c->blur_events_used; i++) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, c->blur_events[i].a, c->blur_events[i].b) == hipSuccess) {
            t /= (float)PROFILE_REPS;
            r.ms_blur += t;
            r.blur_launches++;
            r.blur_alg_bytes += c->blur_events[i].bytes;
            if (c->blur_events[i].big) {
                r.ms_big += t;
                r.big_launches++;
                r.big_alg_bytes += c->blur_events[i].bytes;
            }
        }
    }
    c->finished = true;
    return 0;
}

/* waits for the download popsift_hip_fetch_begin started, if one is pending */
int drain_copy(popsift_hip_ctx* c)
{
    if (!c->copy_pending) return 0;
    c->copy_pending = false;
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    return 0;
}

/* results of the finished batch are readable from the current slabs (not yet handed to fetch_begin) */
int results_here(popsift_hip_ctx* c, int k = 0)
{
    if (int rc = finish(c)) return rc;
    if (k < 0 || k >= c->nb) return fail(c, POPSIFT_HIP_ERR_INVALID, "the batch has %d images", c->nb);
    if (c->slot[k].moved)
        return fail(c, POPSIFT_HIP_ERR_STATE, "the results of this image were handed to popsift_hip_fetch_begin");
    return 0;
}

/* planes of image 0 of the batch (debug / parity hooks) */
int plane_ptr(popsift_hip_ctx* c, int octave, int kind, int level, float** p, const OctDesc** odp)
{
    if (!c || !c->have_image) return POPSIFT_HIP_ERR_STATE;
    if (octave < 0 || octave >= c->pd.n_oct || level < 0) return fail(c, POPSIFT_HIP_ERR_INVALID, "bad octave/level");
    const OctDesc& od = c->pd.o[octave];
    float*         arena = c->slot[0].d_arena;
    if (kind == 0 && level < c->L)
        *p = arena + od.data_off + level * od.plane_stride;
    else if (kind == 1 && level < c->L - 1)
        *p = arena + od.dog_off + (c->pd.dog_fly ? 0 : level * od.plane_stride); /* not stored: the scratch plane */
    else
        return fail(c, POPSIFT_HIP_ERR_INVALID, "bad plane kind/level");
    *odp = &od;
    return 0;
}

}  // namespace

extern "C" {

void popsift_hip_default_params(popsift_hip_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->octaves = -1;
    p->levels = 3;
    p->sigma = 1.6f;
    p->edge_limit = 10.0f;
    p->threshold = 0.04f;
    p->upscale_factor = 1.0f;
    p->sift_mode = POPSIFT_HIP_SIFT_POPSIFT;
    p->gauss_mode = POPSIFT_HIP_GAUSS_VLFEAT_COMPUTE;
    p->desc_mode = POPSIFT_HIP_DESC_LOOP;
    p->norm_mode = POPSIFT_HIP_NORM_ROOTSIFT;
    p->norm_multi = 0;
    p->max_extrema = 100000;
    p->assume_initial_blur = 1;
    p->initial_blur = 0.5f;
    p->filter_grid_size = 2;
    p->filter_max_extrema = -1;
    p->filter_sorting = POPSIFT_HIP_FILTER_RANDOM;
}

const char* popsift_hip_version(void) { return "popsift_hip 0.1 (gfx950, wave64)"; }

const char* popsift_hip_strerror(int status)
{
    switch (status) {
    case POPSIFT_HIP_OK: return "ok";
    case POPSIFT_HIP_ERR_INVALID: return "invalid argument or unsupported mode";
    case POPSIFT_HIP_ERR_DEVICE: return "HIP runtime error";
    case POPSIFT_HIP_ERR_NO_DEVICE: return "no usable GPU";
    case POPSIFT_HIP_ERR_OOM: return "out of memory";
    case POPSIFT_HIP_ERR_STATE: return "call sequence error";
    case POPSIFT_HIP_ERR_TOO_SMALL: return "buffer too small";
    }
    return "unknown status";
}

const char* popsift_hip_last_error(const popsift_hip_ctx* ctx) { return ctx ? ctx->err : ""; }

int popsift_hip_device_count(int* count)
{
    if (!count) return POPSIFT_HIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return POPSIFT_HIP_ERR_NO_DEVICE;
    }
    *count = n;
    return n > 0 ? POPSIFT_HIP_OK : POPSIFT_HIP_ERR_NO_DEVICE;
}

int popsift_hip_get_device_info(int device, popsift_hip_device_info* out)
{
    if (!out) return POPSIFT_HIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return POPSIFT_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return POPSIFT_HIP_ERR_INVALID;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s (%s)", pr.name, pr.gcnArchName);
    out->arch_major = pr.major;
    out->arch_minor = pr.minor;
    out->total_mem = pr.totalGlobalMem;
    out->lds_per_block = pr.sharedMemPerBlock;
    out->wave_size = pr.warpSize;
    out->max_threads_per_block = pr.maxThreadsPerBlock;
    out->max_threads_per_cu = pr.maxThreadsPerMultiProcessor;
    for (int i = 0; i < 3; i++) {
        out->max_block[i] = pr.maxThreadsDim[i];
        out->max_grid[i] = pr.maxGridSize[i];
    }
    out->cu_count = pr.multiProcessorCount;
    out->concurrent_kernels = pr.concurrentKernels;
    out->can_map_host = pr.canMapHostMemory;
    out->unified_addressing = 1; /* HIP on ROCm: one virtual address space for host and device */
    return POPSIFT_HIP_OK;
}

int popsift_hip_device_numa_node(int device, int* node)
{
    if (!node) return POPSIFT_HIP_ERR_INVALID;
    *node = -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return POPSIFT_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return POPSIFT_HIP_ERR_INVALID;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    for (char* q = bus; *q; q++)
        if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a'); /* sysfs spells the address in lower case */
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    if (FILE* f = fopen(path, "r")) {
        int v = -1;
        if (fscanf(f, "%d", &v) == 1) *node = v;
        fclose(f);
    }
    return POPSIFT_HIP_OK;
}

int popsift_hip_ctx_create(int device, const popsift_hip_params* p, popsift_hip_ctx** out)
{
    if (!p || !out) return POPSIFT_HIP_ERR_INVALID;
    *out = nullptr;
    /* gauss_filter.cu:131-144: sigma > 2 or too many levels is fatal in the reference */
    if (!(p->sigma > 0.0f) || p->sigma > 2.0f) return POPSIFT_HIP_ERR_INVALID;
    if (p->levels > POPSIFT_HIP_MAX_LEVELS - 3) return POPSIFT_HIP_ERR_INVALID;
    if (p->gauss_mode != POPSIFT_HIP_GAUSS_VLFEAT_COMPUTE && p->gauss_mode != POPSIFT_HIP_GAUSS_OPENCV_COMPUTE)
        return POPSIFT_HIP_ERR_INVALID;
    if (p->desc_mode < POPSIFT_HIP_DESC_LOOP || p->desc_mode > POPSIFT_HIP_DESC_NOTILE) return POPSIFT_HIP_ERR_INVALID;
    if (p->sift_mode < 0 || p->sift_mode > 2 || p->norm_mode < 0 || p->norm_mode > 1) return POPSIFT_HIP_ERR_INVALID;
    if (p->max_extrema < 1 || !(p->edge_limit > 0.0f)) return POPSIFT_HIP_ERR_INVALID;
    if (p->filter_max_extrema > 0 &&
        (p->filter_grid_size < 1 || p->filter_grid_size > 64 || p->filter_sorting < 0 || p->filter_sorting > 2))
        return POPSIFT_HIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return POPSIFT_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return POPSIFT_HIP_ERR_INVALID;

    popsift_hip_ctx* c = new (std::nothrow) popsift_hip_ctx();
    if (!c) return POPSIFT_HIP_ERR_OOM;
    c->device = device;
    c->p = *p;
    c->levels = std::max(2, p->levels); /* popsift.cpp:71 */
    c->L = c->levels + 3;
    c->frozen_octaves = p->octaves;
    init_tables(c);
    int rc = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        {
            int cus = 0;
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            c->kp_waves = std::max(cus, 8) * 32 * 8; /* a multiple of 32 */
        }
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreate(&c->ev_begin));
        HIP_TRY(c, hipEventCreate(&c->ev_end));
        for (int k = 0; k <= POPSIFT_HIP_STAGE_COUNT; k++) HIP_TRY(c, hipEventCreate(&c->ev_stage[k]));
        HIP_TRY(c, hipMalloc((void**)&c->d_ct, sizeof(Counters) * PS_MAX_BATCH));
        HIP_TRY(c, hipMalloc((void**)&c->d_pd, sizeof(PyrDesc)));
        HIP_TRY(c, hipHostMalloc((void**)&c->h_pd, sizeof(PyrDesc), hipHostMallocDefault));
        HIP_TRY(c, hipHostMalloc((void**)&c->h_ct, sizeof(Counters) * PS_MAX_BATCH, hipHostMallocDefault));
        memset(c->h_ct, 0, sizeof(Counters) * PS_MAX_BATCH);
        return 0;
    }();
    if (rc) {
        popsift_hip_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return POPSIFT_HIP_OK;
}

int popsift_hip_ctx_destroy(popsift_hip_ctx* c)
{
    if (!c) return POPSIFT_HIP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& ep : c->blur_events) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (int k = 0; k <= POPSIFT_HIP_STAGE_COUNT; k++)
        if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    for (ImageSlot& sl : c->slot) {
        if (sl.d_input) (void)hipFree(sl.d_input);
        if (sl.h_input) (void)hipHostFree(sl.h_input);
        if (sl.d_arena) (void)hipFree(sl.d_arena);
        if (sl.d_iext) (void)hipFree(sl.d_iext);
        if (sl.d_iext2) (void)hipFree(sl.d_iext2);
        if (sl.d_fstate) (void)hipFree(sl.d_fstate);
        if (sl.d_fhist) (void)hipFree(sl.d_fhist);
        if (sl.d_ext) (void)hipFree(sl.d_ext);
        if (sl.d_ohist) (void)hipFree(sl.d_ohist);
        if (sl.d_feats) (void)hipFree(sl.d_feats);
        if (sl.d_map) (void)hipFree(sl.d_map);
        if (sl.d_rot) (void)hipFree(sl.d_rot);
        if (sl.d_drec) (void)hipFree(sl.d_drec);
        if (sl.d_desc) (void)hipFree(sl.d_desc);
        if (sl.d_cand) (void)hipFree(sl.d_cand);
        if (sl.d_partial) (void)hipFree(sl.d_partial);
        if (sl.d_ovf) (void)hipFree(sl.d_ovf);
        if (sl.alt_feats) (void)hipFree(sl.alt_feats);
        if (sl.alt_desc) (void)hipFree(sl.alt_desc);
    }
    if (c->d_ct) (void)hipFree(c->d_ct);
    if (c->d_pd) (void)hipFree(c->d_pd);
    if (c->h_pd) (void)hipHostFree(c->h_pd);
    if (c->h_ct) (void)hipHostFree(c->h_ct);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
    return POPSIFT_HIP_OK;
}

int popsift_hip_get_gauss_table(const popsift_hip_ctx* c, float* filter, int* span, float* sigma, int* n_levels)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (filter) memcpy(filter, c->tab.filter, sizeof(float) * (size_t)c->L * PS_GA);
    if (span) memcpy(span, c->tab.span, sizeof(int) * (size_t)c->L);
    if (sigma) memcpy(sigma, c->tab.sigma, sizeof(float) * (size_t)c->L);
    if (n_levels) *n_levels = c->L;
    return POPSIFT_HIP_OK;
}

int popsift_hip_submit_u8(popsift_hip_ctx* c, const uint8_t* img, int w, int h, int pitch)
{
    const void* one = img;
    return submit_common(c, &one, 1, POPSIFT_HIP_IMG_HOST_U8, w, h, pitch);
}
int popsift_hip_submit_f32(popsift_hip_ctx* c, const float* img, int w, int h, int pitch)
{
    const void* one = img;
    return submit_common(c, &one, 1, POPSIFT_HIP_IMG_HOST_F32, w, h, pitch);
}
int popsift_hip_submit_pinned_u8(popsift_hip_ctx* c, const uint8_t* img, int w, int h, int pitch)
{
    const void* one = img;
    return submit_common(c, &one, 1, POPSIFT_HIP_IMG_PINNED_U8, w, h, pitch);
}
int popsift_hip_submit_pinned_f32(popsift_hip_ctx* c, const float* img, int w, int h, int pitch)
{
    const void* one = img;
    return submit_common(c, &one, 1, POPSIFT_HIP_IMG_PINNED_F32, w, h, pitch);
}
int popsift_hip_submit_dev_u8(popsift_hip_ctx* c, const void* d_img, int w, int h, int pitch)
{
    return submit_common(c, &d_img, 1, POPSIFT_HIP_IMG_DEV_U8, w, h, pitch);
}
int popsift_hip_submit_dev_f32(popsift_hip_ctx* c, const void* d_img, int w, int h, int pitch)
{
    return submit_common(c, &d_img, 1, POPSIFT_HIP_IMG_DEV_F32, w, h, pitch);
}
int popsift_hip_submit_batch(popsift_hip_ctx* c, const void* const* imgs, int n, int kind, int w, int h, int pitch)
{
    return submit_common(c, imgs, n, kind, w, h, pitch);
}

int popsift_hip_wait_batch(popsift_hip_ctx* c, int* n_images, int* n_features, int* n_descriptors)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = finish(c)) return rc;
    if (n_images) *n_images = c->nb;
    for (int k = 0; k < c->nb; k++) {
        if (n_features) n_features[k] = c->n_feat[k];
        if (n_descriptors) n_descriptors[k] = c->n_desc[k];
    }
    return POPSIFT_HIP_OK;
}

int popsift_hip_fetch_item(popsift_hip_ctx* c, int k, popsift_hip_feature* feats, size_t feats_cap, float* desc, size_t desc_cap)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = results_here(c, k)) return rc;
    POPSIFT_RANGE("popsift_hip: fetch");
    const size_t nf = (size_t)c->n_feat[k], nd = (size_t)c->n_desc[k];
    if ((nf && !feats) || (nd && !desc)) return fail(c, POPSIFT_HIP_ERR_INVALID, "null output buffer");
    if (feats_cap < nf || desc_cap < nd * 128) return fail(c, POPSIFT_HIP_ERR_TOO_SMALL, "output buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    const ImageSlot& sl = c->slot[k];
    if (nf) HIP_TRY(c, hipMemcpyAsync(feats, sl.d_feats, nf * sizeof(popsift_hip_feature), hipMemcpyDeviceToHost, c->stream));
    if (nd) HIP_TRY(c, hipMemcpyAsync(desc, sl.d_desc, nd * 128 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return POPSIFT_HIP_OK;
}

int popsift_hip_results_dev_item(popsift_hip_ctx* c, int k, const void** d_feats, const void** d_desc)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = results_here(c, k)) return rc;
    if (d_feats) *d_feats = c->slot[k].d_feats;
    if (d_desc) *d_desc = c->slot[k].d_desc;
    return POPSIFT_HIP_OK;
}

int popsift_hip_wait(popsift_hip_ctx* c, int* n_features, int* n_descriptors)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = finish(c)) return rc;
    if (n_features) *n_features = c->rep.ext_total;
    if (n_descriptors) *n_descriptors = c->rep.ori_total;
    return POPSIFT_HIP_OK;
}

int popsift_hip_fetch(popsift_hip_ctx* c, popsift_hip_feature* feats, size_t feats_cap, float* desc, size_t desc_cap)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = results_here(c)) return rc;
    POPSIFT_RANGE("popsift_hip: fetch");
    const size_t nf = (size_t)c->rep.ext_total, nd = (size_t)c->rep.ori_total;
    if ((nf && !feats) || (nd && !desc)) return fail(c, POPSIFT_HIP_ERR_INVALID, "null output buffer");
    if (feats_cap < nf || desc_cap < nd * 128) return fail(c, POPSIFT_HIP_ERR_TOO_SMALL, "output buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    if (nf) HIP_TRY(c, hipMemcpyAsync(feats, c->slot[0].d_feats, nf * sizeof(popsift_hip_feature), hipMemcpyDeviceToHost, c->stream));
    if (nd) HIP_TRY(c, hipMemcpyAsync(desc, c->slot[0].d_desc, nd * 128 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return POPSIFT_HIP_OK;
}

int popsift_hip_fetch_begin_item(popsift_hip_ctx* c, int k, popsift_hip_feature* feats, size_t feats_cap, float* desc,
                                 size_t desc_cap)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = results_here(c, k)) return rc;
    POPSIFT_RANGE("popsift_hip: fetch_begin");
    const size_t nf = (size_t)c->n_feat[k], nd = (size_t)c->n_desc[k];
    if ((nf && !feats) || (nd && !desc)) return fail(c, POPSIFT_HIP_ERR_INVALID, "null output buffer");
    if (feats_cap < nf || desc_cap < nd * 128) return fail(c, POPSIFT_HIP_ERR_TOO_SMALL, "output buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    /* the other slabs may still be the source of the downloads of the batch before this one */
    if (c->copy_pending && c->copy_seq != c->submit_seq)
        if (int rc = drain_copy(c)) return rc;
    /* The copy stream is made on first use: the runtime deals its few hardware queues to streams in the order they
     * are created, so a second stream in EVERY context -- used or not -- takes queues from the streams that do the work
     * (four active contexts next to sixteen idle ones: 7.8 -> 6.5 Gpix/s on the sparse workload). */
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    /* nothing has been issued or swapped yet: a failed allocation leaves the results where they are (plain fetch works) */
    ImageSlot& sl = c->slot[k];
    if (int rc = grow(c, &sl.alt_feats, &sl.alt_feats_cap, sl.feats_cap)) return rc;
    if (sl.alt_desc_cap < sl.desc_cap) {
        if (sl.alt_desc) HIP_TRY(c, hipFree(sl.alt_desc));
        sl.alt_desc = nullptr;
        sl.alt_desc_cap = 0;
        HIP_TRY(c, ctx_malloc(c, (void**)&sl.alt_desc, (size_t)sl.desc_cap * 128 * sizeof(float)));
        sl.alt_desc_cap = sl.desc_cap;
    }
    /* finish() has synchronised the compute stream: the slab is complete, and copy_stream needs no event to wait on */
    if (nf) HIP_TRY(c, hipMemcpyAsync(feats, sl.d_feats, nf * sizeof(popsift_hip_feature), hipMemcpyDeviceToHost, c->copy_stream));
    if (nd) HIP_TRY(c, hipMemcpyAsync(desc, sl.d_desc, nd * 128 * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
    std::swap(sl.d_feats, sl.alt_feats);
    std::swap(sl.feats_cap, sl.alt_feats_cap);
    std::swap(sl.d_desc, sl.alt_desc); /* both hold desc_cap descriptors now; d_map / d_rot stay with the slot */
    refresh_caps(c);
    c->copy_pending = true;
    c->copy_seq = c->submit_seq;
    sl.moved = true;
    return POPSIFT_HIP_OK;
}

int popsift_hip_fetch_begin(popsift_hip_ctx* c, popsift_hip_feature* feats, size_t feats_cap, float* desc, size_t desc_cap)
{
    return popsift_hip_fetch_begin_item(c, 0, feats, feats_cap, desc, desc_cap);
}

int popsift_hip_fetch_end(popsift_hip_ctx* c)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    POPSIFT_RANGE("popsift_hip: fetch_end");
    if (!c->copy_pending) return fail(c, POPSIFT_HIP_ERR_STATE, "no download was started with popsift_hip_fetch_begin");
    HIP_TRY(c, hipSetDevice(c->device));
    return drain_copy(c);
}

int popsift_hip_results_dev(popsift_hip_ctx* c, const void** d_feats, const void** d_desc)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = results_here(c)) return rc;
    if (d_feats) *d_feats = c->slot[0].d_feats;
    if (d_desc) *d_desc = c->slot[0].d_desc;
    return POPSIFT_HIP_OK;
}

/* ------------------------------------------------------------------ MatchingMode (N3) */

int popsift_hip_clone_results(popsift_hip_ctx* c, popsift_hip_devfeatures** out)
{
    if (!c || !out) return POPSIFT_HIP_ERR_INVALID;
    *out = nullptr;
    if (int rc = results_here(c)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    popsift_hip_devfeatures* f = new (std::nothrow) popsift_hip_devfeatures();
    if (!f) return fail(c, POPSIFT_HIP_ERR_OOM, "out of host memory");
    f->device = c->device;
    f->n_feat = c->rep.ext_total;
    f->n_desc = c->rep.ori_total;
    int rc = [&]() -> int {
        /* allocations of at least one element keep the pointers valid for empty results */
        HIP_TRY(c, hipMalloc((void**)&f->d_feat, sizeof(DevFeature) * (size_t)std::max(f->n_feat, 1)));
        HIP_TRY(c, hipMalloc((void**)&f->d_desc, sizeof(float) * 128 * (size_t)std::max(f->n_desc, 1)));
        HIP_TRY(c, hipMalloc((void**)&f->d_rev, sizeof(int) * (size_t)std::max(f->n_desc, 1)));
        /* sift_pyramid.cu:323-345: prep_features into the clone, then the two device-to-device copies */
        HIP_TRY(c, launch_clone_features(c->slot[0].d_feats, f->n_feat, f->d_desc, f->d_feat, c->stream));
        if (f->n_desc > 0) {
            HIP_TRY(c, hipMemcpyAsync(f->d_desc, c->slot[0].d_desc, sizeof(float) * 128 * (size_t)f->n_desc,
                                      hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(f->d_rev, c->slot[0].d_map, sizeof(int) * (size_t)f->n_desc, hipMemcpyDeviceToDevice,
                                      c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return 0;
    }();
    if (rc) {
        popsift_hip_devfeatures_free(f);
        return rc;
    }
    *out = f;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_free(popsift_hip_devfeatures* f)
{
    if (!f) return POPSIFT_HIP_OK;
    (void)hipSetDevice(f->device);
    if (f->d_feat) (void)hipFree(f->d_feat);
    if (f->d_desc) (void)hipFree(f->d_desc);
    if (f->d_rev) (void)hipFree(f->d_rev);
    if (f->m_stream) (void)hipStreamDestroy((hipStream_t)f->m_stream);
    if (f->m_partial) (void)hipFree(f->m_partial);
    if (f->m_out) (void)hipFree(f->m_out);
    if (f->m_host) (void)hipHostFree(f->m_host);
    if (f->m_redo) (void)hipFree(f->m_redo);
    if (f->d_norm) (void)hipFree(f->d_norm);
    if (f->m_rnorm) (void)hipFree(f->m_rnorm);
    delete f;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_info(const popsift_hip_devfeatures* f, int* device, int* n_features, int* n_descriptors)
{
    if (!f) return POPSIFT_HIP_ERR_INVALID;
    if (device) *device = f->device;
    if (n_features) *n_features = f->n_feat;
    if (n_descriptors) *n_descriptors = f->n_desc;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_ptrs(const popsift_hip_devfeatures* f, void** d_features, void** d_descriptors,
                                 void** d_reverse_map)
{
    if (!f) return POPSIFT_HIP_ERR_INVALID;
    if (d_features) *d_features = f->d_feat;
    if (d_descriptors) *d_descriptors = f->d_desc;
    if (d_reverse_map) *d_reverse_map = f->d_rev;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_alloc(int device, int n_feat, int n_desc, popsift_hip_devfeatures** out)
{
    if (!out || n_feat < 0 || n_desc < 0) return POPSIFT_HIP_ERR_INVALID;
    *out = nullptr;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return POPSIFT_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= nd) return POPSIFT_HIP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    popsift_hip_devfeatures* f = new (std::nothrow) popsift_hip_devfeatures();
    if (!f) return POPSIFT_HIP_ERR_OOM;
    f->device = device;
    f->n_feat = n_feat;
    f->n_desc = n_desc;
    const size_t bf = sizeof(DevFeature) * (size_t)std::max(n_feat, 1);
    const size_t bd = sizeof(float) * 128 * (size_t)std::max(n_desc, 1);
    const size_t br = sizeof(int) * (size_t)std::max(n_desc, 1);
    const bool   ok = hipMalloc((void**)&f->d_feat, bf) == hipSuccess && hipMalloc((void**)&f->d_desc, bd) == hipSuccess &&
                    hipMalloc((void**)&f->d_rev, br) == hipSuccess && hipMemset(f->d_feat, 0, bf) == hipSuccess &&
                    hipMemset(f->d_desc, 0, bd) == hipSuccess && hipMemset(f->d_rev, 0, br) == hipSuccess;
    if (!ok) {
        popsift_hip_devfeatures_free(f);
        return POPSIFT_HIP_ERR_OOM;
    }
    *out = f;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_from_host(int device, const float* desc, int n, popsift_hip_devfeatures** out)
{
    if (!out || n < 0 || (n > 0 && !desc)) return POPSIFT_HIP_ERR_INVALID;
    *out = nullptr;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return POPSIFT_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= nd) return POPSIFT_HIP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    popsift_hip_devfeatures* f = new (std::nothrow) popsift_hip_devfeatures();
    if (!f) return POPSIFT_HIP_ERR_OOM;
    f->device = device;
    f->n_feat = 0;
    f->n_desc = n;
    bool ok = hipMalloc((void**)&f->d_feat, sizeof(DevFeature)) == hipSuccess &&
              hipMalloc((void**)&f->d_desc, sizeof(float) * 128 * (size_t)std::max(n, 1)) == hipSuccess &&
              hipMalloc((void**)&f->d_rev, sizeof(int) * (size_t)std::max(n, 1)) == hipSuccess;
    if (ok && n > 0) {
        ok = hipMemcpy(f->d_desc, desc, sizeof(float) * 128 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(f->d_rev, 0xff, sizeof(int) * (size_t)n) == hipSuccess; /* -1: no feature behind it */
    }
    if (!ok) {
        popsift_hip_devfeatures_free(f);
        return POPSIFT_HIP_ERR_OOM;
    }
    *out = f;
    return POPSIFT_HIP_OK;
}

int popsift_hip_devfeatures_download(const popsift_hip_devfeatures* f, float* desc, int32_t* rev)
{
    if (!f) return POPSIFT_HIP_ERR_INVALID;
    if (hipSetDevice(f->device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    if (f->n_desc > 0) {
        if (desc && hipMemcpy(desc, f->d_desc, sizeof(float) * 128 * (size_t)f->n_desc, hipMemcpyDeviceToHost) != hipSuccess)
            return POPSIFT_HIP_ERR_DEVICE;
        if (rev && hipMemcpy(rev, f->d_rev, sizeof(int) * (size_t)f->n_desc, hipMemcpyDeviceToHost) != hipSuccess)
            return POPSIFT_HIP_ERR_DEVICE;
    }
    return POPSIFT_HIP_OK;
}

static std::atomic<int> g_match_path{POPSIFT_HIP_MATCH_AUTO};

int popsift_hip_match_set_path(int path)
{
    if (path < POPSIFT_HIP_MATCH_AUTO || path > POPSIFT_HIP_MATCH_SCREEN) return POPSIFT_HIP_ERR_INVALID;
    g_match_path.store(path);
    return POPSIFT_HIP_OK;
}

int popsift_hip_match_sets(const popsift_hip_devfeatures* lc, const popsift_hip_devfeatures* r, popsift_hip_match* out)
{
    if (!lc || !r || (lc->n_desc > 0 && !out)) return POPSIFT_HIP_ERR_INVALID;
    if (lc->n_desc == 0) return POPSIFT_HIP_OK;
    /* the scratch buffers live in the left set (one match at a time per left set, like one image at a time per context) */
    popsift_hip_devfeatures* l = const_cast<popsift_hip_devfeatures*>(lc);
    if (hipSetDevice(l->device) != hipSuccess) return POPSIFT_HIP_ERR_DEVICE;
    const float* rdesc = r->d_desc;
    float*       r_copy = nullptr;
    int          rc = POPSIFT_HIP_OK;
    auto         ok = [&](hipError_t e) {
        if (e != hipSuccess && rc == POPSIFT_HIP_OK) rc = (e == hipErrorOutOfMemory) ? POPSIFT_HIP_ERR_OOM : POPSIFT_HIP_ERR_DEVICE;
        return e == hipSuccess;
    };
    const size_t out_bytes = sizeof(popsift_hip_match) * (size_t)l->n_desc;
    if (!l->m_stream) {
        hipStream_t s = nullptr;
        if (ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking))) l->m_stream = s;
    }
    if (rc == POPSIFT_HIP_OK && !l->m_out) ok(hipMalloc(&l->m_out, out_bytes));
    if (rc == POPSIFT_HIP_OK && !l->m_host) ok(hipHostMalloc(&l->m_host, out_bytes, hipHostMallocDefault));
    /* large problems: matrix-core screening + exact re-rank (match_mfma.hip); small ones and
     * POPSIFT_HIP_MATCH_EXACT=1: the exact brute-force kernel alone (match.hip) */
    const int    path = g_match_path.load();
    const bool   screen = path == POPSIFT_HIP_MATCH_SCREEN ||
                        (path == POPSIFT_HIP_MATCH_AUTO && (double)l->n_desc * (double)r->n_desc >= 4.0e6);
    const int         n_split = match_splits(l->n_desc, r->n_desc);
    const int         s_split = screen ? screen_splits(l->n_desc, r->n_desc) : 1;
    /* rows the screening pass cannot decide are few: the first REDO_CAP of them are matched with the right set split
     * over many workgroups, a second launch (normally empty) covers the rest of the list */
    const int    REDO_CAP = 2048;
    const int    redo_split = std::min(256, std::max((r->n_desc + 63) / 64, 1));
    size_t       need = match_partial_bytes(l->n_desc, n_split);
    if (screen) need = std::max(need, screen_partial_bytes(l->n_desc, s_split));
    if (screen) need = std::max(need, match_partial_bytes(REDO_CAP, redo_split));
    if (rc == POPSIFT_HIP_OK && need > l->m_partial_cap) {
        if (l->m_partial) (void)hipFree(l->m_partial);
        l->m_partial = nullptr;
        l->m_partial_cap = 0;
        if (ok(hipMalloc(&l->m_partial, need))) l->m_partial_cap = need;
    }
    if (rc == POPSIFT_HIP_OK && r->device != l->device && r->n_desc > 0) {
        /* images of one PopSift object may have been extracted on different GPUs: bring the right set over (xGMI) */
        const size_t bytes = sizeof(float) * 128 * (size_t)r->n_desc;
        if (ok(hipMalloc((void**)&r_copy, bytes)) && ok(hipMemcpyPeer(r_copy, l->device, r->d_desc, r->device, bytes)))
            rdesc = r_copy;
    }
    hipStream_t s = (hipStream_t)l->m_stream;
    float*      rnorm = nullptr;
    if (rc == POPSIFT_HIP_OK && screen) {
        if (!l->m_redo) ok(hipMalloc((void**)&l->m_redo, sizeof(int) * ((size_t)l->n_desc + 1)));
        if (rc == POPSIFT_HIP_OK && !l->d_norm && ok(hipMalloc((void**)&l->d_norm, sizeof(float) * (size_t)l->n_desc)))
            ok(launch_norms(l->d_desc, l->n_desc, l->d_norm, s));
        /* the right set may be the left set of another thread's match: its norms go to a buffer of this call */
        if (rc == POPSIFT_HIP_OK && (size_t)r->n_desc > l->m_rnorm_cap) {
            if (l->m_rnorm) (void)hipFree(l->m_rnorm);
            l->m_rnorm = nullptr;
            l->m_rnorm_cap = 0;
            if (ok(hipMalloc((void**)&l->m_rnorm, sizeof(float) * (size_t)r->n_desc))) l->m_rnorm_cap = (size_t)r->n_desc;
        }
        rnorm = l->m_rnorm;
        if (rc == POPSIFT_HIP_OK) ok(launch_norms(rdesc, r->n_desc, rnorm, s));
    }
    if (rc == POPSIFT_HIP_OK && screen) {
        ok(launch_match_screen(l->d_desc, l->n_desc, rdesc, r->n_desc, l->d_norm, rnorm, s_split, l->m_partial,
                               (popsift_hip_match*)l->m_out, l->m_redo + 1, l->m_redo, s)) &&
            ok(launch_match(l->d_desc, l->n_desc, rdesc, r->n_desc, redo_split, l->m_partial, (popsift_hip_match*)l->m_out,
                            l->m_redo + 1, l->m_redo, 0, REDO_CAP, s)) &&
            ok(launch_match(l->d_desc, l->n_desc, rdesc, r->n_desc, n_split, l->m_partial, (popsift_hip_match*)l->m_out,
                            l->m_redo + 1, l->m_redo, REDO_CAP, l->n_desc, s));
    } else if (rc == POPSIFT_HIP_OK) {
        ok(launch_match(l->d_desc, l->n_desc, rdesc, r->n_desc, n_split, l->m_partial, (popsift_hip_match*)l->m_out, nullptr,
                        nullptr, 0, 0, s));
    }
    if (rc == POPSIFT_HIP_OK && ok(hipMemcpyAsync(l->m_host, l->m_out, out_bytes, hipMemcpyDeviceToHost, s)) &&
        ok(hipStreamSynchronize(s)))
        memcpy(out, l->m_host, out_bytes);
    if (r_copy) (void)hipFree(r_copy);
    return rc;
}

void* popsift_hip_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}

void popsift_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int popsift_hip_get_report(const popsift_hip_ctx* c, popsift_hip_report* rep)
{
    if (!c || !rep) return POPSIFT_HIP_ERR_INVALID;
    *rep = c->rep;
    return POPSIFT_HIP_OK;
}

int popsift_hip_set_profile(popsift_hip_ctx* c, int profile)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    c->profile = (profile == 2) ? 2 : (profile ? 1 : 0);
    return POPSIFT_HIP_OK;
}

int popsift_hip_octave_dims(const popsift_hip_ctx* c, int octave, int* w, int* h)
{
    if (!c || !c->have_image || octave < 0 || octave >= c->pd.n_oct) return POPSIFT_HIP_ERR_INVALID;
    if (w) *w = c->pd.o[octave].w;
    if (h) *h = c->pd.o[octave].h;
    return POPSIFT_HIP_OK;
}

int popsift_hip_download_plane(popsift_hip_ctx* c, int octave, int kind, int level, float* out)
{
    float*         p = nullptr;
    const OctDesc* od = nullptr;
    if (!out) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = plane_ptr(c, octave, kind, level, &p, &od)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (kind == 1 && c->pd.dog_fly) /* not stored: DoG(l) = G(l+1) - G(l) into the (otherwise unused) DoG plane */
        HIP_TRY(c, launch_dog_plane(p, c->slot[0].d_arena + od->data_off + (level + 1) * od->plane_stride,
                                    c->slot[0].d_arena + od->data_off + level * od->plane_stride, (size_t)od->plane_stride, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy2D(out, (size_t)od->w * 4, p, (size_t)od->pitch * 4, (size_t)od->w * 4, od->h,
                           hipMemcpyDeviceToHost));
    return POPSIFT_HIP_OK;
}

int popsift_hip_upload_plane(popsift_hip_ctx* c, int octave, int kind, int level, const float* in)
{
    float*         p = nullptr;
    const OctDesc* od = nullptr;
    if (!in) return POPSIFT_HIP_ERR_INVALID;
    if (c && c->have_image && kind == 1 && c->pd.dog_fly)
        return fail(c, POPSIFT_HIP_ERR_STATE, "DoG planes are not stored (params.store_dog = 0): upload Gaussian planes");
    if (int rc = plane_ptr(c, octave, kind, level, &p, &od)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy2D(p, (size_t)od->pitch * 4, in, (size_t)od->w * 4, (size_t)od->w * 4, od->h,
                           hipMemcpyHostToDevice));
    return POPSIFT_HIP_OK;
}

int popsift_hip_download_extrema(popsift_hip_ctx* c, popsift_hip_extremum* out, size_t cap, int* n)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (int rc = finish(c)) return rc;
    const int total = c->rep.ext_total;
    if (n) *n = total;
    if (!out) return POPSIFT_HIP_OK;
    if (cap < (size_t)total) return fail(c, POPSIFT_HIP_ERR_TOO_SMALL, "output buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<InitExt> tmp((size_t)std::max(c->sc.max_extrema, 1));
    size_t               k = 0;
    for (int o = 0; o < c->pd.n_oct; o++) {
        const int cnt = c->rep.ext_ct[o];
        if (cnt <= 0) continue;
        HIP_TRY(c, hipMemcpy(tmp.data(), final_iext(c) + (size_t)o * c->sc.max_extrema, (size_t)cnt * sizeof(InitExt),
                             hipMemcpyDeviceToHost));
        for (int i = 0; i < cnt; i++, k++) {
            out[k].xpos = tmp[i].xpos;
            out[k].ypos = tmp[i].ypos;
            out[k].lpos = tmp[i].lpos;
            out[k].sigma = tmp[i].sigma;
            out[k].octave = o;
            out[k].cell = tmp[i].cell;
        }
    }
    return POPSIFT_HIP_OK;
}

int popsift_hip_debug_set(popsift_hip_ctx* c, int what, int value)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    switch (what) {
    case POPSIFT_HIP_DEBUG_DET_QCAP:
        c->det_qcap = c->sc.det_qcap = std::max(value, 0);
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_CAND_CAP:
        c->cand_cap_init = std::max(value, DET_SUBQ);
        c->cand_cap_user = true;
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_OHIST_CAP:
        c->ohist_cap_init = std::max(value, 1);
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_FAIL_ALLOC:
        c->fail_alloc_in = std::max(value, 0);
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_KP_WAVES:
        c->kp_waves = std::min(std::max(value / 32 * 32, 32), 1 << 20);
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_PYR_ORDER:
        c->pyr_order = value;
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_BLUR_PATH:
        if (value < 0 || value > 2) return fail(c, POPSIFT_HIP_ERR_INVALID, "BLUR_PATH: 0, 1 or 2");
        c->blur_tune.path = value;
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_BLUR_SEG:
        c->blur_tune.seg_rows = std::max(value, 0);
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_PYR_TAIL:
        c->pyr_tail = value;
        return POPSIFT_HIP_OK;
    case POPSIFT_HIP_DEBUG_DESC_ROWS:
        c->desc_rows = c->sc.desc_rows = std::max(value, 4);
        return POPSIFT_HIP_OK;
    }
    return fail(c, POPSIFT_HIP_ERR_INVALID, "unknown debug switch %d", what);
}

int popsift_hip_rerun_keypoint_stages(popsift_hip_ctx* c)
{
    if (!c) return POPSIFT_HIP_ERR_INVALID;
    if (!c->have_image || !c->batch_ok) return fail(c, POPSIFT_HIP_ERR_STATE, "no image submitted");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    /* a download started by fetch_begin may still read the slab the re-run is about to write (the slabs were swapped and
     * `moved` is cleared below without a new submit_seq): wait for it first */
    if (int rc = drain_copy(c)) return rc;
    c->blur_events_used = 0;
    HIP_TRY(c, hipEventRecord(c->ev_begin, c->stream));
    if (int rc = enqueue_keypoint_stages(c)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_end, c->stream));
    c->finished = false;
    for (ImageSlot& sl : c->slot) sl.moved = false;
    return POPSIFT_HIP_OK;
}

} /* extern "C" */
