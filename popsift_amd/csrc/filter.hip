/*
 * filter.hip -- the grid filter: thin the initial extrema of all octaves to roughly
 * Config::getFilterMaxExtrema() by capping the number kept in every cell of a grid_size x grid_size
 * partition of the image.
 *
 * Reference: Pyramid::extrema_filter_grid (s_filtergrid.cu:109-322), called from Pyramid::orientation
 * when `filter_max > 0 && int(filter_max * 1.1) < ext_total` (s_orientation.cu:353-367).  There it is
 * a Thrust pipeline with a host round trip in the middle: sort ALL extrema by (cell, scale), count per
 * cell, copy the counts to the host, derive one per-cell limit, copy back, mark the tail of every
 * cell's run, and rebuild the per-octave index lists with copy_if.
 *
 * Here nothing leaves the device and nothing is sorted.  What the sort is used for is a per-cell
 * "keep the `limit` first members in (scale, original index) order" -- a per-cell top-k selection:
 *   k_filter_count   per-cell member counts (wave-aggregated atomics)
 *   k_filter_limit   one workgroup: the reference's host arithmetic (ascending counts, `sumup`,
 *                    tail average, integer-division quirk included) -> limit per cell
 *   k_filter_hist /  7 x 8-bit MSD radix select per cell on a UNIQUE 56-bit key
 *   k_filter_pick    (31 bits of scale order | 25 bits of reversed original index): after the last
 *                    digit the prefix is the key of the limit-th member, keep <=> key >= prefix
 *   k_filter_compact survivors -> second InitExt buffer, one returning atomic per 2048 candidates
 *                    (a hot returning atomic saturates near 90/us on MI355X)
 *   k_filter_commit  new per-octave counts replace the old ones
 * All launches are sized by capacities; the real counts stay in device memory (no host sync, the
 * reference blocks on readDescCountersFromDevice here).  When the 10 % test fails the same kernels run
 * with every limit = count, i.e. they copy the list unchanged.
 *
 * Order: "original index" is the position in this build's per-octave lists (octave-major), which like
 * the reference's atomicAdd arrival order carries no meaning; RandomScale therefore agrees with the
 * reference (and the oracle) in the NUMBER kept per cell, the two scale orders agree in the members.
 */
#include "kernels.h"

namespace popsift_hip {
namespace {

constexpr int FILTER_IDX_BITS = 25;
constexpr int FILTER_KEY_BITS = 31 + FILTER_IDX_BITS; /* 56 = 7 digits of 8 bits */
constexpr int FILTER_PASSES = 7;
constexpr int COMPACT_ITEMS = 8; /* candidates per lane in k_filter_compact */
constexpr int COMPACT_CHUNK = 256 * COMPACT_ITEMS;

__device__ __forceinline__ int oct_count(const Counters* ct, const SiftConsts& sc, int o)
{
    return min(ct->ext_ct[o], sc.max_extrema);
}

/* s_filtergrid.cu:56-70 FunctionExtractCell + the two comparators (:34-53), as one integer key whose
 * DEScending order is the reference's sorted order */
__device__ __forceinline__ unsigned long long filter_key(const InitExt& e, int octave, int gidx, int mode)
{
    const float        scale = e.sigma * powf(2.0f, (float)octave);
    const unsigned int bits = __float_as_uint(scale) & 0x7fffffffu;
    unsigned int       primary = 0u;
    if (mode == POPSIFT_HIP_FILTER_LARGEST_FIRST) primary = bits;
    if (mode == POPSIFT_HIP_FILTER_SMALLEST_FIRST) primary = 0x7fffffffu - bits;
    const unsigned int secondary = ((1u << FILTER_IDX_BITS) - 1u) - (unsigned int)gidx;
    return ((unsigned long long)primary << FILTER_IDX_BITS) | secondary;
}

/* one atomicAdd per distinct `slot` value in the wave instead of one per lane */
__device__ __forceinline__ void wave_agg_inc(int* base, int slot, bool active)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int                leader = __ffsll((long long)todo) - 1;
        const int                s = __shfl(slot, leader);
        const unsigned long long same = __ballot(active && slot == s) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(base + s, __popcll(same));
        todo &= ~same;
    }
}

/* flattened index g over the per-octave lists -> (octave, position) */
struct Locator {
    int ps[PS_MAX_OCT + 1];
    int n_oct;
};

__device__ __forceinline__ void locator_init(int* s_ps, const Counters* ct, const SiftConsts& sc, int n_oct)
{
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int o = 0; o < n_oct; o++) {
            s_ps[o] = acc;
            acc += oct_count(ct, sc, o);
        }
        s_ps[n_oct] = acc;
    }
    __syncthreads();
}

/* per-cell member counts: LDS histogram per workgroup (NC <= FILTER_MAX_CELLS ints), then one global atomic per
 * non-empty cell and workgroup -- with a handful of cells a per-wave atomic would hammer a few hot addresses */
__global__ __launch_bounds__(256) void k_filter_count(int n_oct, SiftConsts sc, BatchDesc bd)
{
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const InitExt* __restrict__  iext = bd.s[blockIdx.y].iext;
    FilterState* __restrict__    fs = bd.s[blockIdx.y].fstate;
    __shared__ int s_ps[PS_MAX_OCT + 1];
    __shared__ int s_hist[FILTER_MAX_CELLS];
    const int      ncell = sc.grid_size * sc.grid_size;
    for (int c = threadIdx.x; c < ncell; c += 256) s_hist[c] = 0;
    locator_init(s_ps, ct, sc, n_oct); /* ends with a barrier */
    const int total = s_ps[n_oct];
    const int span = gridDim.x * 256;
    for (int g0 = blockIdx.x * 256; g0 < total; g0 += span) {
        const int  g = g0 + threadIdx.x;
        const bool act = g < total;
        int        cell = 0;
        if (act) {
            int o = 0;
            while (o + 1 < n_oct && g >= s_ps[o + 1]) o++;
            cell = iext[(size_t)o * sc.max_extrema + (g - s_ps[o])].cell;
            cell = min(max(cell, 0), ncell - 1);
        }
        wave_agg_inc(s_hist, cell, act);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncell; c += 256)
        if (s_hist[c] > 0) atomicAdd(&fs->cell_count[c], s_hist[c]);
}

/* The host part of extrema_filter_grid (s_filtergrid.cu:204-262) on one workgroup.
 * NC = grid_size^2 <= FILTER_MAX_CELLS counts, sorted ascending by a bitonic network in LDS. */
__global__ __launch_bounds__(256) void k_filter_limit(int n_oct, SiftConsts sc, BatchDesc bd)
{
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    FilterState* __restrict__    fs = bd.s[blockIdx.y].fstate;
    __shared__ int s_cnt[FILTER_MAX_CELLS];
    __shared__ int s_scan[2][FILTER_MAX_CELLS];
    __shared__ int s_ct, s_total, s_limit, s_active;
    const int      tid = threadIdx.x;
    const int      n = sc.grid_size * sc.grid_size;
    int            np2 = 1;
    while (np2 < n) np2 <<= 1;

    if (tid == 0) {
        int acc = 0;
        for (int o = 0; o < n_oct; o++) acc += oct_count(ct, sc, o);
        s_total = acc;
        s_ct = 0;
        /* s_orientation.cu:362: int(filter_max * 1.1) with the double constant */
        s_active = (sc.filter_max > 0 && (int)((double)sc.filter_max * 1.1) < acc) ? 1 : 0;
    }
    /* empty cells count 0, as in the reference's zero-initialised count vector; the padding up to a
     * power of two sorts to the end and is ignored */
    for (int i = tid; i < np2; i += 256) s_cnt[i] = (i < n) ? fs->cell_count[i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const int  a = s_cnt[i], b = s_cnt[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s_cnt[i] = b;
                        s_cnt[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    /* inclusive prefix sums of the ascending counts (Hillis-Steele, double buffered) */
    for (int i = tid; i < n; i += 256) s_scan[0][i] = s_cnt[i];
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < n; d <<= 1) {
        for (int i = tid; i < n; i += 256) s_scan[cur ^ 1][i] = s_scan[cur][i] + (i >= d ? s_scan[cur][i - d] : 0);
        cur ^= 1;
        __syncthreads();
    }
    /* sumup[i] = count[i] * (n-1-i) + prefix[i]: the total if every later cell were cut to count[i] */
    int mine = 0;
    for (int i = tid; i < n; i += 256) {
        const long long sumup = (long long)s_cnt[i] * (n - 1 - i) + s_scan[cur][i];
        if (sumup > sc.filter_max) mine++;
    }
    if (mine) atomicAdd(&s_ct, mine);
    __syncthreads();
    if (tid == 0) {
        const int cnt = s_ct;
        int       limit = 0x7fffffff;
        if (s_active && cnt > 0) {
            const int   tail = s_scan[cur][n - 1] - (cnt < n ? s_scan[cur][n - 1 - cnt] : 0);
            const float tailaverage = (float)tail / (float)cnt;
            /* (ext_total - max) / ct is an int / int in the reference (s_filtergrid.cu:252) */
            limit = (int)ceilf(tailaverage - (float)((s_total - sc.filter_max) / cnt));
        }
        s_limit = limit;
        fs->active = s_active;
        fs->newlimit = limit;
        for (int o = 0; o < PS_MAX_OCT; o++) fs->new_ct[o] = 0;
    }
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
        const int count = fs->cell_count[c];
        const int keep = max(min(count, s_limit), 0);
        fs->cell_limit[c] = keep;
        fs->remaining[c] = keep;
        fs->prefix[c] = 0ull;
        /* 0: select by radix passes, 1: keep every member, 2: keep none */
        fs->cell_mode[c] = keep >= count ? 1 : (keep == 0 ? 2 : 0);
    }
}

__global__ __launch_bounds__(256) void k_filter_hist(int pass, int n_oct, SiftConsts sc, BatchDesc bd)
{
    const Counters* __restrict__    ct = bd.s[blockIdx.y].ct;
    const InitExt* __restrict__     iext = bd.s[blockIdx.y].iext;
    const FilterState* __restrict__ fs = bd.s[blockIdx.y].fstate;
    int* __restrict__               hist = bd.s[blockIdx.y].fhist;
    __shared__ int s_ps[PS_MAX_OCT + 1];
    locator_init(s_ps, ct, sc, n_oct);
    const int total = s_ps[n_oct];
    const int ncell = sc.grid_size * sc.grid_size;
    const int span = gridDim.x * 256;
    const int shift = FILTER_KEY_BITS - 8 * (pass + 1); /* position of this pass's digit */
    for (int g0 = blockIdx.x * 256; g0 < total; g0 += span) {
        const int g = g0 + threadIdx.x;
        bool      act = g < total;
        int       slot = 0;
        if (act) {
            int o = 0;
            while (o + 1 < n_oct && g >= s_ps[o + 1]) o++;
            const InitExt e = iext[(size_t)o * sc.max_extrema + (g - s_ps[o])];
            const int     cell = min(max(e.cell, 0), ncell - 1);
            if (fs->cell_mode[cell] != 0) {
                act = false;
            } else {
                const unsigned long long key = filter_key(e, o, g, sc.filter_mode);
                /* still a candidate for the threshold: all higher digits equal the chosen prefix */
                act = (pass == 0) || ((key >> (shift + 8)) == fs->prefix[cell]);
                slot = cell * 256 + (int)((key >> shift) & 255ull);
            }
        }
        wave_agg_inc(hist, slot, act);
    }
}

/* one workgroup per cell: walk the 256 digit counts from the top until `remaining` members are covered */
__global__ __launch_bounds__(256) void k_filter_pick(BatchDesc bd)
{
    FilterState* __restrict__ fs = bd.s[blockIdx.y].fstate;
    int* __restrict__         hist = bd.s[blockIdx.y].fhist;
    __shared__ int s_above[256];
    const int      cell = blockIdx.x, d = threadIdx.x;
    const int      mine = hist[cell * 256 + d];
    hist[cell * 256 + d] = 0; /* ready for the next pass */
    if (fs->cell_mode[cell] != 0) return;
    s_above[d] = mine;
    __syncthreads();
    /* suffix sums: above[d] = number of candidates with a larger digit */
    for (int s = 1; s < 256; s <<= 1) {
        const int v = (d + s < 256) ? s_above[d + s] : 0;
        __syncthreads();
        s_above[d] += v;
        __syncthreads();
    }
    const int incl = s_above[d], above = incl - mine;
    const int rem = fs->remaining[cell];
    __syncthreads();
    if (above < rem && rem <= incl) { /* exactly one digit satisfies this */
        fs->prefix[cell] = (fs->prefix[cell] << 8) | (unsigned long long)d;
        fs->remaining[cell] = rem - above;
    }
}

__global__ __launch_bounds__(256) void k_filter_compact(int chunks_per_oct, SiftConsts sc, BatchDesc bd, int n_oct)
{
    const Counters* __restrict__ ct = bd.s[blockIdx.y].ct;
    const InitExt* __restrict__  iext = bd.s[blockIdx.y].iext;
    InitExt* __restrict__        out = bd.s[blockIdx.y].iext2;
    FilterState* __restrict__    fs = bd.s[blockIdx.y].fstate;
    __shared__ int s_ps[PS_MAX_OCT + 1];
    __shared__ int s_wsum[4];
    __shared__ int s_base;
    locator_init(s_ps, ct, sc, n_oct);
    const int o = blockIdx.x / chunks_per_oct, chunk = blockIdx.x % chunks_per_oct;
    const int cnt = s_ps[o + 1] - s_ps[o];
    const int i0 = chunk * COMPACT_CHUNK;
    if (i0 >= cnt) return; /* whole workgroup */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncell = sc.grid_size * sc.grid_size;

    InitExt e[COMPACT_ITEMS];
    bool    keep[COMPACT_ITEMS];
    int     self = 0;
#pragma unroll
    for (int k = 0; k < COMPACT_ITEMS; k++) {
        const int i = i0 + tid * COMPACT_ITEMS + k;
        keep[k] = false;
        if (i < cnt) {
            e[k] = iext[(size_t)o * sc.max_extrema + i];
            const int cell = min(max(e[k].cell, 0), ncell - 1);
            const int m = fs->cell_mode[cell];
            keep[k] = (m == 1) || (m == 0 && filter_key(e[k], o, s_ps[o] + i, sc.filter_mode) >= fs->prefix[cell]);
        }
        self += keep[k] ? 1 : 0;
    }
    int incl = self;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int v = __shfl_up(incl, s);
        if (lane >= s) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += s_wsum[w];
    if (tid == 255) s_base = atomicAdd(&fs->new_ct[o], woff + incl);
    __syncthreads();
    int pos = s_base + woff + incl - self;
#pragma unroll
    for (int k = 0; k < COMPACT_ITEMS; k++)
        if (keep[k]) out[(size_t)o * sc.max_extrema + pos++] = e[k];
}

__global__ void k_filter_commit(int n_oct, BatchDesc bd)
{
    Counters* __restrict__          ct = bd.s[blockIdx.y].ct;
    const FilterState* __restrict__ fs = bd.s[blockIdx.y].fstate;
    const int o = threadIdx.x;
    if (o < PS_MAX_OCT) ct->ext_ct[o] = (o < n_oct) ? fs->new_ct[o] : 0;
}

}  // namespace

size_t filter_hist_bytes(int grid_size) { return (size_t)grid_size * grid_size * 256 * sizeof(int); }

bool filter_supported(int n_oct, int max_extrema, int grid_size)
{
    return grid_size >= 1 && grid_size * grid_size <= FILTER_MAX_CELLS &&
           (long long)n_oct * max_extrema < (1ll << FILTER_IDX_BITS);
}

hipError_t launch_filter(int n_oct, const SiftConsts& sc, const BatchDesc& bd, int nb, hipStream_t s)
{
    const int ncell = sc.grid_size * sc.grid_size;
    for (int k = 0; k < nb; k++) {
        hipError_t err = hipMemsetAsync(bd.s[k].fstate, 0, sizeof(FilterState), s);
        if (err != hipSuccess) return err;
        err = hipMemsetAsync(bd.s[k].fhist, 0, filter_hist_bytes(sc.grid_size), s);
        if (err != hipSuccess) return err;
    }
    const int sweep = 512; /* grid-stride workgroups of the per-candidate sweeps */
    hipLaunchKernelGGL(k_filter_count, dim3(64, nb), dim3(256), 0, s, n_oct, sc, bd);
    hipLaunchKernelGGL(k_filter_limit, dim3(1, nb), dim3(256), 0, s, n_oct, sc, bd);
    for (int pass = 0; pass < FILTER_PASSES; pass++) {
        hipLaunchKernelGGL(k_filter_hist, dim3(sweep, nb), dim3(256), 0, s, pass, n_oct, sc, bd);
        hipLaunchKernelGGL(k_filter_pick, dim3(ncell, nb), dim3(256), 0, s, bd);
    }
    const int chunks = (sc.max_extrema + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
    hipLaunchKernelGGL(k_filter_compact, dim3(n_oct * chunks, nb), dim3(256), 0, s, chunks, sc, bd, n_oct);
    hipLaunchKernelGGL(k_filter_commit, dim3(1, nb), dim3(64), 0, s, n_oct, bd);
    return hipGetLastError();
}

}  // namespace popsift_hip
