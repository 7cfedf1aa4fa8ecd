/*
 * blur_march.hip -- one Gaussian level of a LARGE plane as a march down strips (gfx950, wave64).
 *
 * Same arithmetic as pyramid.hip's k_blur_tile<HALO, 0, ..> -- gauss::absoluteSource::horiz + ::vert
 * (s_pyramid_build_aa.cu:17-91), outermost tap first, explicit fmaf, -ffp-contract=off: planes bit-identical to the
 * oracle -- in another shape.  The tile kernel stages a 128 x 64 tile + halo, filters it and is done: load, H pass, V pass
 * and store of a workgroup follow each other, two or three workgroups per CU run in lock-step rounds, and the 2*HALO halo
 * rows are loaded and filtered horizontally twice (1.25x ... 1.41x the rows).  Its octave-0 launches took the SUM of
 * their memory, LDS and vector time (round 3: 17 ... 26 us for 66 MB).
 *
 * Here a workgroup of 256 lanes owns a strip of 128 columns and a segment of rows and marches down it 32 rows a step:
 *   - the horizontally filtered rows live in an LDS ring of 64 rows (two chunks of 32); a step filters the 32 NEW rows
 *     and then produces 32 output rows from ring rows [32k, 32k + 32 + 2*HALO): every source row is loaded and filtered
 *     horizontally once per segment (plus 2*HALO rows at its start);
 *   - the raw rows of step k+2 are requested (16-byte loads into registers) BEFORE the arithmetic of step k, so a
 *     workgroup's memory latency lies under its own H and V passes; the barriers wait for LDS only (s_waitcnt lgkmcnt +
 *     s_barrier -- __syncthreads() would drain the loads in flight);
 *   - a raw row is written to LDS and filtered by the same half-wave (LDS operations of a wave execute in order), so
 *     there is no barrier between staging and the H pass: two barriers per step;
 *   - 28 ... 41 KB of LDS per workgroup: three or four workgroups per CU at different phases.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sift_types.h"
#include "kernels.h"
#include "blur_common.h"

namespace popsift_hip {

namespace {

constexpr int TW = BLUR_TW;
constexpr int MCH = 32;  /* rows per step           */
constexpr int MNR = 64;  /* ring rows: two chunks   */
constexpr int MNT = 256; /* lanes per workgroup     */

/* wait for this wave's LDS operations, then the workgroup barrier; vector-memory loads stay in flight */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool EDGE>
__device__ __forceinline__ v4f load_chunk(const float* __restrict__ row, int gx0, int w)
{
    if (!EDGE || (gx0 >= 0 && gx0 + 3 < w)) return *reinterpret_cast<const v4f*>(row + gx0);
    v4f v;
    v.x = row[clampi(gx0 + 0, 0, w - 1)];
    v.y = row[clampi(gx0 + 1, 0, w - 1)];
    v.z = row[clampi(gx0 + 2, 0, w - 1)];
    v.w = row[clampi(gx0 + 3, 0, w - 1)];
    return v;
}

/* Pins the order of the instruction stream to the order of the source: the four accumulators pass through an empty asm
 * statement (every multiply-add written before it is issued before it, every one after it after it -- the IR passes
 * would otherwise regroup the chains row by row and read the whole window first), the memory clobber keeps the LDS reads
 * on their side, the scheduling barrier holds the machine scheduler.  No instruction is emitted; the waits on the LDS
 * counter are still the compiler's. */
#define MARCH_PIN4(acc)                                                                                      \
    do {                                                                                                     \
        asm volatile("" : "+v"((acc)[0]), "+v"((acc)[1]), "+v"((acc)[2]), "+v"((acc)[3]) : : "memory");     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define MARCH_PIN()                        \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

/* experiment knobs (tools/build_variants.py): waves per SIMD the register allocator must leave room for (0 = its own
 * choice), LDS-read lead of the vertical pass in taps */
#ifndef MARCH_WPE
#define MARCH_WPE 0
#endif
#ifndef MARCH_VPD
#define MARCH_VPD 2
#endif
#if MARCH_WPE > 0
#define MARCH_BOUNDS __launch_bounds__(MNT, MARCH_WPE)
#else
#define MARCH_BOUNDS __launch_bounds__(MNT)
#endif

template <bool B>
using flag = std::integral_constant<bool, B>;

/* EDGE: some 16-byte chunk of the strip crosses the plane's left / right border (element loads with clamped columns).
 * The two instances are separate bodies, not a flag inside one: what the compiler hoists out of the step loop for the
 * clamped path (eight clamped column offsets per lane) would otherwise sit in registers in every workgroup. */
template <int HALO, bool EDGE>
__device__ __forceinline__ void march_body(const BlurArgs& a, const BatchDesc& bd, float* __restrict__ s_t, int strip, int seg)
{
    constexpr int HP = (HALO + 3) & ~3; /* left / right halo, padded to 16 B          */
    constexpr int SW = TW + 2 * HP;     /* LDS row pitch (floats)                     */
    constexpr int NW = 1 + HP / 2;      /* H-pass window in 16-byte chunks            */
    constexpr int HCH = HP / 2;         /* halo chunks of a row: HP / 4 left + HP / 4 right */
    constexpr int VW = 4 + 2 * HALO;    /* V-pass window rows of a 4 x 4 block        */
    static_assert(2 * HALO <= MCH, "a step's window spans two chunks");
    static_assert(4 * HCH <= 32, "the halo chunks of a half-wave's four rows fit one load");

    float* const       arena = bd.s[blockIdx.y].arena;
    const float* const src = arena + a.src_off;
    float* const       dst = arena + a.dst_off;
    float* const       next0 = a.next0_off >= 0 ? arena + a.next0_off : nullptr;

    const int w = a.w, h = a.h, pitch = a.pitch;
    const int tx0 = strip * TW;
    const int Y0 = seg * a.seg_rows, Y1 = min(Y0 + a.seg_rows, h);
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const bool wide = tx0 + TW <= w; /* the strip's outputs all lie inside the plane */

    const int R = (Y1 - Y0) + 2 * HALO; /* ring rows this segment needs: ring row r = plane row Y0 - HALO + r */
    const int K = (Y1 - Y0 + MCH - 1) / MCH;

    /* role of this lane in the halo load of its half-wave's four rows */
    const int  hi = l / HCH, hc = l - hi * HCH;
    const bool hact = l < 4 * HCH;
    const int  hgx = hc < HP / 4 ? tx0 - HP + 4 * hc : tx0 + TW + 4 * (hc - HP / 4);
    const int  hlx = hc < HP / 4 ? 4 * hc : HP + TW + 4 * (hc - HP / 4);

    v4f pre[5]; /* the raw rows of the next chunk: four centre chunks (rows hw, hw + 8, hw + 16, hw + 24) + one halo chunk */

    /* FULL: all 32 rows of the chunk are needed (no per-row predicate: straight-line code) */
    auto issue_loads_t = [&](int m, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int      nr = R - MCH * m; /* rows of chunk m that are needed */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = 8 * i + hw;
            if (FULL || n < nr) {
                const int gy = clampi(Y0 - HALO + MCH * m + n, 0, h - 1);
                pre[i] = load_chunk<EDGE>(src + (size_t)gy * pitch, tx0 + 4 * l, w);
            }
        }
        {
            const int n = 8 * hi + hw;
            if (hact && (FULL || n < nr)) {
                const int gy = clampi(Y0 - HALO + MCH * m + n, 0, h - 1);
                pre[4] = load_chunk<EDGE>(src + (size_t)gy * pitch, hgx, w);
            }
        }
    };
    auto issue_loads = [&](int m) {
        if (R - MCH * m >= MCH)
            issue_loads_t(m, flag<true>{});
        else
            issue_loads_t(m, flag<false>{});
    };
    /* horizontal pass of one ring row, in place: all reads of the row precede its write in the wave's instruction stream */
    auto hrow = [&](int slot) {
        v4f        win[NW];
        const v4f* p = reinterpret_cast<const v4f*>(&s_t[slot * SW + 4 * l]);
#pragma unroll
        for (int j = 0; j < NW; j++) win[j] = p[j];
#define PS_W(i) win[(i) >> 2][(i) & 3]
        v4f out;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int cpos = HP + o;
            float     acc = PS_W(cpos) * a.taps.g[0];
#pragma unroll
            for (int k = HALO; k > 0; k--) acc = fmaf(PS_W(cpos - k) + PS_W(cpos + k), a.taps.g[k], acc);
            out[o] = acc;
        }
#undef PS_W
        *reinterpret_cast<v4f*>(&s_t[slot * SW + HP + 4 * l]) = out;
    };
    /* the raw rows of chunk m (in `pre`) to their ring slots, then the horizontal pass of the same rows by the same
     * half-waves: no barrier in between.  Two rows at a time, so that at most two windows are in registers. */
    auto stage_hpass_t = [&](int m, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int      nr = R - MCH * m;
        const int      sb = (m & 1) * MCH;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = 8 * i + hw;
            if (FULL || n < nr) *reinterpret_cast<v4f*>(&s_t[(sb + n) * SW + HP + 4 * l]) = pre[i];
        }
        {
            const int n = 8 * hi + hw;
            if (hact && (FULL || n < nr)) *reinterpret_cast<v4f*>(&s_t[(sb + n) * SW + hlx]) = pre[4];
        }
    };
    auto hpass_t = [&](int m, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int      nr = R - MCH * m;
        const int      sb = (m & 1) * MCH;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = 8 * i + hw;
            if (FULL || n < nr) hrow(sb + n);
            if (i == 1) MARCH_PIN();
        }
    };
    /* vertical pass of step k: lane (hw, l) owns the 4 x 4 block rows 4 hw .., columns 4 l .. of the step's 32 x 128 outputs.
     * The four output rows advance tap by tap together: at tap kk they read window rows [HALO - kk, HALO - kk + 3] and
     * [HALO + kk, HALO + kk + 3] (each row's own order of operations -- outermost tap first, upper then lower sample, centre
     * last -- is the reference's), so a window row is read from LDS VPD taps before its first use and dies after its last:
     * about twelve rows are live whatever HALO is (the compiler, left alone, reads all 4 + 2 HALO rows first: 182 registers
     * at 27 taps). */
    v4f  acc[4]; /* the 4 x 4 outputs of the last vertical pass: stored one phase later (store_out) */
    auto vpass = [&](int k) {
        constexpr int  VPD = MARCH_VPD;
        const int      s4 = (MCH / 4) * k + hw; /* first ring row of the window / 4 */
        const int      cb = HP + 4 * l;
        v4f            win[VW];
        /* window rows 4q .. 4q+3 lie in ring slots ((s4 + q) & 15) * 4 .. + 3: a window wraps between groups of four only */
        auto LD = [&](int j) { win[j] = *reinterpret_cast<const v4f*>(&s_t[(((s4 + (j >> 2)) & (MNR / 4 - 1)) * 4 + (j & 3)) * SW + cb]); };
        /* the tap at which window row j is used first (taps run from HALO down to 1; the centre rows come in on the way) */
        auto first_use = [](int j) -> int {
            const int lowk = j <= 3 ? HALO : (j <= HALO + 2 ? HALO + 3 - j : 0);
            const int highk = j >= 2 * HALO ? HALO : (j >= HALO + 1 ? j - HALO : 0);
            return lowk > highk ? lowk : highk;
        };
#pragma unroll
        for (int j = 0; j < VW; j++)
            if (first_use(j) > HALO - VPD) LD(j);
#pragma unroll
        for (int oo = 0; oo < 4; oo++) acc[oo] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = HALO; kk > 0; kk--) {
#pragma unroll
            for (int j = 0; j < VW; j++)
                if (first_use(j) == kk - VPD) LD(j);
            MARCH_PIN4(acc);
            const float gk = a.taps.g[kk];
#pragma unroll
            for (int oo = 0; oo < 4; oo++) {
                const int cpos = HALO + oo;
#pragma unroll
                for (int c = 0; c < 4; c++) acc[oo][c] = fmaf(win[cpos - kk][c], gk, acc[oo][c]);
#pragma unroll
                for (int c = 0; c < 4; c++) acc[oo][c] = fmaf(win[cpos + kk][c], gk, acc[oo][c]);
            }
        }
        MARCH_PIN4(acc);
#pragma unroll
        for (int oo = 0; oo < 4; oo++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[oo][c] = fmaf(win[HALO + oo][c], a.taps.g[0], acc[oo][c]);
    };
    /* the outputs of step k leave one phase after they were formed, behind the next phase's staging: the wait for the raw
     * rows there (the compiler waits for ALL vector-memory operations of the wave: its counter is in order) then finds
     * these stores a whole phase old instead of just issued */
    auto store_out_t = [&](int k, auto full) {
        constexpr bool FULL = decltype(full)::value; /* all 32 x 128 outputs of the step lie inside the segment and the plane */
        const int      gx = tx0 + 4 * l;
#pragma unroll
        for (int oo = 0; oo < 4; oo++) {
            const int gy = Y0 + MCH * k + 4 * hw + oo;
            if (FULL || (gx < w && gy < Y1)) {
                /* rows are padded to 64 floats, so a 16 B store at gx < w stays inside the row */
                *reinterpret_cast<v4f*>(&dst[(size_t)gy * pitch + gx]) = acc[oo];
                /* level 0 of the next octave = pixel (2x, 2y) of this plane (get_by_2_pick_every_second,
                 * s_pyramid_build.cu:50-71): its width is ceil(w / 2), so 2x <= w - 1 and the reference's min() never clamps */
                if (next0 && (gy & 1) == 0) {
                    float* q = next0 + (size_t)(gy >> 1) * a.next_pitch + (gx >> 1);
                    q[0] = acc[oo].x;
                    if (gx + 2 < w) q[1] = acc[oo].z;
                }
            }
        }
    };
    auto store_out = [&](int k) {
        if (wide && Y0 + MCH * (k + 1) <= Y1)
            store_out_t(k, flag<true>{});
        else
            store_out_t(k, flag<false>{});
    };
    /* phase p: chunk p is staged and filtered horizontally while the raw rows of chunk p + 1 are on their way; then the
     * vertical pass of step p - 1 (its window ends in chunk p).  Chunk p replaces chunk p - 2 in the ring, which the vertical
     * pass of step p - 2 read: the first barrier of a phase. */
    issue_loads(0);
    for (int p = 0; p <= K; p++) {
        const bool full = R - MCH * p >= MCH;
        if (p > 1) lds_barrier();
        if (full)
            stage_hpass_t(p, flag<true>{});
        else
            stage_hpass_t(p, flag<false>{});
        if (p > 1) store_out(p - 2);
        issue_loads(p + 1);
        if (full)
            hpass_t(p, flag<true>{});
        else
            hpass_t(p, flag<false>{});
        if (p == 0) continue;
        lds_barrier(); /* the rows of chunks p - 1 and p are filtered */
        vpass(p - 1);
    }
    store_out(K - 1);
}

template <int HALO>
__global__ MARCH_BOUNDS void k_blur_march(BlurArgs a, BatchDesc bd)
{
    constexpr int HP = (HALO + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float s_t[MNR * (TW + 2 * HP)];
    const int tile = xcd_remap(blockIdx.x, a.tiles_x * a.tiles_y);
    const int strip = tile % a.tiles_x, seg = tile / a.tiles_x;
    if (strip * TW - HP < 0 || strip * TW + TW + HP > a.w)
        march_body<HALO, true>(a, bd, s_t, strip, seg);
    else
        march_body<HALO, false>(a, bd, s_t, strip, seg);
}

}  // namespace

/* Rows per segment (a multiple of 32): segments of EQUAL length, about MARCH_SEG_ROWS rows (seven steps) each, more and shorter
 * ones where a launch over nb images would otherwise have fewer than want_wgs workgroups, never less than two steps where the
 * plane allows.  Sixteen 3840 x 2160 planes per launch, summed time of the level launches of an image's octaves 0 and 1
 * (tools/r04_seg_sweep.sh, one context): 160 / 224 / 288 / 384 / 448 / 544 / 736 / 1088 rows -> 83.6 / 84.4 / 86.9 / 89.3 /
 * 88.2 / 88.8 / 94.6 / 102.4 us; the first version of this function took the LONGEST segment that gave want_wgs workgroups
 * and left the remainder as the last one -- 2144 + 16 rows at sixteen planes, 704 + 704 + 704 + 48 at eight: 129.1 us. */
constexpr int MARCH_SEG_ROWS = 224;
int blur_march_seg_rows(int w, int h, int nb, int want_wgs)
{
    const int  strips = (w + TW - 1) / TW;
    const long per_seg = (long)strips * std::max(nb, 1);
    int        nseg = std::max((h + MARCH_SEG_ROWS - 1) / MARCH_SEG_ROWS, (int)((want_wgs + per_seg - 1) / per_seg));
    nseg = std::max(1, std::min(nseg, std::max(h / (2 * MCH), 1)));
    const int rows = ((h + nseg - 1) / nseg + MCH - 1) / MCH * MCH;
    return std::max(rows, MCH);
}

bool blur_march_supported(const BlurArgs& a, int halo) { return a.dog_off < 0 && halo >= 0 && halo <= 16; }

hipError_t launch_blur_march(BlurArgs a, const BatchDesc& bd, int nb, int halo, int seg_rows, hipStream_t s)
{
    if (!blur_march_supported(a, halo) || seg_rows < MCH || seg_rows % MCH) return hipErrorInvalidValue;
    a.seg_rows = seg_rows;
    a.tiles_x = (a.w + TW - 1) / TW;
    a.tiles_y = (a.h + seg_rows - 1) / seg_rows;
    const dim3 grid(a.tiles_x * a.tiles_y, nb), block(MNT);
#define PS_CASE(H)                                                    \
    if (halo <= H) {                                                  \
        hipLaunchKernelGGL((k_blur_march<H>), grid, block, 0, s, a, bd); \
        return hipGetLastError();                                     \
    }
    PS_CASE(4)
    PS_CASE(5)
    PS_CASE(6)
    PS_CASE(7)
    PS_CASE(8)
    PS_CASE(10)
    PS_CASE(13)
    PS_CASE(16)
#undef PS_CASE
    return hipErrorInvalidValue;
}

}  // namespace popsift_hip
