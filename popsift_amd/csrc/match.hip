/*
 * match.hip -- brute-force 2-nearest-neighbour matching of two device-resident descriptor sets
 * (SURVEY N3).  Reference: FeaturesDev::match / compute_distance / l2_in_t0 (features.cu:157-300):
 * one 32-thread block per left descriptor walks ALL right descriptors one after the other, the 32
 * lanes each square-and-sum one float4 of the difference, a shuffle tree adds the 32 partial sums and
 * lane 0 keeps the two smallest distances (strict '<', so ties go to the lower index); a left
 * descriptor is "accepted" if best / second < 0.8 (squared distances).
 *
 * Here: a workgroup owns 32 left descriptors and streams the right set through LDS in tiles of 64;
 * every lane evaluates a 4 x 2 block of pairs from LDS-resident operands (each right tile is read from
 * HBM / L2 once per 32 left descriptors instead of once per left descriptor, each LDS operand is
 * used for 2 resp. 4 pairs).  A distance is computed in exactly the reference's arithmetic -- the 32
 * per-float4 partial sums x*x + y*y + z*z + w*w as an FMA chain, added in the order of the
 * shuffle_down(16, 8, 4, 2, 1) tree -- so that best / second / accept are bit-identical to a serial
 * restatement (the oracle); the tree is walked depth-first, so a pair needs 6 live registers, not 32.
 * "two smallest with ties to the lower index" is order-independent (lexicographic (distance, index)),
 * which is what allows lanes, and several workgroups per left block for small left sets, to keep
 * private candidates that are merged at the end.
 */
#include "devfeatures.h"
#include "kernels.h"

namespace popsift_hip {
namespace {

constexpr int M_LT = 32;    /* left descriptors per workgroup */
constexpr int M_RT = 64;    /* right descriptors per LDS tile */
constexpr int M_ROW = 132;  /* floats per LDS row: 128 + 4 padding (conflict-free ds_read_b128) */
constexpr int M_LPT = 4;    /* left rows per lane  */
constexpr int M_RPT = 2;    /* right rows per lane */
constexpr int M_PAIRS = M_LPT * M_RPT;

typedef float v4f __attribute__((ext_vector_type(4)));

struct Top2 {
    float v1, v2;
    int   i1, i2;
};

__device__ __forceinline__ bool lex_less(float d, int i, float e, int j) { return d < e || (d == e && i < j); }

__device__ __forceinline__ void top2_insert(Top2& t, float d, int i)
{
    if (lex_less(d, i, t.v1, t.i1)) {
        t.v2 = t.v1;
        t.i2 = t.i1;
        t.v1 = d;
        t.i1 = i;
    } else if (lex_less(d, i, t.v2, t.i2)) {
        t.v2 = d;
        t.i2 = i;
    }
}

struct Acc {
    float v[M_PAIRS];
};

/* partial sum of float4 chunk `ch` for the lane's 4 x 2 pairs: l2_in_t0's per-lane value
 * (features.cu:159-170; nvcc contracts the sum of squares into an FMA chain) */
__device__ __forceinline__ Acc leaf(const float* __restrict__ lrow, const float* __restrict__ rrow, int ch)
{
    v4f l[M_LPT], r[M_RPT];
#pragma unroll
    for (int a = 0; a < M_LPT; a++) l[a] = *(const v4f*)(lrow + a * M_ROW + 4 * ch);
#pragma unroll
    for (int b = 0; b < M_RPT; b++) r[b] = *(const v4f*)(rrow + b * M_ROW + 4 * ch);
    Acc out;
#pragma unroll
    for (int a = 0; a < M_LPT; a++)
#pragma unroll
        for (int b = 0; b < M_RPT; b++) {
            const float x = l[a].x - r[b].x, y = l[a].y - r[b].y, z = l[a].z - r[b].z, w = l[a].w - r[b].w;
            out.v[a * M_RPT + b] = fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x)));
        }
    return out;
}

__device__ __forceinline__ Acc add(const Acc& a, const Acc& b)
{
    Acc s;
#pragma unroll
    for (int p = 0; p < M_PAIRS; p++) s.v[p] = a.v[p] + b.v[p];
    return s;
}

/*
 * The shuffle_down(16, 8, 4, 2, 1) tree of l2_in_t0 (features.cu:171-175) over the 32 per-chunk sums p:
 *   n1(i) = p(i) + p(i+16), n2(i) = n1(i) + n1(i+8), n3(i) = n2(i) + n2(i+4), n4(i) = n3(i) + n3(i+2),
 *   result = n4(0) + n4(1).
 * Walked depth-first: the eight n2 nodes in the order i = 0,4,2,6,1,5,3,7 (bit-reversed counter) with a
 * three-entry stack, so a pair holds at most 5 partial values at any time.  The loop is deliberately NOT
 * unrolled: unrolled, the compiler hoists all 192 LDS reads of a tile to the top and spills.
 */
__device__ __forceinline__ Acc pair_distances(const float* __restrict__ lrow, const float* __restrict__ rrow)
{
    Acc h0 = {}, h1 = {}, h2 = {}, v = {};
#pragma nounroll
    for (int j = 0; j < 8; j++) {
        const int i = ((j & 1) << 2) | (j & 2) | ((j & 4) >> 2);
        v = add(add(leaf(lrow, rrow, i), leaf(lrow, rrow, i + 16)), add(leaf(lrow, rrow, i + 8), leaf(lrow, rrow, i + 24)));
        if (j & 1) {
            v = add(h0, v); /* n3 */
            if (j & 2) {
                v = add(h1, v); /* n4 */
                if (j & 4) v = add(h2, v); /* the root */
                else h2 = v;
            } else {
                h1 = v;
            }
        } else {
            h0 = v;
        }
    }
    return v;
}

/*
 * grid (ceil(l_len / 32), n_split): workgroup (bx, by) matches left rows 32*bx.. against the right
 * tiles by, by + n_split, ...; partial[(l * n_split + by)] receives its two best candidates.
 */
__global__ __launch_bounds__(256, 2) void k_match(const float* __restrict__ ldesc, int l_len,
                                               const float* __restrict__ rdesc, int r_len, int n_split,
                                               Top2* __restrict__ partial, const int* __restrict__ rows,
                                               const int* __restrict__ n_rows, int first_row, int row_end)
{
    /* rows != null: only the left descriptors rows[first_row .. min(*n_rows, row_end)) -- the list the screening
     * pass could not decide; positions in that list index `partial` */
    if (rows) l_len = min(min(*n_rows, row_end), l_len);
    if (first_row + (int)blockIdx.x * M_LT >= l_len) return; /* whole workgroup, before any barrier */
    __shared__ float s_l[M_LT * M_ROW];
    __shared__ float s_r[M_RT * M_ROW];
    const int        tid = threadIdx.x;
    const int        lg = tid >> 5, rg = tid & 31; /* lane's left rows 4*lg.., right rows 2*rg.. */
    const int        l0 = first_row + blockIdx.x * M_LT;

    /* left tile, zero rows past the end */
    for (int c = tid; c < M_LT * 32; c += 256) {
        const int row = c >> 5, ch = c & 31;
        v4f       v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (l0 + row < l_len) v = *(const v4f*)(ldesc + (size_t)(rows ? rows[l0 + row] : l0 + row) * 128 + 4 * ch);
        *(v4f*)(s_l + row * M_ROW + 4 * ch) = v;
    }

    Top2 best[M_LPT];
#pragma unroll
    for (int a = 0; a < M_LPT; a++) best[a] = Top2{INFINITY, INFINITY, 0, 0}; /* features.cu:185-188 */

    const int n_tiles = (r_len + M_RT - 1) / M_RT;
    v4f       stage[8]; /* this lane's share of the next right tile (64 rows x 32 chunks / 256 lanes) */
    auto      fetch = [&](int tile) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = tid + 256 * k, row = c >> 5, ch = c & 31;
            const int r = tile * M_RT + row;
            /* unconditional (clamped) loads: a load under a condition is waited for one by one; rows past the end
             * are never inserted (their index test fails), so their values do not matter */
            stage[k] = *(const v4f*)(rdesc + (size_t)min(r, r_len - 1) * 128 + 4 * ch);
        }
    };
    int tile = blockIdx.y;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += n_split) {
        __syncthreads(); /* previous tile fully consumed (and the left tile written) */
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = tid + 256 * k, row = c >> 5, ch = c & 31;
            *(v4f*)(s_r + row * M_ROW + 4 * ch) = stage[k];
        }
        __syncthreads();
        if (tile + n_split < n_tiles) fetch(tile + n_split); /* in flight during the arithmetic */

        const Acc d = pair_distances(s_l + (M_LPT * lg) * M_ROW, s_r + (M_RPT * rg) * M_ROW);
#pragma unroll
        for (int b = 0; b < M_RPT; b++) { /* increasing right index per lane */
            const int r = tile * M_RT + M_RPT * rg + b;
            if (r < r_len) {
#pragma unroll
                for (int a = 0; a < M_LPT; a++) top2_insert(best[a], d.v[a * M_RPT + b], r);
            }
        }
    }

    /* merge the 32 lanes that share a left row */
    __syncthreads();
    Top2* s_top = (Top2*)s_r; /* [32 left rows][32 lanes] = 16 KB */
#pragma unroll
    for (int a = 0; a < M_LPT; a++) s_top[(M_LPT * lg + a) * 32 + rg] = best[a];
    __syncthreads();
    if (tid < M_LT && l0 + tid < l_len) {
        Top2 t = s_top[tid * 32];
        for (int k = 1; k < 32; k++) {
            const Top2 o = s_top[tid * 32 + k];
            top2_insert(t, o.v1, o.i1);
            top2_insert(t, o.v2, o.i2);
        }
        partial[(size_t)(l0 + tid) * n_split + blockIdx.y] = t;
    }
}

/* merge the per-split candidates; accept as in features.cu:217-218 */
__global__ __launch_bounds__(256) void k_match_finish(const Top2* __restrict__ partial, int l_len, int n_split,
                                                      popsift_hip_match* __restrict__ out, const int* __restrict__ rows,
                                                      const int* __restrict__ n_rows, int first_row, int row_end)
{
    const int l = first_row + blockIdx.x * 256 + threadIdx.x;
    if (rows) l_len = min(min(*n_rows, row_end), l_len);
    if (l >= l_len) return;
    Top2 t = partial[(size_t)l * n_split];
    for (int k = 1; k < n_split; k++) {
        const Top2 o = partial[(size_t)l * n_split + k];
        top2_insert(t, o.v1, o.i1);
        top2_insert(t, o.v2, o.i2);
    }
    popsift_hip_match m;
    m.best = t.i1;
    m.second = t.i2;
    m.accept = (__fdiv_rn(t.v1, t.v2) < 0.8f) ? 1 : 0;
    m.dist_best = t.v1;
    m.dist_second = t.v2;
    out[rows ? rows[l] : l] = m;
}

/* prep_features writing into a FeaturesDev (sift_pyramid.cu:323-345): indices -> device pointers */
__global__ __launch_bounds__(256) void k_clone_features(const popsift_hip_feature* __restrict__ in, int n,
                                                        float* __restrict__ desc_base, DevFeature* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const popsift_hip_feature f = in[i];
    DevFeature                o;
    o.debug_octave = f.debug_octave;
    o.xpos = f.xpos;
    o.ypos = f.ypos;
    o.sigma = f.sigma;
    o.num_ori = f.num_ori;
    o.pad = 0;
#pragma unroll
    for (int k = 0; k < POPSIFT_HIP_ORI_MAX; k++) {
        o.orientation[k] = f.orientation[k];
        o.desc[k] = f.desc_idx[k] >= 0 ? desc_base + (size_t)f.desc_idx[k] * 128 : nullptr;
    }
    out[i] = o;
}

}  // namespace

hipError_t launch_clone_features(const popsift_hip_feature* feats, int n_feat, float* desc_base, void* out, hipStream_t s)
{
    if (n_feat <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_clone_features, dim3((n_feat + 255) / 256), dim3(256), 0, s, feats, n_feat, desc_base,
                       (DevFeature*)out);
    return hipGetLastError();
}

int match_splits(int l_len, int r_len)
{
    const int l_blocks = (l_len + M_LT - 1) / M_LT;
    const int tiles = (r_len + M_RT - 1) / M_RT;
    int       s = (2048 + l_blocks - 1) / std::max(l_blocks, 1); /* aim at >= 8 workgroups per CU */
    s = std::min(s, std::max(tiles, 1));
    return std::max(s, 1);
}

hipError_t launch_match(const float* ldesc, int l_len, const float* rdesc, int r_len, int n_split, void* partial,
                        popsift_hip_match* out, const int* rows, const int* n_rows, int first_row, int row_end,
                        hipStream_t s)
{
    /* without a row list: all of [0, l_len).  With one: list positions [first_row, row_end); the launch is sized for
     * that range and workgroups past the device-side count leave at once */
    const int begin = rows ? first_row : 0;
    const int end = rows ? std::min(row_end, l_len) : l_len;
    if (end <= begin) return hipSuccess;
    const int cover = end - begin;
    hipLaunchKernelGGL(k_match, dim3((cover + M_LT - 1) / M_LT, n_split), dim3(256), 0, s, ldesc, l_len, rdesc, r_len, n_split,
                       (Top2*)partial, rows, n_rows, begin, end);
    hipLaunchKernelGGL(k_match_finish, dim3((cover + 255) / 256), dim3(256), 0, s, (const Top2*)partial, l_len, n_split, out,
                       rows, n_rows, begin, end);
    return hipGetLastError();
}

size_t match_partial_bytes(int l_len, int n_split) { return (size_t)std::max(l_len, 1) * n_split * sizeof(Top2); }

}  // namespace popsift_hip
