/*
 * match_mfma.hip -- matrix-core screening for the brute-force matcher ("the only place a dense
 * contraction -- and hence MFMA -- would ever apply").
 *
 * The result of FeaturesDev::match is index work (best, second, accept): it has to be the reference's, so the
 * final distances are always computed in the reference's own arithmetic (match.hip).  But WHICH right
 * descriptors can be the two nearest of a left one is a question a GEMM answers:
 *     d(l, r) = |l|^2 + |r|^2 - 2 l.r
 * evaluated with v_mfma_f32_32x32x2_f32 (exact f32 products, k-ordered FMA chain) is within eps = 3e-5 (|l|^2 + |r|^2)
 * of the reference's sum of squared differences, so every right descriptor whose screened distance is not within
 * 2 eps of the second smallest screened distance is out -- typically all but two or three.
 *   k_norms          |x|^2 of every descriptor
 *   k_match_screen   GEMM tiles on the matrix cores; every lane keeps the 4 smallest screened distances of "its"
 *                    left descriptor (the 32x32 accumulator tile has the left index on the lane and 16 right
 *                    indices in registers: no cross-lane traffic in the epilogue)
 *   k_match_select   one wave per left descriptor: merge the lanes' / splits' candidates, check that the 4th
 *                    screened distance lies outside the 2 eps margin (otherwise the row is put on a list for the
 *                    exact brute-force kernel), recompute the survivors' distances exactly as l2_in_t0 does
 *                    (features.cu:157-176: 32 lanes x float4 FMA chain, shuffle_down tree) and pick best / second
 *                    with ties to the lower index.
 * Rows on the list (exact duplicates in the right set, ties inside the margin) go through k_match with a row
 * indirection, without a host round trip.  tests/test_gpu_match.py compares both paths with the oracle bit for bit.
 */
#include "devfeatures.h"
#include "kernels.h"

namespace popsift_hip {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int S_LB = 128;   /* left descriptors per workgroup: 32 per wave (the N dimension of its MFMA tiles) */
constexpr int S_RB = 128;   /* right descriptors per LDS tile: 4 row blocks of 32 (the M dimension)            */
constexpr int S_ROW = 132;  /* floats per LDS row: +4 keeps the 16 rows of a ds_read_b128 group on distinct slots */
constexpr int S_K = 4;      /* screened candidates kept per left descriptor */

struct Cand {
    float d[S_K];
    int   i[S_K];
};

__device__ __forceinline__ bool cand_less(float d, int i, float e, int j) { return d < e || (d == e && i < j); }

/* sorted insert into the 4 smallest (ascending; ties to the lower index) */
__device__ __forceinline__ void cand_insert(Cand& c, float d, int i)
{
    if (!cand_less(d, i, c.d[S_K - 1], c.i[S_K - 1])) return;
    c.d[S_K - 1] = d;
    c.i[S_K - 1] = i;
#pragma unroll
    for (int k = S_K - 1; k > 0; k--) {
        if (cand_less(c.d[k], c.i[k], c.d[k - 1], c.i[k - 1])) {
            const float td = c.d[k];
            const int   ti = c.i[k];
            c.d[k] = c.d[k - 1];
            c.i[k] = c.i[k - 1];
            c.d[k - 1] = td;
            c.i[k - 1] = ti;
        }
    }
}

__global__ __launch_bounds__(256) void k_norms(const float* __restrict__ desc, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v4f* p = (const v4f*)(desc + (size_t)i * 128);
    float      s = 0.0f;
    for (int k = 0; k < 32; k++) {
        const v4f v = p[k];
        s = fmaf(v.x, v.x, s);
        s = fmaf(v.y, v.y, s);
        s = fmaf(v.z, v.z, s);
        s = fmaf(v.w, v.w, s);
    }
    out[i] = s;
}

/*
 * grid (ceil(l_len / 128), n_split), 256 lanes.  Wave w owns left descriptors l0 + 32 w .. +31 (lane & 31) and, per
 * right tile, the four 32 x 32 products of the tile's row blocks with them.  The k dimension is split between
 * the two lane halves (half h takes k = 64 h .. 64 h + 63; a 32x32x2 MFMA step adds one k of each half), so both
 * operands are contiguous per lane: the left row chunk lives in 64 registers for the whole sweep, the right one is
 * read from LDS with ds_read_b128.
 */
__global__ __launch_bounds__(256, 2) void k_match_screen(const float* __restrict__ ldesc, int l_len,
                                                         const float* __restrict__ rdesc, int r_len,
                                                         const float* __restrict__ rnorm, int n_split,
                                                         Cand* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float s_r[S_RB * S_ROW];
    __shared__ float                               s_rn[S_RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int l = blockIdx.x * S_LB + wave * 32 + col;

    /* B operand: this lane's half of its left descriptor (zeros past the end) */
    float bl[64];
    {
        const v4f* p = (const v4f*)(ldesc + (size_t)min(l, l_len - 1) * 128 + 64 * half);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v4f v = p[k];
            if (l >= l_len) v = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            bl[4 * k + 0] = v.x;
            bl[4 * k + 1] = v.y;
            bl[4 * k + 2] = v.z;
            bl[4 * k + 3] = v.w;
        }
    }
    Cand best;
#pragma unroll
    for (int k = 0; k < S_K; k++) {
        best.d[k] = INFINITY;
        best.i[k] = 0;
    }

    const int n_tiles = (r_len + S_RB - 1) / S_RB;
    for (int tile = blockIdx.y; tile < n_tiles; tile += n_split) {
        const int r0 = tile * S_RB;
        __syncthreads(); /* previous tile consumed */
        {
            /* all 16 loads of a lane in flight before the first LDS store (unconditional: rows past the end are
             * clamped and zeroed afterwards -- a load under a condition makes the compiler wait for each one) */
            v4f st[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int c = tid + 256 * k, row = c >> 5, ch = c & 31;
                st[k] = *(const v4f*)(rdesc + (size_t)min(r0 + row, r_len - 1) * 128 + 4 * ch);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int c = tid + 256 * k, row = c >> 5, ch = c & 31;
                if (r0 + row >= r_len) st[k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                *(v4f*)(s_r + row * S_ROW + 4 * ch) = st[k];
            }
        }
        if (tid < S_RB) s_rn[tid] = (r0 + tid < r_len) ? rnorm[r0 + tid] : INFINITY;
        __syncthreads();

        v16f acc[4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[m][v] = 0.0f;
        const float* arow = s_r + col * S_ROW + 64 * half; /* A operand: row (32 m + col), this half's k range */
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
            v4f a[4];
#pragma unroll
            for (int m = 0; m < 4; m++) a[m] = *(const v4f*)(arow + m * 32 * S_ROW + 4 * k4);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bl[4 * k4 + 0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bl[4 * k4 + 1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bl[4 * k4 + 2], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bl[4 * k4 + 3], acc[m], 0, 0, 0);
            }
        }
        /* C/D layout: column = lane & 31 (the left descriptor), row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5).
         * |l|^2 is the same for every candidate of a lane: it is added in k_match_select. */
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int   row = 32 * m + (v & 3) + 8 * (v >> 2) + 4 * half;
                const float d = fmaf(-2.0f, acc[m][v], s_rn[row]); /* INFINITY for rows past the end */
                cand_insert(best, d, r0 + row);
            }
    }
    /* the two halves hold different right rows of the same left descriptor */
    Cand other;
#pragma unroll
    for (int k = 0; k < S_K; k++) {
        other.d[k] = __shfl_xor(best.d[k], 32);
        other.i[k] = __shfl_xor(best.i[k], 32);
    }
#pragma unroll
    for (int k = 0; k < S_K; k++) cand_insert(best, other.d[k], other.i[k]);
    if (half == 0 && l < l_len) partial[(size_t)l * n_split + blockIdx.y] = best;
}

/* the reference's distance of one pair, 32 lanes per pair (two pairs per wave): features.cu:157-176 */
__device__ __forceinline__ float exact_distance(const float* __restrict__ lrow, const float* __restrict__ rrow, int t)
{
    const v4f a = *(const v4f*)(lrow + 4 * t), b = *(const v4f*)(rrow + 4 * t);
    const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z, w = a.w - b.w;
    float       res = fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x)));
    res += __shfl_down(res, 16, 32);
    res += __shfl_down(res, 8, 32);
    res += __shfl_down(res, 4, 32);
    res += __shfl_down(res, 2, 32);
    res += __shfl_down(res, 1, 32);
    return res; /* valid in lane t == 0 of each 32-lane group */
}

/* one wave per left descriptor */
__global__ __launch_bounds__(256) void k_match_select(const float* __restrict__ ldesc, int l_len,
                                                      const float* __restrict__ rdesc, int r_len,
                                                      const float* __restrict__ lnorm, const float* __restrict__ rnorm,
                                                      const Cand* __restrict__ partial, int n_split,
                                                      popsift_hip_match* __restrict__ out, int* __restrict__ redo_list,
                                                      int* __restrict__ redo_count)
{
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= l_len) return;
    /* merge the splits (every lane does the same few inserts: wave-uniform data) */
    Cand c = partial[(size_t)l * n_split];
    for (int s = 1; s < n_split; s++) {
        const Cand o = partial[(size_t)l * n_split + s];
#pragma unroll
        for (int k = 0; k < S_K; k++) cand_insert(c, o.d[k], o.i[k]);
    }
    const float ln = lnorm[l];
    /* screened distance = ln + d; margin 2 eps with eps = 3e-5 (|l|^2 + |r|^2) >= the GEMM-form rounding plus the
     * reference's own (both ~ 1e-5 relative for 128 terms) */
    const float rn2 = (c.d[1] < INFINITY) ? rnorm[c.i[1]] : 0.0f;
    const float eps = 3e-5f * (ln + fmaxf(rn2, ln));
    const bool  separable = !(c.d[S_K - 1] < INFINITY) || (c.d[S_K - 1] > c.d[1] + 2.0f * eps);
    if (!separable) {
        if (lane == 0) redo_list[atomicAdd(redo_count, 1)] = l;
        return;
    }
    /* exact distances of the (at most S_K - 1 relevant, all S_K evaluated) survivors, two per step */
    float       ed[S_K];
    const int   t = lane & 31, grp = lane >> 5;
    const float* lrow = ldesc + (size_t)l * 128;
#pragma unroll
    for (int k = 0; k < S_K; k += 2) {
        const int   idx = c.i[k + grp];
        const bool  valid = c.d[k + grp] < INFINITY;
        const float e = exact_distance(lrow, rdesc + (size_t)(valid ? idx : 0) * 128, t);
        const float e0 = __shfl(e, 0), e1 = __shfl(e, 32);
        ed[k] = (c.d[k] < INFINITY) ? e0 : INFINITY;
        ed[k + 1] = (c.d[k + 1] < INFINITY) ? e1 : INFINITY;
    }
    if (lane == 0) {
        float v1 = INFINITY, v2 = INFINITY;
        int   i1 = 0, i2 = 0; /* features.cu:185-188 */
#pragma unroll
        for (int k = 0; k < S_K; k++) {
            if (!(c.d[k] < INFINITY)) continue;
            const float d = ed[k];
            const int   i = c.i[k];
            if (cand_less(d, i, v1, i1)) {
                v2 = v1;
                i2 = i1;
                v1 = d;
                i1 = i;
            } else if (cand_less(d, i, v2, i2)) {
                v2 = d;
                i2 = i;
            }
        }
        popsift_hip_match m;
        m.best = i1;
        m.second = i2;
        m.accept = (__fdiv_rn(v1, v2) < 0.8f) ? 1 : 0;
        m.dist_best = v1;
        m.dist_second = v2;
        out[l] = m;
    }
}

}  // namespace

int screen_splits(int l_len, int r_len)
{
    const int l_blocks = (l_len + S_LB - 1) / S_LB;
    const int tiles = (r_len + S_RB - 1) / S_RB;
    int       s = (1024 + l_blocks - 1) / std::max(l_blocks, 1); /* >= 2 resident workgroups per CU x 2 rounds */
    s = std::min(s, std::max(tiles, 1));
    return std::max(s, 1);
}

size_t screen_partial_bytes(int l_len, int n_split) { return (size_t)std::max(l_len, 1) * n_split * sizeof(Cand); }

hipError_t launch_norms(const float* desc, int n, float* out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_norms, dim3((n + 255) / 256), dim3(256), 0, s, desc, n, out);
    return hipGetLastError();
}

hipError_t launch_match_screen(const float* ldesc, int l_len, const float* rdesc, int r_len, const float* lnorm,
                               const float* rnorm, int n_split, void* partial, popsift_hip_match* out, int* redo_list,
                               int* redo_count, hipStream_t s)
{
    if (l_len <= 0) return hipSuccess;
    hipError_t err = hipMemsetAsync(redo_count, 0, sizeof(int), s);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_match_screen, dim3((l_len + S_LB - 1) / S_LB, n_split), dim3(256), 0, s, ldesc, l_len, rdesc, r_len,
                       rnorm, n_split, (Cand*)partial);
    hipLaunchKernelGGL(k_match_select, dim3((l_len + 3) / 4), dim3(256), 0, s, ldesc, l_len, rdesc, r_len, lnorm, rnorm,
                       (const Cand*)partial, n_split, out, redo_list, redo_count);
    return hipGetLastError();
}

}  // namespace popsift_hip
