// ss_heuristics.hip -- weighted common-neighbour scores of node pairs: the per-link precompute next to the sketches in
// HashDataset.__init__ (reference datasets/elph.py:76-77,314 -> heuristics.py:10-70: CN, AA and RA are the same sum
//     score(u, v) = sum_w A[u, w] * (A[v, w] * mult[w])        mult = 1 | 1/log(colsum) | 1/colsum
// over the columns both rows hold).  A is a CSR with sorted, duplicate-free column ids (what scipy hands the reference).
// 16 lanes per pair, 4 pairs per wavefront: the lanes stride over the SHORTER row and binary-search the longer one, so a
// pair costs deg_short/16 * log2(deg_long) probes; fp64 products with the reference's association, one fp32 rounding at
// the end (the reference sums in fp64 and casts with torch.FloatTensor).  HBM-bound gather of two short rows per pair.
#include "ss_common.hpp"

namespace ss {

constexpr int kPairLanes = 16;

__device__ inline double row16_sum_d(double x)
{
    for (int off = 1; off < kPairLanes; off <<= 1) x += __shfl_xor(x, off);
    return x;
}

__global__ __launch_bounds__(256) void common_neighbour_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const double *__restrict__ val, const double *__restrict__ mult,
                                                               int64_t N, const int64_t *__restrict__ links, int64_t B,
                                                               float *__restrict__ out, int32_t *__restrict__ err_flag)
{
    const int l = threadIdx.x & (kPairLanes - 1);
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x / kPairLanes) + threadIdx.x / kPairLanes;
    if (q >= B) return;
    const int64_t u = links[2 * q], v = links[2 * q + 1];
    if (u < 0 || u >= N || v < 0 || v >= N) {
        if (l == 0) {
            out[q] = 0.0f;
            if (err_flag) *err_flag = 1;
        }
        return;
    }
    const int64_t ub = rowptr[u], vb = rowptr[v];
    const int du = (int)(rowptr[u + 1] - ub), dv = (int)(rowptr[v + 1] - vb);
    // walk the shorter row, search the longer one; `swapped` keeps the roles of the reference's product:
    // term = A[src, w] * (A[dst, w] * mult[w])
    const bool swapped = du > dv;
    const int64_t sb = swapped ? vb : ub, lb = swapped ? ub : vb;
    const int ds = swapped ? dv : du, dl = swapped ? du : dv;
    double acc = 0.0;
    for (int i = l; i < ds; i += kPairLanes) {
        const int32_t w = col[sb + i];
        int lo = 0, hi = dl;  // first position in the long row with col >= w
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (col[lb + mid] < w) lo = mid + 1;
            else hi = mid;
        }
        if (lo < dl && col[lb + lo] == w) {
            const double a_short = val ? val[sb + i] : 1.0, a_long = val ? val[lb + lo] : 1.0;
            const double a_src = swapped ? a_long : a_short, a_dst = swapped ? a_short : a_long;
            acc += a_src * (mult ? a_dst * mult[w] : a_dst);
        }
    }
    acc = row16_sum_d(acc);
    if (l == 0) out[q] = (float)acc;
}

}  // namespace ss

extern "C" int ss_common_neighbour_scores(const int64_t *rowptr, const int32_t *col, const double *val, const double *mult,
                                          int64_t N, const int64_t *links, int64_t B, float *out, int32_t *err_flag, void *stream)
{
    using namespace ss;
    if (N < 0 || B < 0 || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (B == 0) return SS_OK;
    if (!rowptr || !col || !links || !out) return SS_ERR_INVALID_ARG;
    const int pairs_per_block = 256 / kPairLanes;
    const int64_t blocks = (B + pairs_per_block - 1) / pairs_per_block;
    if (blocks >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(common_neighbour_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, val, mult, N,
                       links, B, out, err_flag);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
