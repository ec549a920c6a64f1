// ss_spmm.hip -- node-feature propagation of HashDataset._generate_sign_features (reference datasets/elph.py:87-110,
// SURVEY 8(f) row N4): out = A_norm * x with A_norm given as a row-grouped CSR with fp32 values.
// torch_sparse.spmm multiplies (fp32), then scatter-adds the products in edge order (sequential on the CPU).  This kernel
// keeps exactly that arithmetic: every output element is accumulated by ONE lane, edge after edge in CSR order (the host
// builds the CSR with a stable sort, so CSR order is the reference's edge order), multiply and add rounded separately
// (-ffp-contract=off).  Mapping: a lane owns one float4 of the feature row; G = 64 / pow2(F/4) rows share a wavefront
// (F = 128: two rows per wave).  HBM-bound gather of (E + N) feature rows -- the same shape as the MinHash table hop.
#include "ss_common.hpp"
#include "ss_walks.hpp"

namespace ss {

__global__ __launch_bounds__(256) void spmm_rows_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const float *__restrict__ val, int64_t N, const float *__restrict__ x, int F,
                                                        float *__restrict__ out, int lanes_per_row)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_per_wave = kWave / lanes_per_row;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t i = wave * rows_per_wave + lane / lanes_per_row;
    if (i >= N) return;
    const int cl = lane % lanes_per_row;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    const int CF = F >> 2;  // float4 chunks per row
    for (int c = cl; c < CF; c += lanes_per_row) {
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int64_t e = e0;
        for (; e + 3 < e1; e += 4) {  // four feature rows requested before the first is used; the adds stay in edge order
            float4 r[4];
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = val[e + k];
                r[k] = *reinterpret_cast<const float4 *>(x + (int64_t)col[e + k] * F + 4 * c);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x += r[k].x * w[k];
                acc.y += r[k].y * w[k];
                acc.z += r[k].z * w[k];
                acc.w += r[k].w * w[k];
            }
        }
        for (; e < e1; ++e) {
            const float w = val[e];
            const float4 r = *reinterpret_cast<const float4 *>(x + (int64_t)col[e] * F + 4 * c);
            acc.x += r.x * w;
            acc.y += r.y * w;
            acc.z += r.z * w;
            acc.w += r.w * w;
        }
        *reinterpret_cast<float4 *>(out + i * F + 4 * c) = acc;
    }
}

// ---- gcn_norm + spmm without materialising the normalised edge list (reference datasets/elph.py:100-108) -------------------------
// torch_geometric's gcn_norm(edge_index, edge_weight, num_nodes) with its defaults [PyG semantics restated: the package is not in
// this image; G15 pins the reference's own function under that restatement]:
//   add_remaining_self_loops: existing self loops (row == col) leave the edge list, every node gets ONE loop behind all other
//   edges, weighted with its existing self loop's weight (the last one in edge order) or 1;  deg[c] = sum of the weights of the
//   edges into c (index_add over `col`, sequential: edge order, the loop last);  dinv = deg^-1/2 with inf -> 0;
//   norm_e = dinv[row_e] * w_e * dinv[col_e]  (left to right).
// torch_sparse.spmm then computes out[row_e] += norm_e * x[col_e] in edge order.  Both accumulations are reproduced bit for bit
// from STABLE groupings of the edge indices (ss_csr_group_ids + ss_csr_sort_rows): by column for the degrees, by row for the product.

// one 16-lane group per node: the weights of its in-edges summed in edge order (lane 0 adds, 16 loads at a time)
__global__ __launch_bounds__(256) void gcn_degree_kernel(const int64_t *__restrict__ rowptr_c, const int32_t *__restrict__ order_c,
                                                         const int64_t *__restrict__ row, const float *__restrict__ w, int64_t N,
                                                         float *__restrict__ dinv, float *__restrict__ loop_w)
{
    const int l = threadIdx.x & (kRow - 1);
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kRow;
    if (c >= N) return;
    const int64_t e0 = rowptr_c[c], e1 = rowptr_c[c + 1];
    float deg = 0.0f, lw = 1.0f;
    const int base = (threadIdx.x & (kWave - 1)) & ~(kRow - 1);
    for (int64_t j0 = e0; j0 < e1; j0 += kRow) {
        const int64_t j = j0 + l;
        const int32_t e = order_c[j < e1 ? j : e0];
        const float we = w[e];
        const bool self = row[e] == c;
        const int n = (int)(e1 - j0 < kRow ? e1 - j0 : kRow);
        for (int k = 0; k < n; ++k) {  // (group-uniform trip count)
            const float wk = __shfl(we, base + k);
            const bool sk = __shfl((int)self, base + k) != 0;
            if (sk) lw = wk; else deg += wk;
        }
    }
    deg += lw;
    if (l == 0) {
        float d = 1.0f / sqrtf(deg);  // deg.pow(-0.5): torch evaluates it as the reciprocal of the square root, both correctly rounded
        if (isinf(d)) d = 0.0f;
        dinv[c] = d;
        loop_w[c] = lw;
    }
}

// spmm_rows_kernel with the normalised weights formed on the fly: row i's entries are the edge indices e (ascending) with
// row[e] == i; existing self loops are skipped, the node's one remaining loop comes last
__global__ __launch_bounds__(256) void sign_spmm_kernel(const int64_t *__restrict__ rowptr_r, const int32_t *__restrict__ order_r,
                                                        const int64_t *__restrict__ col, const float *__restrict__ w,
                                                        const float *__restrict__ dinv, const float *__restrict__ loop_w, int64_t N,
                                                        const float *__restrict__ x, int F, float *__restrict__ out, int lanes_per_row)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_per_wave = kWave / lanes_per_row;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t i = wave * rows_per_wave + lane / lanes_per_row;
    if (i >= N) return;
    const int cl = lane % lanes_per_row;
    const int64_t e0 = rowptr_r[i], e1 = rowptr_r[i + 1];
    const int CF = F >> 2;  // float4 chunks per row
    const float di = dinv[i];
    for (int c = cl; c < CF; c += lanes_per_row) {
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int64_t j0 = e0; j0 < e1; j0 += 4) {  // four feature rows requested before the first is used; the adds stay in edge order
            float4 r[4];
            float nw[4];
            bool use[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int32_t e = order_r[j0 + k < e1 ? j0 + k : e0];  // (a valid entry for the slots past the end; not used)
                const int64_t cj = col[e];
                use[k] = j0 + k < e1 && cj != i;
                nw[k] = di * w[e] * dinv[cj];
                r[k] = *reinterpret_cast<const float4 *>(x + cj * F + 4 * c);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (use[k]) {
                    acc.x += r[k].x * nw[k];
                    acc.y += r[k].y * nw[k];
                    acc.z += r[k].z * nw[k];
                    acc.w += r[k].w * nw[k];
                }
        }
        const float nl = di * loop_w[i] * di;
        const float4 rs = *reinterpret_cast<const float4 *>(x + i * F + 4 * c);
        acc.x += rs.x * nl;
        acc.y += rs.y * nl;
        acc.z += rs.z * nl;
        acc.w += rs.w * nl;
        *reinterpret_cast<float4 *>(out + i * F + 4 * c) = acc;
    }
}

}  // namespace ss

extern "C" int ss_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, int64_t N, const float *x, int32_t F,
                           float *out, void *stream)
{
    using namespace ss;
    if (N < 0 || F <= 0 || (F & 3) || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if (!rowptr || !col || !val || !x || !out) return SS_ERR_INVALID_ARG;
    int lanes = pow2_ceil(F >> 2);
    if (lanes > kWave) lanes = kWave;
    const int rows_per_block = (256 / kWave) * (kWave / lanes);
    const int64_t blocks = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(spmm_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, val, N, x, (int)F, out,
                       lanes);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// dinv[c] = (sum of the weights into c, existing self loops replaced by one loop of their weight or 1)^-1/2 (inf -> 0) and that loop's
// weight, from the STABLE grouping of the edge indices by column (rowptr_c / order_c: ss_csr_group_ids(edge_index[1]) +
// ss_csr_sort_rows); row = edge_index[0] (device int64[E]), w device fp32[E].  (reference datasets/elph.py:100-101: gcn_norm)
extern "C" int ss_gcn_degree(const int64_t *rowptr_c, const int32_t *order_c, const int64_t *row, const float *w, int64_t N, float *dinv,
                             float *loop_w, void *stream)
{
    using namespace ss;
    if (N < 0 || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if (!rowptr_c || !order_c || !row || !w || !dinv || !loop_w) return SS_ERR_INVALID_ARG;
    const int64_t blocks = (N * kRow + 255) / 256;
    hipLaunchKernelGGL(gcn_degree_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr_c, order_c, row, w, N, dinv, loop_w);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// out = gcn_norm(A) * x: the product of reference datasets/elph.py:102-107 without the normalised edge list in memory.
// rowptr_r / order_r: the STABLE grouping of the edge indices by row (edge_index[0]); col = edge_index[1] (device int64[E]);
// dinv, loop_w from ss_gcn_degree; x, out fp32 [N, F], F % 4 == 0.
extern "C" int ss_sign_spmm(const int64_t *rowptr_r, const int32_t *order_r, const int64_t *col, const float *w, const float *dinv,
                            const float *loop_w, int64_t N, const float *x, int32_t F, float *out, void *stream)
{
    using namespace ss;
    if (N < 0 || F <= 0 || (F & 3) || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if (!rowptr_r || !order_r || !col || !w || !dinv || !loop_w || !x || !out) return SS_ERR_INVALID_ARG;
    int lanes = pow2_ceil(F >> 2);
    if (lanes > kWave) lanes = kWave;
    const int rows_per_block = (256 / kWave) * (kWave / lanes);
    const int64_t blocks = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(sign_spmm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr_r, order_r, col, w, dinv, loop_w, N, x,
                       (int)F, out, lanes);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
