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

// one pass over the edge list: are all weights exactly 1 (then every degree is a count and no order matters), the existing self loops
// per node (count, and the index of the LAST one: add_remaining_self_loops is a scatter assignment in edge order)
struct GcnScan {
    int32_t not_unit;   // some weight differs from 1.0f
    int32_t bad_ids;    // some endpoint lies outside [0, N) -- negative ids included: the groupings would wrap them torch-style, PyG does not
    int32_t pad[62];
};
__global__ __launch_bounds__(256) void gcn_scan_edges_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                                             const float *__restrict__ w, int64_t E, int64_t N, GcnScan *__restrict__ scan,
                                                             int32_t *__restrict__ self_count, int32_t *__restrict__ last_self)
{
    bool odd = false, bad = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = row[e], c = col[e];
        odd |= w[e] != 1.0f;
        bad |= (uint64_t)r >= (uint64_t)N || (uint64_t)c >= (uint64_t)N;
        if (r == c && (uint64_t)r < (uint64_t)N) {
            atomicAdd(&self_count[r], 1);
            atomicMax(&last_self[r], (int32_t)e);
        }
    }
    if (__ballot(odd) && (threadIdx.x & (kWave - 1)) == 0) scan->not_unit = 1;  // (plain store of the same value by whoever sees one)
    if (__ballot(bad) && (threadIdx.x & (kWave - 1)) == 0) scan->bad_ids = 1;
}

// dinv / loop_w per node.  Unit weights (scan->not_unit == 0): deg = (entries of the column group - existing self loops) + loop weight,
// exact in fp32 up to 2^24 -- the order of a sum of ones does not matter, and the group need not be sorted.  Otherwise one 16-lane
// group per node: the weights of its in-edges summed in edge order (lane by lane, 16 loads at a time).
__global__ __launch_bounds__(256) void gcn_degree_kernel(const int64_t *__restrict__ rowptr_c, const int32_t *__restrict__ order_c,
                                                         const int64_t *__restrict__ row, const float *__restrict__ w, int64_t N,
                                                         const GcnScan *__restrict__ scan, const int32_t *__restrict__ self_count,
                                                         const int32_t *__restrict__ last_self, float *__restrict__ dinv,
                                                         float *__restrict__ loop_w)
{
    const bool unit = scan->not_unit == 0;  // (uniform)
    if (unit) {
        const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= N) return;
        const int32_t ls = last_self[c];
        const float lw = ls >= 0 ? w[ls] : 1.0f;
        const float deg = (float)(int)(rowptr_c[c + 1] - rowptr_c[c] - self_count[c]) + lw;
        float d = 1.0f / sqrtf(deg);
        if (isinf(d)) d = 0.0f;
        dinv[c] = d;
        loop_w[c] = lw;
        return;
    }
    const int l = threadIdx.x & (kRow - 1);
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kRow; c < N; c += (int64_t)gridDim.x * blockDim.x / kRow) {
        const int64_t e0 = rowptr_c[c], e1 = rowptr_c[c + 1];
        const int32_t ls = last_self[c];
        const float lw = ls >= 0 ? w[ls] : 1.0f;
        float deg = 0.0f;
        const int base = (threadIdx.x & (kWave - 1)) & ~(kRow - 1);
        for (int64_t j0 = e0; j0 < e1; j0 += kRow) {
            const int64_t j = j0 + l;
            const int32_t e = order_c[j < e1 ? j : e0];
            const float we = row[e] == c ? 0.0f : w[e];  // (an existing self loop is not part of the sum; 0 is added exactly ... but
            const bool self = row[e] == c;               //  -0 / NaN weights would not be: skip it instead)
            const int n = (int)(e1 - j0 < kRow ? e1 - j0 : kRow);
            for (int k = 0; k < n; ++k) {  // (group-uniform trip count)
                const float wk = __shfl(we, base + k);
                if (!__shfl((int)self, base + k)) deg += wk;
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
}

// spmm_rows_kernel with the normalised weights formed on the fly: row i's entries are the edge indices e (ascending) with
// row[e] == i; existing self loops are skipped, the node's one remaining loop comes last.  The lanes of a row first fetch one entry
// each (edge index -> column, weight -> the column's dinv: three dependent loads, once per lanes_per_row entries instead of once
// per batch of four) and hand them out with shuffles; the feature rows are then requested four at a time.
__global__ __launch_bounds__(256) void sign_spmm_kernel(const int64_t *__restrict__ rowptr_r, const int32_t *__restrict__ order_r,
                                                        const int64_t *__restrict__ col, const float *__restrict__ w,
                                                        const float *__restrict__ dinv, const float *__restrict__ loop_w, int64_t N,
                                                        const float *__restrict__ x, int F, float *__restrict__ out, int lanes_per_row,
                                                        const GcnScan *__restrict__ scan)
{
    const bool unit = scan->not_unit == 0;  // (uniform) all weights are 1: w[e] need not be gathered (a 64-byte sector per entry)
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_per_wave = kWave / lanes_per_row;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t i_raw = wave * rows_per_wave + lane / lanes_per_row;
    const int64_t i = i_raw < N ? i_raw : N - 1;  // (lanes past the end shadow the last row and do not store: shuffles need every lane)
    const int cl = lane % lanes_per_row, base = lane - cl;
    const int64_t e0 = rowptr_r[i], e1 = rowptr_r[i + 1];
    const int CF = F >> 2;  // float4 chunks per row
    const float di = dinv[i];
    for (int cbase = 0; cbase < CF; cbase += lanes_per_row) {  // (one iteration unless F > 256)
        // lanes past the last chunk (F / 4 not a multiple of the lane group: the padded Planetoid widths 1436, 3704, 500) read the last
        // chunk and do not store -- in EVERY iteration (only the first one was clamped)
        const int c = cbase + cl < CF ? cbase + cl : CF - 1;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int64_t j0 = e0; j0 < e1; j0 += lanes_per_row) {
            // this lane's entry of the stretch
            const int64_t j = j0 + cl;
            const int32_t e = order_r[j < e1 ? j : e0];
            const int64_t raw_c = col[e];
            const bool in_range = (uint64_t)raw_c < (uint64_t)N;  // (an id out of range is reported by ss_gcn_scan_edges; here it must only not be followed)
            const int64_t my_c = in_range ? raw_c : i;
            const float my_nw = di * (unit ? 1.0f : w[e]) * dinv[my_c];
            const int my_use = j < e1 && in_range && my_c != i;
            const int n = (int)(e1 - j0 < lanes_per_row ? e1 - j0 : lanes_per_row);
            for (int k0 = 0; k0 < n; k0 += 4) {  // (uniform per row group)
                float4 r[4];
                float nw[4];
                int use[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int src = base + (k0 + k < lanes_per_row ? k0 + k : 0);
                    const int64_t cj = ((int64_t)__shfl((int)(my_c >> 32), src) << 32) | (uint32_t)__shfl((int)my_c, src);
                    nw[k] = __shfl(my_nw, src);
                    use[k] = k0 + k < n ? __shfl(my_use, src) : 0;
                    r[k] = *reinterpret_cast<const float4 *>(x + (use[k] ? cj : i) * F + 4 * c);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (use[k]) {
                        a.x += r[k].x * nw[k];
                        a.y += r[k].y * nw[k];
                        a.z += r[k].z * nw[k];
                        a.w += r[k].w * nw[k];
                    }
            }
        }
        const float nl = di * loop_w[i] * di;
        const float4 rs = *reinterpret_cast<const float4 *>(x + i * F + 4 * c);
        a.x += rs.x * nl;
        a.y += rs.y * nl;
        a.z += rs.z * nl;
        a.w += rs.w * nl;
        if (i_raw < N && cbase + cl < CF) *reinterpret_cast<float4 *>(out + i * F + 4 * c) = a;
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
// weight.  ss_gcn_scan_edges first (one pass over the edge list: unit weights? existing self loops per node) into `scan`
// (SS_GCN_SCAN_BYTES(N) device bytes); ss_gcn_degree then reads the grouping of the edge indices by column (rowptr_c / order_c:
// ss_csr_group_ids(edge_index[1]) -- STABLE, i.e. followed by ss_csr_sort_rows, unless the weights are all 1: pass scan as that
// call's `skip`).  row = edge_index[0] (device int64[E]), w device fp32[E].  (reference datasets/elph.py:100-101: gcn_norm)
extern "C" size_t ss_gcn_scan_bytes(int64_t N) { return N < 0 ? 0 : sizeof(ss::GcnScan) + (size_t)N * 8; }

extern "C" int ss_gcn_scan_edges(const int64_t *row, const int64_t *col, const float *w, int64_t E, int64_t N, void *scan, void *stream_)
{
    using namespace ss;
    if (N < 0 || E < 0 || !scan) return SS_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    GcnScan *sc = reinterpret_cast<GcnScan *>(scan);
    int32_t *self_count = reinterpret_cast<int32_t *>(sc + 1), *last_self = self_count + N;
    if (hipMemsetAsync(scan, 0, sizeof(GcnScan) + (size_t)N * 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
    if (N && hipMemsetAsync(last_self, 0xFF, (size_t)N * 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
    if (E == 0) return SS_OK;
    if (!row || !col || !w) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gcn_scan_edges_kernel, dim3(2048), dim3(256), 0, stream, row, col, w, E, N, sc, self_count, last_self);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_gcn_degree(const int64_t *rowptr_c, const int32_t *order_c, const int64_t *row, const float *w, int64_t N, const void *scan,
                             float *dinv, float *loop_w, void *stream)
{
    using namespace ss;
    if (N < 0 || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if (!rowptr_c || !order_c || !row || !w || !scan || !dinv || !loop_w) return SS_ERR_INVALID_ARG;
    const GcnScan *sc = reinterpret_cast<const GcnScan *>(scan);
    const int32_t *self_count = reinterpret_cast<const int32_t *>(sc + 1), *last_self = self_count + N;
    const int64_t blocks = (N + 255) / 256;  // (unit weights: one thread per node; else 16-lane groups striding over the nodes)
    hipLaunchKernelGGL(gcn_degree_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr_c, order_c, row, w, N, sc, self_count,
                       last_self, dinv, loop_w);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// out = gcn_norm(A) * x: the product of reference datasets/elph.py:102-107 without the normalised edge list in memory.
// rowptr_r / order_r: the STABLE grouping of the edge indices by row (edge_index[0]); col = edge_index[1] (device int64[E]);
// dinv, loop_w from ss_gcn_degree, scan from ss_gcn_scan_edges; x, out fp32 [N, F], F % 4 == 0.
extern "C" int ss_sign_spmm(const int64_t *rowptr_r, const int32_t *order_r, const int64_t *col, const float *w, const float *dinv,
                            const float *loop_w, const void *scan, int64_t N, const float *x, int32_t F, float *out, void *stream)
{
    using namespace ss;
    if (N < 0 || F <= 0 || (F & 3) || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if (!rowptr_r || !order_r || !col || !w || !dinv || !loop_w || !scan || !x || !out) return SS_ERR_INVALID_ARG;
    int lanes = pow2_ceil(F >> 2);
    if (lanes > kWave) lanes = kWave;
    const int rows_per_block = (256 / kWave) * (kWave / lanes);
    const int64_t blocks = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(sign_spmm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr_r, order_r, col, w, dinv, loop_w, N, x,
                       (int)F, out, lanes, reinterpret_cast<const GcnScan *>(scan));
    SS_LAUNCH_CHECK();
    return SS_OK;
}
