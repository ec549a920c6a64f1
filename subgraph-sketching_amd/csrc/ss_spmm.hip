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
