// ss_count.hip -- HLL++ cardinality of register rows (reference hashing.py:212-232).
// One 16-byte chunk per lane, SG = min(64, M/16) lanes per row, 64/SG rows per wavefront.
#include "ss_common.hpp"

namespace ss {

constexpr int kCountRows = 4;  // rows per lane group and workgroup pass: 4 independent 16-byte loads in flight per lane

__global__ __launch_bounds__(256) void hll_count_kernel(const uint8_t *__restrict__ regs, int64_t n, int M,
                                                        float *__restrict__ out, int64_t out_stride, ss_hll_params prm)
{
    __shared__ EstimatorLds lds;
    const EstimatorTables est = stage_tables(lds, prm);
    const int CH = M >> 4;
    const int SG = CH > kWave ? kWave : CH;  // M is a power of two >= 16, so CH is a power of two
    const int G = kWave / SG;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / SG, cl = lane % SG;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t row0 = (wave * G + g) * kCountRows;
    int nonzero[kCountRows];
    float hsum[kCountRows];
#pragma unroll
    for (int r = 0; r < kCountRows; ++r) {
        nonzero[r] = 0;
        hsum[r] = 0.0f;
    }
    for (int c = cl; c < CH; c += SG) {
        u32x4 x[kCountRows];
#pragma unroll
        for (int r = 0; r < kCountRows; ++r)
            x[r] = row0 + r < n ? *reinterpret_cast<const u32x4 *>(regs + (row0 + r) * M + 16 * c) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < kCountRows; ++r) {
            hll_dword_stats(x[r].x, nonzero[r], hsum[r]);
            hll_dword_stats(x[r].y, nonzero[r], hsum[r]);
            hll_dword_stats(x[r].z, nonzero[r], hsum[r]);
            hll_dword_stats(x[r].w, nonzero[r], hsum[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < kCountRows; ++r) {
        for (int off = 1; off < SG; off <<= 1) {
            nonzero[r] += __shfl_xor(nonzero[r], off);
            hsum[r] += __shfl_xor(hsum[r], off);
        }
        // lane r of the group finishes row r (the estimator is a few dozen dependent instructions: 4 lanes run it at once)
    }
    int my_nz = nonzero[0];
    float my_hs = hsum[0];
#pragma unroll
    for (int r = 1; r < kCountRows; ++r) {
        my_nz = (cl == r) ? nonzero[r] : my_nz;
        my_hs = (cl == r) ? hsum[r] : my_hs;
    }
    const int my_r = SG >= kCountRows ? cl : 0;
    if (SG >= kCountRows) {
        if (cl < kCountRows && row0 + my_r < n) out[(row0 + my_r) * out_stride] = hll_estimate(est, M - my_nz, my_hs);
    } else {  // M == 16 or 32: fewer lanes than rows per group
        if (cl == 0) {
#pragma unroll
            for (int r = 0; r < kCountRows; ++r)
                if (row0 + r < n) out[(row0 + r) * out_stride] = hll_estimate(est, M - nonzero[r], hsum[r]);
        }
    }
}

__global__ __launch_bounds__(256) void estimate_bias_kernel(const float *__restrict__ e, int64_t n, float *__restrict__ out,
                                                            int refine, ss_hll_params prm)
{
    __shared__ EstimatorLds lds;
    const EstimatorTables est = stage_tables(lds, prm);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = e[i];
        const float b = bias_of_6nn(est, x);
        out[i] = refine ? (x <= est.five_m ? x - b : x) : b;
    }
}

}  // namespace ss

extern "C" int ss_estimate_bias(const float *e, int64_t n, const ss_hll_params *prm, float *out, int32_t refine, void *stream)
{
    using namespace ss;
    if (n < 0) return SS_ERR_INVALID_ARG;
    const int rc = check_params(prm);
    if (rc != SS_OK) return rc;
    if (n == 0) return SS_OK;
    if (!e || !out) return SS_ERR_INVALID_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(estimate_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, e, n, out, (int)refine, *prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_hll_count(const uint8_t *regs, int64_t n, const ss_hll_params *prm, float *out, int64_t out_stride,
                            void *stream)
{
    using namespace ss;
    if (n < 0) return SS_ERR_INVALID_ARG;
    const int rc = check_params(prm);
    if (rc != SS_OK) return rc;
    if (n == 0) return SS_OK;
    if (!regs || !out || out_stride < 1) return SS_ERR_INVALID_ARG;
    const int M = 1 << prm->p;
    const int CH = M >> 4;
    const int SG = CH > kWave ? kWave : CH;
    const int rows_per_block = (256 / kWave) * (kWave / SG) * kCountRows;
    const int64_t blocks = (n + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(hll_count_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, regs, n, M, out, out_stride,
                       *prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
