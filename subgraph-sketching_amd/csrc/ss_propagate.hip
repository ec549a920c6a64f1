// ss_propagate.hip -- one hop of sketch propagation (CSR pull) + fused HLL++ cardinality.
//
// Replaces MinhashPropagation / HllPropagation (reference hashing.py:28-45) and the per-hop hll_count
// of build_hash_tables (hashing.py:160-163).  HBM-bound: per hop it streams (E' + N) sketch rows.
//
// Mapping: one wavefront per destination row.  A 16-byte chunk per lane; a MinHash row of P u32 is
// CM = P/4 chunks, an HLL row of M bytes is CH = M/16 chunks.  The wave is split into G = 64/SG
// sub-groups (SG = pow2 >= chunks per row, <= 64) that walk G neighbours concurrently
// (P=128: 2 neighbours per step, M=256: 4), so every global load is a coalesced dwordx4 and a step
// moves 1 KiB per wave.  Partial min / max are combined across sub-groups once per row with
// cross-lane shuffles.  The implicit self loop (i < n_self) is one extra virtual neighbour.
#include "ss_common.hpp"

namespace ss {

__device__ __forceinline__ u32x4 shfl_xor4(u32x4 v, int mask)
{
    u32x4 r;
    r.x = (uint32_t)__shfl_xor((int)v.x, mask);
    r.y = (uint32_t)__shfl_xor((int)v.y, mask);
    r.z = (uint32_t)__shfl_xor((int)v.z, mask);
    r.w = (uint32_t)__shfl_xor((int)v.w, mask);
    return r;
}

__host__ __device__ constexpr int pow2_ceil(int x)
{
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// TP / TM > 0: compile-time row sizes (fast path); 0: run-time.
template <int TP, int TM>
__global__ __launch_bounds__(256) void propagate_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        int64_t N, int64_t n_self_arg, const int64_t *__restrict__ n_self_dev,
                                                        const uint32_t *__restrict__ mh_in, uint32_t *__restrict__ mh_out, int P_rt,
                                                        const uint8_t *__restrict__ hll_in, uint8_t *__restrict__ hll_out, int M_rt,
                                                        float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm)
{
    __shared__ EstimatorLds lds;
    const bool want_cards = cards_out != nullptr && hll_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);

    const int P = TP ? TP : P_rt;
    const int M = TM ? TM : M_rt;
    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform row id (readfirstlane makes the uniformity visible to the compiler: scalar loads, scalar loop control)
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    if (i >= N) return;

    const int64_t rb = rowptr[i];
    const int deg = (int)(rowptr[i + 1] - rb);
    const int64_t n_self = n_self_dev ? *n_self_dev : n_self_arg;
    const int self = i < n_self ? 1 : 0;
    const int total = deg + self;
    const int32_t *nb = col + rb;

    // ---------------- MinHash: min over neighbours ----------------
    if (mh_out) {
        const int CM = P >> 2;
        const int SG = TP ? (pow2_ceil(TP >> 2) > kWave ? kWave : pow2_ceil(TP >> 2)) : (pow2_ceil(CM) > kWave ? kWave : pow2_ceil(CM));
        const int G = kWave / SG;
        const int g = lane / SG, cl = lane % SG;
        for (int cb = 0; cb < CM; cb += SG) {
            const int c = cb + cl;
            const bool act = c < CM;
            u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (act) {
#pragma unroll 4
                for (int t = g; t < total; t += G) {
                    const int64_t j = t < deg ? (int64_t)nb[t] : i;
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(mh_in + j * P + 4 * c);
                    acc = min4(acc, x);
                }
            }
            for (int off = SG; off < kWave; off <<= 1) acc = min4(acc, shfl_xor4(acc, off));
            if (total == 0) acc = u32x4{0u, 0u, 0u, 0u};
            if (act && g == 0) *reinterpret_cast<u32x4 *>(mh_out + i * P + 4 * c) = acc;
        }
    }

    // ---------------- HLL: byte-wise max over neighbours, fused cardinality ----------------
    if (hll_out) {
        const int CH = M >> 4;
        const int SG = TM ? (pow2_ceil(TM >> 4) > kWave ? kWave : pow2_ceil(TM >> 4)) : (pow2_ceil(CH) > kWave ? kWave : pow2_ceil(CH));
        const int G = kWave / SG;
        const int g = lane / SG, cl = lane % SG;
        int zeros = 0;
        float hsum = 0.0f;
        for (int cb = 0; cb < CH; cb += SG) {
            const int c = cb + cl;
            const bool act = c < CH;
            // accumulate even / odd bytes separately as packed u16 (see ss_common.hpp)
            u32x4 ae = {0u, 0u, 0u, 0u}, ao = {0u, 0u, 0u, 0u};
            if (act) {
#pragma unroll 4
                for (int t = g; t < total; t += G) {
                    const int64_t j = t < deg ? (int64_t)nb[t] : i;
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(hll_in + j * M + 16 * c);
                    ae.x = pk_max_u16(ae.x, x.x & 0x00FF00FFu); ao.x = pk_max_u16(ao.x, x.x & 0xFF00FF00u);
                    ae.y = pk_max_u16(ae.y, x.y & 0x00FF00FFu); ao.y = pk_max_u16(ao.y, x.y & 0xFF00FF00u);
                    ae.z = pk_max_u16(ae.z, x.z & 0x00FF00FFu); ao.z = pk_max_u16(ao.z, x.z & 0xFF00FF00u);
                    ae.w = pk_max_u16(ae.w, x.w & 0x00FF00FFu); ao.w = pk_max_u16(ao.w, x.w & 0xFF00FF00u);
                }
            }
            u32x4 acc = {ae.x | ao.x, ae.y | ao.y, ae.z | ao.z, ae.w | ao.w};
            for (int off = SG; off < kWave; off <<= 1) acc = bytemax16(acc, shfl_xor4(acc, off));
            if (act && g == 0) {
                *reinterpret_cast<u32x4 *>(hll_out + i * M + 16 * c) = acc;
                if (want_cards) {
                    hll_dword_stats(acc.x, zeros, hsum);
                    hll_dword_stats(acc.y, zeros, hsum);
                    hll_dword_stats(acc.z, zeros, hsum);
                    hll_dword_stats(acc.w, zeros, hsum);
                }
            }
        }
        if (want_cards) {
            // lanes of sub-group 0 hold partial stats; the rest hold 0
            for (int off = 1; off < kWave; off <<= 1) {
                zeros += __shfl_xor(zeros, off);
                hsum += __shfl_xor(hsum, off);
            }
            if (lane == 0) cards_out[i * cards_stride] = hll_estimate(est, zeros, hsum);
        }
    }
}

template <int TP, int TM>
int launch_propagate(const int64_t *rowptr, const int32_t *col, int64_t N, int64_t n_self, const int64_t *n_self_dev,
                     const uint32_t *mh_in,
                     uint32_t *mh_out, int P, const uint8_t *hll_in, uint8_t *hll_out, int M, float *cards_out,
                     int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    const int rows_per_block = 256 / kWave;
    const int64_t blocks = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((propagate_kernel<TP, TM>), dim3((unsigned)blocks), dim3(256), 0, stream, rowptr, col, N, n_self, n_self_dev,
                       mh_in, mh_out, P, hll_in, hll_out, M, cards_out, cards_stride, prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

}  // namespace ss

extern "C" int ss_propagate(const int64_t *rowptr, const int32_t *col, int64_t N, int64_t n_self_loops,
                            const int64_t *n_self_loops_dev, const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                            const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                            float *cards_out, int64_t cards_stride, const ss_hll_params *prm, void *stream)
{
    using namespace ss;
    if (N < 0 || !rowptr) return SS_ERR_INVALID_ARG;
    if (N == 0) return SS_OK;
    if ((mh_in == nullptr) != (mh_out == nullptr) || (hll_in == nullptr) != (hll_out == nullptr)) return SS_ERR_INVALID_ARG;
    if (!mh_out && !hll_out) return SS_ERR_INVALID_ARG;
    if (mh_out && (P <= 0 || (P & 3) || P > 2048)) return SS_ERR_INVALID_ARG;
    if (hll_out && (M < 16 || (M & (M - 1)) || M > 65536)) return SS_ERR_INVALID_ARG;
    if (N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    ss_hll_params p0 = {};
    if (cards_out) {
        if (!hll_out) return SS_ERR_INVALID_ARG;
        const int rc = check_params(prm);
        if (rc != SS_OK) return rc;
        if ((1 << prm->p) != M) return SS_ERR_INVALID_ARG;
        p0 = *prm;
    }
    const bool fast = (!mh_out || P == 128) && (!hll_out || M == 256);
    if (fast)
        return launch_propagate<128, 256>(rowptr, col, N, n_self_loops, n_self_loops_dev, mh_in, mh_out, 128, hll_in, hll_out, 256, cards_out,
                                          cards_stride, p0, (hipStream_t)stream);
    return launch_propagate<0, 0>(rowptr, col, N, n_self_loops, n_self_loops_dev, mh_in, mh_out, P, hll_in, hll_out, M, cards_out, cards_stride, p0,
                                  (hipStream_t)stream);
}
