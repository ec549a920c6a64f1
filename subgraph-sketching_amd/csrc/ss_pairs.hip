// ss_pairs.hip -- per-node-pair subgraph features (the query side of the hot path).
//
// Replaces ElphHashes._get_intersections / jaccard / _hll_merge / hll_count / get_subgraph_features
// for one chunk of pairs (reference hashing.py:167-189, 234-237, 247-256, 258-323).
//
// HBM-bound gather: a pair (u, v) touches 2h sketch rows of 4P + M bytes (768 B at P=128, p=8) exactly
// once -- the reference gathers every row h times per call (hashing.py:180-183).
//
// Mapping: one 16-lane DPP row per pair, 4 pairs per wavefront, 16 per 256-thread block.  Lane l of a
// row owns 16-byte chunks l, l+16, ... of each sketch row (P=128, M=256: two MinHash chunks and one HLL
// chunk per row, all dwordx4 loads, 2h rows in flight per pair = 12 KiB per wave at h=2).  Per (k1,k2):
// MinHash equality count, byte-wise HLL union max, zero-register count and harmonic sum are reduced
// inside the row with DPP (quad_perm / row_half_mirror / row_mirror -- no LDS traffic), then lane c < h^2
// runs the HLL++ estimator for combination c and the h(h+2) features are assembled in fp32 in the
// reference's own operation order.
#include <cstdlib>
#include <cstring>

#include "ss_common.hpp"

namespace ss {

constexpr int kPairGrid = 256 * 32;  // most workgroups a launch uses; each loops over its share of the pairs

struct PairTables {
    const uint32_t *mh[SS_MAX_HOPS];
    const uint8_t *hll[SS_MAX_HOPS];
};

// (measured and rejected in round 3: counting equal dwords as 4 - sum(min(a ^ b, 1)) instead of v_cmp_eq_u32 + v_addc_co_u32 -- hipcc
// puts an `s_nop 1` behind each of the 72 compares of a pair at h = 3, gfx950 wanting two wait states between a VALU write of VCC
// and its VALU read -- removes 59 of 102 s_nops but adds 47 VALU instructions: level on cache-resident tables, 3.5 % SLOWER on
// citation2-size tables (3 713 against 3 580 us for 4 M pairs): other wavefronts fill the wait states, nobody fills extra instructions)
__device__ __forceinline__ int eq4(u32x4 a, u32x4 b)
{
    return (int)(a.x == b.x) + (int)(a.y == b.y) + (int)(a.z == b.z) + (int)(a.w == b.w);
}

// generic path: union of two 16-register chunks -> non-zero count and harmonic sum
__device__ __forceinline__ void union_stats(u32x4 a, u32x4 b, int &nonzero, float &hsum)
{
    const u32x4 m = bytemax16(a, b);
    hll_dword_stats(m.x, nonzero, hsum);
    hll_dword_stats(m.y, nonzero, hsum);
    hll_dword_stats(m.z, nonzero, hsum);
    hll_dword_stats(m.w, nonzero, hsum);
}

// fast path: a 16-register HLL chunk pre-digested once per row so that each of the h^2 unions costs
// 4 instructions per dword: bf16 patterns of 2^-r (even / odd registers; max of registers == unsigned min of
// patterns) and a 16-bit "register is zero" mask (union register zero <=> zero in both rows).
struct HllChunk {
    uint32_t pe[4], po[4];
    uint32_t zero_mask;
};

__device__ __forceinline__ HllChunk digest_chunk(u32x4 x)
{
    HllChunk c;
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
    uint32_t nzbits = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        c.pe[d] = regs_even_to_bf16(w[d]);
        c.po[d] = regs_odd_to_bf16(w[d]);
        nzbits |= nonzero_byte_flags(w[d]) >> (7 - d);  // flags live in bits 7,15,23,31 -> bits d, 8+d, 16+d, 24+d
    }
    c.zero_mask = ~nzbits & 0x0F0F0F0Fu;
    return c;
}

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
}

__device__ __forceinline__ void union_stats_digested(const HllChunk &a, const HllChunk &b, int &zeros, float &hsum)
{
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        hsum = dot2_ones(pk_min_u16(a.pe[d], b.pe[d]), hsum);
        hsum = dot2_ones(pk_min_u16(a.po[d], b.po[d]), hsum);
    }
    zeros += __builtin_popcount(a.zero_mask & b.zero_mask);
}

// feature algebra of get_subgraph_features (hashing.py:276-320); I is indexed [k1-1][k2-1].
template <int H>
__device__ __forceinline__ void assemble_features(const float (&I)[H][H], const float (&c1)[H], const float (&c2)[H],
                                                  uint32_t flags, float (&f)[H * (H + 2)])
{
    f[0] = I[0][0];
    if constexpr (H == 1) {
        f[1] = c2[0] - f[0];
        f[2] = c1[0] - f[0];
    } else if constexpr (H == 2) {
        f[1] = I[1][0] - f[0];
        f[2] = I[0][1] - f[0];
        f[3] = I[1][1] - f[0] - f[1] - f[2];
        f[4] = c2[0] - (f[0] + f[1]);
        f[5] = c1[0] - f[0] - f[2];
        f[6] = c2[1] - ((((f[0] + f[4]) + f[1]) + f[2]) + f[3]);  /* torch.sum order over 5 strided floats, see note */
        f[7] = c1[1] - f[0] - (((f[0] + f[1]) + f[2]) + f[3]) - f[5];  // f0 twice, as the reference (:287)
    } else {
        f[1] = I[1][0] - f[0];
        f[2] = I[0][1] - f[0];
        f[3] = I[1][1] - f[0] - f[1] - f[2];
        f[4] = I[2][0] - f[0] - f[1];
        f[5] = I[0][2] - f[0] - f[2];
        const float s04 = ((f[0] + f[1]) + f[2]) + f[3];
        f[6] = I[2][1] - s04 - f[4];
        f[7] = I[1][2] - s04 - f[5];
        f[8] = I[2][2] - (((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]);
        f[9] = c2[0] - f[0] - f[1] - f[4];
        f[10] = c1[0] - f[0] - f[2] - f[5];
        const float s05 = (((f[0] + f[4]) + f[1]) + f[2]) + f[3];
        f[11] = c2[1] - s05 - f[6] - f[9];
        f[12] = c1[1] - s05 - f[7] - f[10];
        const float s09 = (((((((f[8] + f[0]) + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7];
        f[13] = c2[2] - s09 - f[9] - f[11];
        f[14] = c1[2] - s09 - f[10] - f[12];
    }
    if (!(flags & SS_FLAG_USE_ZERO_ONE)) {
        if constexpr (H == 2) { f[4] = 0.0f; f[5] = 0.0f; }
        if constexpr (H == 3) { f[4] = 0.0f; f[5] = 0.0f; f[11] = 0.0f; f[12] = 0.0f; }
    }
    if (flags & SS_FLAG_FLOOR_SF) {
#pragma unroll
        for (int k = 0; k < H * (H + 2); ++k) f[k] = f[k] < 0.0f ? 0.0f : f[k];
    }
}

// TP, TM > 0: compile-time sketch sizes, all 2H rows register-resident (fast path).
// TP = TM = 0: run-time sizes, rows re-read per (k1,k2) (parameter sweeps / tests; not tuned).
// OCC (walks with locality, ss_pair_features_grouped: links grouped by their first node or listed that way): left alone the register
// allocator takes 117 (h = 2) / 151 (h = 3) VGPRs at P = 128 -- four / three wavefronts per SIMD.  Held to five / four (96 / 128 VGPRs,
// 20 / 12 bytes of scratch) the grouped walks are 5-7 % faster (collab-size BUDDY precompute 809-813 -> 771 us, citation2-size
// evaluation lists 4 267-4 297 -> 3 962 us per 8 M links: the first node's rows come from the L2, a fifth wavefront hides more of what
// is left); on uniformly random pairs the same budget is level to 9 % slower (HBM-resident tables, B = 262 144: 143 -> 157 us), so
// ss_pair_features keeps the default budget; six / five wavefronts spill in earnest (+25 %).
template <int H, int TP, int TM, bool OCC = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC ? (H == 2 ? 5 : H == 3 ? 4 : 1) : 1)))
void pair_features_kernel(const int64_t *__restrict__ links, int64_t B, int64_t N, PairTables tabs,
                                                            int P_rt, int M_rt, const float *__restrict__ cards, int64_t cards_stride,
                                                            ss_hll_params prm, uint32_t flags, float *__restrict__ out,
                                                            int32_t *__restrict__ dbg_match, int32_t *__restrict__ dbg_zero,
                                                            float *__restrict__ dbg_inter, int32_t *__restrict__ err,
                                                            const float *__restrict__ degrees, const int32_t *__restrict__ order)
{
    // order (nullable): position t of the walk is pair order[t] (ss_pair_features_grouped: pairs with the same first node are
    // consecutive positions, i.e. neighbouring lane groups of one workgroup -- their rows of u meet in the CU's L1 / the L2)
    __shared__ EstimatorLds lds;
    // the estimator tables are staged into LDS AFTER the first pair's sketch rows have been requested (first loop iteration
    // below): their 2.6 KB come out of the L2 while the 12 KiB of rows per wavefront travel, instead of 2 us before them
    EstimatorTables est = {};
    bool staged = false;

    constexpr int NF = H * (H + 2);
    constexpr int NC = H * H;
    const int P = TP ? TP : P_rt;
    const int M = TM ? TM : M_rt;
    const int l = threadIdx.x & (kRow - 1);
    // persistent workgroups: the estimator tables are staged once, every 16-lane group then takes pairs q, q + stride, ...;
    // the node ids of the NEXT pair are requested before the sketch rows of the current one are consumed, so only one
    // dependent global round trip (ids -> rows) per pair is exposed instead of two
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x / kRow);
    int64_t q_raw = (int64_t)blockIdx.x * (blockDim.x / kRow) + threadIdx.x / kRow;
    int64_t u_next = 0, v_next = 0;
    int64_t q_cur = 0, q_next = 0;  // pair index of this / the next position (positions themselves without an order)
    if (q_raw < B) {
        q_cur = order ? (int64_t)order[q_raw] : q_raw;
        u_next = links[2 * q_cur];
        v_next = links[2 * q_cur + 1];
    }
    if (q_raw + stride < B) q_next = order ? (int64_t)order[q_raw + stride] : q_raw + stride;
    // the trip count is WORKGROUP-UNIFORM (q_raw - group index is the same for every thread): stage_tables(), which contains a
    // __syncthreads(), runs inside the first iteration and relies on exactly that
    for (; q_raw - (threadIdx.x / kRow) < B; q_raw += stride) {
    const bool q_ok = q_raw < B;
    const int64_t q = q_ok ? q_cur : B - 1;
    int64_t u = q_ok ? u_next : 0, v = q_ok ? v_next : 0;
    u = u < 0 ? u + N : u;  // torch-style negative indexing
    v = v < 0 ? v + N : v;
    const bool bad = (uint64_t)u >= (uint64_t)N || (uint64_t)v >= (uint64_t)N;
    if (bad) { u = 0; v = 0; }

    int mz[NC];     // (match << 20) | zeros, row total
    float hs[NC];   // harmonic sum, row total

    if constexpr (TP > 0) {
        constexpr int CMPL = TP / 4 / kRow;   // MinHash chunks per lane
        constexpr int CHPL = TM / 16 / kRow;  // HLL chunks per lane
        static_assert(CMPL >= 1 && CHPL >= 1 && (TP / 4) % kRow == 0 && (TM / 16) % kRow == 0, "fast path shape");
        u32x4 mu[H][CMPL], mv[H][CMPL], xu[H][CHPL], xv[H][CHPL];
#pragma unroll
        for (int k = 0; k < H; ++k) {
#pragma unroll
            for (int c = 0; c < CMPL; ++c) {
                mu[k][c] = *reinterpret_cast<const u32x4 *>(tabs.mh[k] + u * TP + 4 * (l + kRow * c));
                mv[k][c] = *reinterpret_cast<const u32x4 *>(tabs.mh[k] + v * TP + 4 * (l + kRow * c));
            }
#pragma unroll
            for (int c = 0; c < CHPL; ++c) {
                xu[k][c] = *reinterpret_cast<const u32x4 *>(tabs.hll[k] + u * TM + 16 * (l + kRow * c));
                xv[k][c] = *reinterpret_cast<const u32x4 *>(tabs.hll[k] + v * TM + 16 * (l + kRow * c));
            }
        }
        if (q_raw + stride < B) {  // ids of this group's next pair (returns after the rows above: vmcnt is in order)
            u_next = links[2 * q_next];
            v_next = links[2 * q_next + 1];
        }
        q_cur = q_next;
        if (q_raw + 2 * stride < B) q_next = order ? (int64_t)order[q_raw + 2 * stride] : q_raw + 2 * stride;  // one position further ahead
        if (!staged) {  // workgroup-uniform (the trip count is): the barrier inside is reached by every thread
            est = stage_tables(lds, prm);
            staged = true;
        }
        // MinHash first: the equality counts consume (and free) the 2H MinHash rows before the HLL rows are expanded into their
        // digests (120 -> 113 VGPRs at H = 2, 166 -> 147 at H = 3)
        int match[NC];
#pragma unroll
        for (int k1 = 0; k1 < H; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) {
                int m = 0;
#pragma unroll
                for (int c = 0; c < CMPL; ++c) m += eq4(mu[k1][c], mv[k2][c]);
                match[k1 * H + k2] = m;
            }
        HllChunk hu[H][CHPL], hv[H][CHPL];
#pragma unroll
        for (int k = 0; k < H; ++k)
#pragma unroll
            for (int c = 0; c < CHPL; ++c) {
                hu[k][c] = digest_chunk(xu[k][c]);
                hv[k][c] = digest_chunk(xv[k][c]);
            }
#pragma unroll
        for (int k1 = 0; k1 < H; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) {
                int zeros = 0;
                float hsum = 0.0f;
#pragma unroll
                for (int c = 0; c < CHPL; ++c) union_stats_digested(hu[k1][c], hv[k2][c], zeros, hsum);
                mz[k1 * H + k2] = row16_sum_i((match[k1 * H + k2] << 20) | zeros);
                hs[k1 * H + k2] = row16_sum_f(hsum);
            }
    } else {
        if (q_raw + stride < B) {
            u_next = links[2 * q_next];
            v_next = links[2 * q_next + 1];
        }
        q_cur = q_next;
        if (q_raw + 2 * stride < B) q_next = order ? (int64_t)order[q_raw + 2 * stride] : q_raw + 2 * stride;
        if (!staged) {
            est = stage_tables(lds, prm);
            staged = true;
        }
        const int CM = P >> 2, CH = M >> 4;
#pragma unroll
        for (int k1 = 0; k1 < H; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) {
                int match = 0, nonzero = 0;
                float hsum = 0.0f;
                for (int c = l; c < CM; c += kRow)
                    match += eq4(*reinterpret_cast<const u32x4 *>(tabs.mh[k1] + u * P + 4 * c),
                                 *reinterpret_cast<const u32x4 *>(tabs.mh[k2] + v * P + 4 * c));
                int chunks = 0;
                for (int c = l; c < CH; c += kRow, ++chunks)
                    union_stats(*reinterpret_cast<const u32x4 *>(tabs.hll[k1] + u * M + 16 * c),
                                *reinterpret_cast<const u32x4 *>(tabs.hll[k2] + v * M + 16 * c), nonzero, hsum);
                mz[k1 * H + k2] = row16_sum_i((match << 20) | (16 * chunks - nonzero));
                hs[k1 * H + k2] = row16_sum_f(hsum);
            }
    }

    // lane c < H^2 finishes combination c: I = (match / P) * hll_count(union)   (hashing.py:184-187)
    int my_mz = mz[0];
    float my_hs = hs[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) {
        my_mz = (l == c) ? mz[c] : my_mz;
        my_hs = (l == c) ? hs[c] : my_hs;
    }
    float my_I = 0.0f;
    const int my_match = (int)((uint32_t)my_mz >> 20), my_zeros = my_mz & 0xFFFFF;  // P <= 2048: the packed word uses all 32 bits
    if (l < NC) {
        const float jac = (float)my_match / (float)P;
        my_I = jac * hll_estimate(est, my_zeros, my_hs);
    }
    const int row_base = (threadIdx.x & (kWave - 1)) & ~(kRow - 1);
    float I[H][H];
#pragma unroll
    for (int c = 0; c < NC; ++c) I[c / H][c % H] = __shfl(my_I, row_base + c);

    float c1[H], c2[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        c1[k] = cards[u * cards_stride + k];
        c2[k] = cards[v * cards_stride + k];
    }
    float f[NF];
    assemble_features<H>(I, c1, c2, flags, f);

    float my_f = f[0];
#pragma unroll
    for (int k = 1; k < NF; ++k) my_f = (l == k) ? f[k] : my_f;
    if (bad) my_f = __uint_as_float(0x7FC00000u);
    if (q_ok && degrees) {
        // fused BUDDY._append_degree_normalised (reference models/elph.py:276-293): rows become [f, f / sqrt(d_u * d_v)]
        // with NaN / Inf (zero-degree nodes) replaced by 0
        const float normaliser = sqrtf(degrees[u] * degrees[v]);
        float normed = my_f / normaliser;
        if (isnan(normed) || isinf(normed)) normed = 0.0f;
        if (bad) normed = __uint_as_float(0x7FC00000u);
        if (l < NF) {
            out[q * (2 * NF) + l] = my_f;
            out[q * (2 * NF) + NF + l] = normed;
        }
    }
    if (q_ok) {
        if (l < NF && !degrees) out[q * NF + l] = my_f;
        if (l < NC) {
            if (dbg_match) dbg_match[q * NC + l] = my_match;
            if (dbg_zero) dbg_zero[q * NC + l] = my_zeros;
            if (dbg_inter) dbg_inter[q * NC + l] = my_I;
        }
        if (bad && l == 0 && err) *err = 1;
    }
    }  // pairs of this lane group
}

// ---- run-aware query: consecutive pairs that share their first node reuse its rows ------------------------------------------
// BUDDY's link sets repeat every source node many times (ogbl-citation2's evaluation set lists 1 000 negatives per source; any
// coalesced edge list is ordered by source; ss_group_links_by_source puts an arbitrary link set into that form): half of the bytes
// of a pair are rows of u that the previous pair has just read.  Here a 16-lane group takes a CONTIGUOUS chunk of K <= 16 pairs
// (positions t of `order` when given, else pairs themselves), lane l fetches the ids of the chunk's l-th pair with one coalesced
// load, and the group walks the chunk keeping u's MinHash rows, HLL digests and cardinalities in registers while u does not
// change: such a pair costs H * R + 8 + 4H + 4H(H+2) bytes instead of 2H * R + ... (2 412 instead of 4 708 at H = 3).  A pair's
// features depend on its own rows only, so rows are bit-identical to pair_features_kernel's whatever the order or the chunking.
// CAP: ask the register allocator for 4 (H <= 2) / 3 (H = 3) wavefronts per SIMD (some spills) instead of 3 / 2
template <int H, int TP, int TM, bool CAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CAP ? (H == 3 ? 3 : 4) : 1))) void pair_features_runs_kernel(const int64_t *__restrict__ links, const int32_t *__restrict__ order,
                                                                 int64_t B, int64_t N, int K, PairTables tabs,
                                                                 const float *__restrict__ cards, int64_t cards_stride, ss_hll_params prm,
                                                                 uint32_t flags, float *__restrict__ out, int32_t *__restrict__ err,
                                                                 const float *__restrict__ degrees)
{
    __shared__ EstimatorLds lds;
    EstimatorTables est = {};
    bool staged = false;
    constexpr int NF = H * (H + 2);
    constexpr int NC = H * H;
    constexpr int CMPL = TP / 4 / kRow;   // MinHash chunks per lane
    constexpr int CHPL = TM / 16 / kRow;  // HLL chunks per lane
    static_assert(CMPL >= 1 && CHPL >= 1 && (TP / 4) % kRow == 0 && (TM / 16) % kRow == 0, "fast path shape");
    const int l = threadIdx.x & (kRow - 1);
    const int row_base = (threadIdx.x & (kWave - 1)) & ~(kRow - 1);
    const int64_t n_groups = (int64_t)gridDim.x * (blockDim.x / kRow);
    const int64_t n_chunks = (B + K - 1) / K;
    const float nan = __uint_as_float(0x7FC00000u);
    // workgroup-uniform trip count (stage_tables() below contains a barrier): every group of the workgroup runs the same number
    // of outer iterations, groups past the end walk an empty chunk
    for (int64_t c0 = (int64_t)blockIdx.x * (blockDim.x / kRow); c0 < n_chunks; c0 += n_groups) {
        const int64_t chunk = c0 + threadIdx.x / kRow;
        const int64_t base = chunk * K;
        const int cnt = chunk < n_chunks ? (int)(B - base < K ? B - base : K) : 0;
        // lane l: position, ids and validity of the chunk's l-th pair
        int64_t q_l = 0;
        int u_l = -1, v_l = -1;
        if (l < cnt) {
            q_l = order ? (int64_t)order[base + l] : base + l;
            int64_t u = links[2 * q_l], v = links[2 * q_l + 1];
            u = u < 0 ? u + N : u;  // torch-style negative indexing
            v = v < 0 ? v + N : v;
            if ((uint64_t)u < (uint64_t)N && (uint64_t)v < (uint64_t)N) {
                u_l = (int)u;
                v_l = (int)v;
            } else if (err) {
                *err = 1;
            }
        }
        int prev_u = -1;
        // what stays in registers while u does not change: its MinHash rows, its RAW HLL rows (4 registers per chunk; their
        // digests -- 9 -- are remade for every pair: ~20 instructions per chunk against a third wavefront per SIMD) and its cards
        u32x4 mu[H][CMPL], xu[H][CHPL];
        float c1[H];
        for (int t = 0; t < K; ++t) {  // workgroup-uniform
            const int u = __shfl(u_l, row_base + t), v = __shfl(v_l, row_base + t);
            const int q_lo = __shfl((int)q_l, row_base + t), q_hi = __shfl((int)(q_l >> 32), row_base + t);
            const int64_t q = ((int64_t)q_hi << 32) | (uint32_t)q_lo;
            const bool live = t < cnt;      // group-uniform
            const bool bad = u < 0;          // group-uniform
            const bool fresh = live && !bad && u != prev_u;
            const int64_t ur = bad ? 0 : u, vr = bad ? 0 : v;  // (a bad pair's rows are requested -- node 0's -- and never used)
            u32x4 mv[H][CMPL], xv[H][CHPL];
            if (fresh) {
#pragma unroll
                for (int k = 0; k < H; ++k) {
#pragma unroll
                    for (int c = 0; c < CMPL; ++c) mu[k][c] = *reinterpret_cast<const u32x4 *>(tabs.mh[k] + ur * TP + 4 * (l + kRow * c));
#pragma unroll
                    for (int c = 0; c < CHPL; ++c) xu[k][c] = *reinterpret_cast<const u32x4 *>(tabs.hll[k] + ur * TM + 16 * (l + kRow * c));
                }
            }
            if (live) {
#pragma unroll
                for (int k = 0; k < H; ++k) {
#pragma unroll
                    for (int c = 0; c < CMPL; ++c) mv[k][c] = *reinterpret_cast<const u32x4 *>(tabs.mh[k] + vr * TP + 4 * (l + kRow * c));
#pragma unroll
                    for (int c = 0; c < CHPL; ++c) xv[k][c] = *reinterpret_cast<const u32x4 *>(tabs.hll[k] + vr * TM + 16 * (l + kRow * c));
                }
            }
            if (!staged) {  // first pair of the workgroup: the tables travel under the rows requested above
                est = stage_tables(lds, prm);
                staged = true;
            }
            if (!live) continue;  // (no barrier below)
            if (bad) {             // ids out of range: a NaN row, reported through err; u's rows stay valid for the next pair
                if (degrees) {
                    if (l < 2 * NF) out[q * (2 * NF) + l] = nan;
                    if (l + kRow < 2 * NF) out[q * (2 * NF) + l + kRow] = nan;
                } else if (l < NF) {
                    out[q * NF + l] = nan;
                }
                continue;
            }
            float c2[H];
#pragma unroll
            for (int k = 0; k < H; ++k) {
                if (fresh) c1[k] = cards[ur * cards_stride + k];
                c2[k] = cards[vr * cards_stride + k];
            }
            if (fresh) prev_u = u;
            int mz[NC];
            float hs[NC];
            {
                int match[NC];
#pragma unroll
                for (int k1 = 0; k1 < H; ++k1)
#pragma unroll
                    for (int k2 = 0; k2 < H; ++k2) {
                        int m = 0;
#pragma unroll
                        for (int c = 0; c < CMPL; ++c) m += eq4(mu[k1][c], mv[k2][c]);
                        match[k1 * H + k2] = m;
                    }
                HllChunk hv[H][CHPL];
#pragma unroll
                for (int k = 0; k < H; ++k)
#pragma unroll
                    for (int c = 0; c < CHPL; ++c) hv[k][c] = digest_chunk(xv[k][c]);
#pragma unroll
                for (int k1 = 0; k1 < H; ++k1) {
                    HllChunk hu[CHPL];  // one row of u at a time: its digest lives for H unions only
#pragma unroll
                    for (int c = 0; c < CHPL; ++c) hu[c] = digest_chunk(xu[k1][c]);
#pragma unroll
                    for (int k2 = 0; k2 < H; ++k2) {
                        int zeros = 0;
                        float hsum = 0.0f;
#pragma unroll
                        for (int c = 0; c < CHPL; ++c) union_stats_digested(hu[c], hv[k2][c], zeros, hsum);
                        mz[k1 * H + k2] = row16_sum_i((match[k1 * H + k2] << 20) | zeros);
                        hs[k1 * H + k2] = row16_sum_f(hsum);
                    }
                }
            }
            // lane c < H^2 finishes combination c, exactly as pair_features_kernel does
            int my_mz = mz[0];
            float my_hs = hs[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) {
                my_mz = (l == c) ? mz[c] : my_mz;
                my_hs = (l == c) ? hs[c] : my_hs;
            }
            float my_I = 0.0f;
            if (l < NC) {
                const float jac = (float)(int)((uint32_t)my_mz >> 20) / (float)TP;
                my_I = jac * hll_estimate(est, my_mz & 0xFFFFF, my_hs);
            }
            float I[H][H];
#pragma unroll
            for (int c = 0; c < NC; ++c) I[c / H][c % H] = __shfl(my_I, row_base + c);
            float f[NF];
            assemble_features<H>(I, c1, c2, flags, f);
            float my_f = f[0];
#pragma unroll
            for (int k = 1; k < NF; ++k) my_f = (l == k) ? f[k] : my_f;
            if (degrees) {  // fused BUDDY._append_degree_normalised, as in pair_features_kernel
                const float normaliser = sqrtf(degrees[ur] * degrees[vr]);
                float normed = my_f / normaliser;
                if (isnan(normed) || isinf(normed)) normed = 0.0f;
                if (l < NF) {
                    out[q * (2 * NF) + l] = my_f;
                    out[q * (2 * NF) + NF + l] = normed;
                }
            } else if (l < NF) {
                out[q * NF + l] = my_f;
            }
        }
    }
}

template <int H, int TP, int TM>
int launch_pairs(const int64_t *links, int64_t B, int64_t N, const PairTables &tabs, int P, int M, const float *cards,
                 int64_t cards_stride, const ss_hll_params &prm, uint32_t flags, float *out, int32_t *dbg_match,
                 int32_t *dbg_zero, float *dbg_inter, int32_t *err, const float *degrees, hipStream_t stream, const int32_t *order = nullptr,
                 bool grouped = false)
{
    const int pairs_per_block = 256 / kRow;
    // pairs per 16-lane group the grid is sized for: ONE while that still fits the chip in a single round of workgroups (ELPH
    // batches: B = 2 048 takes 5.4 us instead of 8.2), TWO beyond (the second pair's ids travel under the first one's rows:
    // B = 65 536: 35.6 us with 2 048 workgroups, 39.0 with 4 096), at most kPairGrid workgroups (B = 4 M: 2.23 G pairs/s with
    // 8 192, 2.18 with 2 048, 2.06 with 1 280).  Measured and rejected: 64- / 128-thread workgroups, persistent
    // grids of 512 - 1 024 workgroups, and a software-pipelined variant that requests the next pair's rows in the middle of the
    // current pair's arithmetic (165 VGPRs: +1.5 % on batches of millions, -7 % at B = 65 536).
    // SS_PAIR_GRID / SS_PAIR_PER_GROUP: tuning hooks of tools/probe_pairs.py
    static const int grid_env = getenv("SS_PAIR_GRID") ? atoi(getenv("SS_PAIR_GRID")) : 0;
    static const int per_group_env = getenv("SS_PAIR_PER_GROUP") ? atoi(getenv("SS_PAIR_PER_GROUP")) : 0;
    const int per_group = per_group_env > 0 ? per_group_env : (B <= 1024 * pairs_per_block ? 1 : 2);
    const int64_t max_blocks = grid_env > 0 ? grid_env : kPairGrid;
    int64_t blocks = (B + per_group * pairs_per_block - 1) / (per_group * pairs_per_block);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    // Which register budget (TP = 128 only: one more instantiation per hop count).  Walks with locality: the capped build (OCC, see the
    // kernel).  As-listed random pairs, round 6 (tools/probe_pairs_variants.py, profiles/round6_pairs_variants.txt; tables of 0.2 - 6.4
    // GB, cold and after 2 000 launches): at H = 3 the fourth wavefront per SIMD wins at every table size while the launch is small --
    // B = 65 536: -4 .. -9 %, B = 261 424 (ogbl-citation2's batch): -1 .. -5 % -- and is level (+-0.6 %) at 4 M pairs; at H = 2 the fifth
    // wins 1 - 4 % at B = 65 536 (the bench step's query, ELPH batches) and LOSES 2 - 9 % at 261 424 on tables above ~1.5 GB.
    // SS_PAIR_OCC = 0 / 1 forces a build (probes).
    static const int occ_env = getenv("SS_PAIR_OCC") ? atoi(getenv("SS_PAIR_OCC")) : -1;
    const bool small_launch = H == 3 ? B <= ((int64_t)1 << 21) : B <= 65536;
    const bool occ = occ_env >= 0 ? occ_env != 0 : (grouped || small_launch);
    {
        ProfileSpan span(stream, SS_PROF_PAIRS, true);
        if (occ && TP == 128 && H >= 2)
            span.launch(pair_features_kernel<H, TP, TM, (TP == 128 && H >= 2)>, dim3((unsigned)blocks), dim3(256), links, B, N, tabs, P, M, cards,
                        cards_stride, prm, flags, out, dbg_match, dbg_zero, dbg_inter, err, degrees, order);
        else
            span.launch(pair_features_kernel<H, TP, TM>, dim3((unsigned)blocks), dim3(256), links, B, N, tabs, P, M, cards, cards_stride, prm,
                        flags, out, dbg_match, dbg_zero, dbg_inter, err, degrees, order);
    }
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <int H>
int dispatch_pairs(const int64_t *links, int64_t B, int64_t N, const PairTables &tabs, int P, int M, const float *cards,
                   int64_t cards_stride, const ss_hll_params &prm, uint32_t flags, float *out, int32_t *dbg_match,
                   int32_t *dbg_zero, float *dbg_inter, int32_t *err, const float *degrees, hipStream_t stream, const int32_t *order = nullptr,
                   bool grouped = false)
{
#define SS_PAIRS_FAST(TP)                                                                                                      \
    if (P == TP && M == 256)                                                                                                    \
        return launch_pairs<H, TP, 256>(links, B, N, tabs, P, M, cards, cards_stride, prm, flags, out, dbg_match, dbg_zero,     \
                                        dbg_inter, err, degrees, stream, order, grouped);
    SS_PAIRS_FAST(128)  // the reference's default shape
    SS_PAIRS_FAST(64)   // the other permutation counts the first hop is specialised for (ss_first_hop: P / 64 = 1 .. 4)
    SS_PAIRS_FAST(192)
    SS_PAIRS_FAST(256)
#undef SS_PAIRS_FAST
    return launch_pairs<H, 0, 0>(links, B, N, tabs, P, M, cards, cards_stride, prm, flags, out, dbg_match, dbg_zero, dbg_inter,
                                 err, degrees, stream, order, grouped);
}

template <int H, int TP>
int launch_pair_runs(const int64_t *links, const int32_t *order, int64_t B, int64_t N, const PairTables &tabs, const float *cards,
                     int64_t cards_stride, const ss_hll_params &prm, uint32_t flags, float *out, int32_t *err, const float *degrees,
                     hipStream_t stream)
{
    // pairs per chunk: 16 once that still leaves >= 32 768 chunks (eight rounds of 16-group workgroups over 256 CUs); fewer for
    // smaller batches so that the chip stays full.  SS_PAIR_RUN_CHUNK: tuning hook
    static const int k_env = getenv("SS_PAIR_RUN_CHUNK") ? atoi(getenv("SS_PAIR_RUN_CHUNK")) : 0;
    int64_t K = k_env > 0 && k_env <= kRow ? k_env : B / 32768;
    K = K < 1 ? 1 : (K > kRow ? kRow : K);
    const int64_t chunks = (B + K - 1) / K;
    int64_t blocks = (chunks + 256 / kRow - 1) / (256 / kRow);
    if (blocks > kPairGrid) blocks = kPairGrid;
    static const bool cap = !(getenv("SS_PAIR_RUN_CAP") && atoi(getenv("SS_PAIR_RUN_CAP")) == 0);
    {
        ProfileSpan span(stream, SS_PROF_PAIRS);
        if (cap)
            hipLaunchKernelGGL((pair_features_runs_kernel<H, TP, 256, true>), dim3((unsigned)blocks), dim3(256), 0, stream, links, order, B, N, (int)K,
                               tabs, cards, cards_stride, prm, flags, out, err, degrees);
        else
            hipLaunchKernelGGL((pair_features_runs_kernel<H, TP, 256, false>), dim3((unsigned)blocks), dim3(256), 0, stream, links, order, B, N, (int)K,
                               tabs, cards, cards_stride, prm, flags, out, err, degrees);
    }
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <int H>
int dispatch_pair_runs(const int64_t *links, const int32_t *order, int64_t B, int64_t N, const PairTables &tabs, int P, const float *cards,
                       int64_t cards_stride, const ss_hll_params &prm, uint32_t flags, float *out, int32_t *err, const float *degrees,
                       hipStream_t stream)
{
    switch (P) {
        case 64: return launch_pair_runs<H, 64>(links, order, B, N, tabs, cards, cards_stride, prm, flags, out, err, degrees, stream);
        case 128: return launch_pair_runs<H, 128>(links, order, B, N, tabs, cards, cards_stride, prm, flags, out, err, degrees, stream);
        case 192: return launch_pair_runs<H, 192>(links, order, B, N, tabs, cards, cards_stride, prm, flags, out, err, degrees, stream);
        default: return launch_pair_runs<H, 256>(links, order, B, N, tabs, cards, cards_stride, prm, flags, out, err, degrees, stream);
    }
}

}  // namespace ss

static int pair_features_impl(const int64_t *links, int64_t B, int64_t N, int32_t h, const uint32_t *const *mh, int32_t P,
                              const uint8_t *const *hll, const float *cards, int64_t cards_stride, const ss_hll_params *prm,
                              uint32_t flags, const float *degrees, float *out, int32_t *dbg_match, int32_t *dbg_zero,
                              float *dbg_inter, int32_t *err_flag, void *stream, const int32_t *order = nullptr, bool grouped = false)
{
    using namespace ss;
    if (h < 1 || h > SS_MAX_HOPS) return SS_ERR_UNSUPPORTED;  // hashing.py:54, 308-309
    if (B < 0 || N < 0) return SS_ERR_INVALID_ARG;
    const int rc = check_params(prm);
    if (rc != SS_OK) return rc;
    if (B == 0) return SS_OK;
    if (N == 0 || !links || !mh || !hll || !cards || !out || cards_stride < h) return SS_ERR_INVALID_ARG;
    if (P <= 0 || (P & 3) || P > 2048) return SS_ERR_INVALID_ARG;
    PairTables tabs = {};
    for (int k = 0; k < h; ++k) {
        if (!mh[k] || !hll[k]) return SS_ERR_INVALID_ARG;
        tabs.mh[k] = mh[k];
        tabs.hll[k] = hll[k];
    }
    const int M = 1 << prm->p;
    hipStream_t s = (hipStream_t)stream;
    switch (h) {
        case 1: return dispatch_pairs<1>(links, B, N, tabs, P, M, cards, cards_stride, *prm, flags, out, dbg_match, dbg_zero, dbg_inter, err_flag, degrees, s, order, grouped);
        case 2: return dispatch_pairs<2>(links, B, N, tabs, P, M, cards, cards_stride, *prm, flags, out, dbg_match, dbg_zero, dbg_inter, err_flag, degrees, s, order, grouped);
        default: return dispatch_pairs<3>(links, B, N, tabs, P, M, cards, cards_stride, *prm, flags, out, dbg_match, dbg_zero, dbg_inter, err_flag, degrees, s, order, grouped);
    }
}

extern "C" int ss_pair_features(const int64_t *links, int64_t B, int64_t N, int32_t h,
                                const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                                const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                                float *out, int32_t *dbg_match, int32_t *dbg_zero, float *dbg_inter, int32_t *err_flag,
                                void *stream)
{
    return pair_features_impl(links, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, nullptr, out, dbg_match, dbg_zero,
                              dbg_inter, err_flag, stream);
}

extern "C" int ss_pair_features_normalised(const int64_t *links, int64_t B, int64_t N, int32_t h,
                                           const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                                           const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                                           const float *degrees, float *out, int32_t *err_flag, void *stream)
{
    if (!degrees) return SS_ERR_INVALID_ARG;
    return pair_features_impl(links, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, degrees, out, nullptr, nullptr, nullptr,
                              err_flag, stream);
}

// The query over a link list whose pairs are walked in runs of equal first nodes (see pair_features_runs_kernel): `order`
// (nullable) = a permutation of the pair indices, e.g. from ss_group_links_by_source; without it the pairs are walked as they
// are listed (a coalesced edge list, or an evaluation set that lists all negatives of a source together, already has the
// runs).  Row q of `out` is pair q whatever the order; rows are bit-identical to ss_pair_features[_normalised]'s.  degrees
// nullable (non-null: the degree-normalised copy is appended, as ss_pair_features_normalised does).  Shapes without a
// specialised kernel (P not in {64, 128, 192, 256} or p != 8) take the ordinary path -- same rows, no reuse.
// ---- the two permutations around a grouped query over a link set too large for its caches -------------------------------------
// Walking `order` directly makes every position read links[order[t]] (16 B) and write out[order[t]] (<= 120 B) at random places
// of arrays of gigabytes (ogbl-citation2: 5.7 GB of links, 21 GB of features): gfx9 counts loads and stores in ONE in-order
// counter, so the next pair's rows wait behind the previous pair's scattered store and its address translation -- measured 0.69
// against 1.63 G pairs/s for the same kernel on a link set that fits the Infinity Cache.  These two kernels do the same
// permutations as dedicated streaming passes (every access independent, thousands in flight per CU) around a query over the
// gathered -- now contiguous -- chunk.
namespace ss {

__global__ __launch_bounds__(256) void gather_links_kernel(const int64_t *__restrict__ links, const int32_t *__restrict__ order, int64_t n,
                                                           int64_t *__restrict__ out)
{
    typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<i64x2 *>(out)[t] = reinterpret_cast<const i64x2 *>(links)[order[t]];
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float *__restrict__ rows, const int32_t *__restrict__ order, int64_t n, int width,
                                                           float *__restrict__ out)
{
    const int64_t total = n * width;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = e / width;
        out[(int64_t)order[t] * width + (e - t * width)] = rows[e];
    }
}

}  // namespace ss

// out_links[t] := links[order[t]] (int64 [n, 2]);  out[order[t], :] := rows[t, :] (float [n, width]).  See above.
extern "C" int ss_gather_links(const int64_t *links, const int32_t *order, int64_t n, int64_t *out_links, void *stream)
{
    if (n < 0 || (n > 0 && (!links || !order || !out_links))) return SS_ERR_INVALID_ARG;
    if (n == 0) return SS_OK;
    // the kernel moves a pair as ONE 16-byte vector: both arrays must be 16-byte aligned (rows of a torch int64 [n, 2] tensor are)
    if ((reinterpret_cast<uintptr_t>(links) | reinterpret_cast<uintptr_t>(out_links)) & 15) return SS_ERR_INVALID_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(ss::gather_links_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, links, order, n, out_links);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_scatter_feature_rows(const float *rows, const int32_t *order, int64_t n, int32_t width, float *out, void *stream)
{
    if (n < 0 || width <= 0 || (n > 0 && (!rows || !order || !out))) return SS_ERR_INVALID_ARG;
    if (n == 0) return SS_OK;
    int64_t blocks = (n * width + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(ss::scatter_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, order, n, (int)width, out);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

static int pair_features_grouped_impl(int which /* -1: chosen per hop count, 0: ordinary kernel, 1: run-aware kernel */, const int64_t *links,
                                      const int32_t *order, int64_t B, int64_t N, int32_t h, const uint32_t *const *mh, int32_t P,
                                      const uint8_t *const *hll, const float *cards, int64_t cards_stride, const ss_hll_params *prm,
                                      uint32_t flags, const float *degrees, float *out, int32_t *err_flag, void *stream)
{
    using namespace ss;
    if (h < 1 || h > SS_MAX_HOPS) return SS_ERR_UNSUPPORTED;
    if (B < 0 || N < 0) return SS_ERR_INVALID_ARG;
    const int rc = check_params(prm);
    if (rc != SS_OK) return rc;
    if (B == 0) return SS_OK;
    if (N == 0 || N >= ((int64_t)1 << 31) || !links || !mh || !hll || !cards || !out || cards_stride < h) return SS_ERR_INVALID_ARG;
    if (P <= 0 || (P & 3) || P > 2048) return SS_ERR_INVALID_ARG;
    const bool fast = prm->p == 8 && (P == 64 || P == 128 || P == 192 || P == 256);
    // Which kernel (measured, tools/probe_pair_runs.py -> profiles/round3_pair_runs_*.json, 4 M links): WITH an order the ordinary
    // kernel -- neighbouring lane groups take neighbouring positions, so the rows of a shared first node meet in the CU's L1 / the
    // L2 -- is ahead or level at every hop count (ppa-size tables, h = 2, random links grouped: 2.69 against 2.63 G pairs/s;
    // citation2-size, h = 3: 1.63 against 1.51); WITHOUT one (runs as listed: a coalesced edge list, an evaluation set) the
    // run-aware kernel wins at h <= 2 (3.49 against 3.24 G pairs/s) and loses at h = 3, where its registers leave two wavefronts
    // per SIMD (1.73 against 1.93).  Round 6 (same probe, profiles/round6_pair_runs.txt): since the ordinary kernel runs the walks with
    // locality under its capped register budget (five / four wavefronts per SIMD) it also wins the lists walked AS LISTED at h = 2 --
    // collab-size tables 3.81 against 3.58 G pairs/s (sorted by source) and 4.05 against 3.89 (1 000 negatives per source), ppa size
    // 3.66 / 3.73 against 3.40 / 3.59, citation2 size 2.59 / 3.16 against 2.32 / 3.02 -- and at h = 3 (2.04-2.08 against 1.34); the
    // run-aware kernel keeps h = 1 (6.84 against 6.57 on evaluation lists).  A run-aware kernel with u's rows in LDS instead of registers
    // (global_load_lds landings, ds_read_b128 per comparison) was built and measured: 128 VGPRs + 124 bytes of scratch at four
    // wavefronts per SIMD (1.34 G pairs/s at h = 3), 170 VGPRs at three (1.93): level with the ordinary kernel's default budget, behind
    // its capped one -- not shipped.  SS_PAIR_GROUPED_KERNEL = runs | plain forces one.
    static const char *forced = getenv("SS_PAIR_GROUPED_KERNEL");
    const bool runs = fast && (which >= 0 ? which == 1 : (forced ? !strcmp(forced, "runs") : (order == nullptr && h <= 1)));
    if (!runs) {
        if (order && B >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
        return pair_features_impl(links, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, degrees, out, nullptr, nullptr, nullptr,
                                  err_flag, stream, order, /*grouped=*/true);
    }
    if (order && B >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;  // (order entries are int32 pair indices, as on the other path)
    PairTables tabs = {};
    for (int k = 0; k < h; ++k) {
        if (!mh[k] || !hll[k]) return SS_ERR_INVALID_ARG;
        tabs.mh[k] = mh[k];
        tabs.hll[k] = hll[k];
    }
    hipStream_t s = (hipStream_t)stream;
    switch (h) {
        case 1: return dispatch_pair_runs<1>(links, order, B, N, tabs, P, cards, cards_stride, *prm, flags, out, err_flag, degrees, s);
        case 2: return dispatch_pair_runs<2>(links, order, B, N, tabs, P, cards, cards_stride, *prm, flags, out, err_flag, degrees, s);
        default: return dispatch_pair_runs<3>(links, order, B, N, tabs, P, cards, cards_stride, *prm, flags, out, err_flag, degrees, s);
    }
}

extern "C" int ss_pair_features_grouped(const int64_t *links, const int32_t *order, int64_t B, int64_t N, int32_t h,
                                        const uint32_t *const *mh, int32_t P, const uint8_t *const *hll, const float *cards,
                                        int64_t cards_stride, const ss_hll_params *prm, uint32_t flags, const float *degrees, float *out,
                                        int32_t *err_flag, void *stream)
{
    return pair_features_grouped_impl(-1, links, order, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, degrees, out, err_flag, stream);
}

// measurement / test hook (subgraph_sketch_debug.h): the same call with the kernel forced -- 0 ordinary, 1 run-aware
extern "C" int ss_pair_features_grouped_kernel(int32_t which, const int64_t *links, const int32_t *order, int64_t B, int64_t N, int32_t h,
                                               const uint32_t *const *mh, int32_t P, const uint8_t *const *hll, const float *cards,
                                               int64_t cards_stride, const ss_hll_params *prm, uint32_t flags, const float *degrees,
                                               float *out, int32_t *err_flag, void *stream)
{
    if (which != 0 && which != 1) return SS_ERR_INVALID_ARG;
    return pair_features_grouped_impl(which, links, order, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, degrees, out, err_flag, stream);
}
