// ss_first_hop.hip -- hop 1 computed straight from node ids (fused hop-0 init + first propagation).
//
// Hop-0 sketches are pure functions of the node id (reference hashing.py:118-137): row t of the MinHash
// table is ((a_j * hv_t + b_j) mod 2^64 mod (2^61-1)) & 0xFFFFFFFF, row t of the HLL table has the single
// register (hv_t & (m-1)) = rank(hv_t).  Reading them back from HBM for the first propagation
// (hashing.py:160-161 at k = 1) costs (E' + N) * 768 B; recomputing them in registers costs 3 integer
// multiplies + 7 simple VALU per (neighbour, permutation) and no table traffic at all -- the kernel reads only
// the CSR and writes the hop-1 rows.  Results are identical to ss_minhash_init + ss_hll_init + ss_propagate.
// (VALU-bound: v_mul_lo_u32 / v_mad_u64_u32 issue at quarter rate, 16 cycles per wave64 on gfx950; measured
// 0.17 ms vs 0.36 ms for init + propagate on the bench graph.)
//
// Mapping: one wavefront per destination row.  Lane l owns permutations l, l+64, .. (P/64 of them) and
// HLL registers 4l..4l+3 (M = 256).  Neighbour ids are fetched 64 at a time (one coalesced load), every
// lane hashes ITS neighbour (64 hashes in parallel) and scatter-maxes that neighbour's single HLL register
// into the wave's LDS row; then the wave walks the batch with v_readlane: the neighbour's hash is
// wave-uniform (SGPR pair), each lane evaluates its own permutations on it.
// Hub rows are hub units (ss_hub.hpp): a whole workgroup per row or slice of a row, waves take alternate 64-neighbour
// batches and combine through LDS atomics -- the HLL side hosted by the leading workgroups of hll_first_hop_kernel's launch,
// the MinHash side by the fused kernel's (ss_fused_hop.hip) or by first_hop_hub_kernel, a launch of its own.
#include <cstdlib>
#include <type_traits>

#include "ss_hub.hpp"

namespace ss {

#ifndef SS_FORCE_EXACT_FIRST_HOP
#define SS_FORCE_EXACT_FIRST_HOP 0
#endif
constexpr bool kForceExactFirstHop = SS_FORCE_EXACT_FIRST_HOP;  // build-time switch: every row through the exact walk

template <int PPL /* permutations per lane = P / 64 */, bool DO_MH, bool DO_HLL>
__global__ __launch_bounds__(256) void first_hop_kernel(GraphArgs g, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb,
                                                        uint32_t *__restrict__ mh_out, int p, uint8_t *__restrict__ hll_out,
                                                        float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm,
                                                        bool skip_hubs)
{
    __shared__ EstimatorLds lds;
    __shared__ __attribute__((aligned(16))) uint32_t hll_rows[256 / kWave][256];  // one u32 per register and wave
    const bool want_cards = DO_HLL && cards_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);

    constexpr int P = PPL * kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t i = g.row0 + (int64_t)blockIdx.x * (blockDim.x / kWave) + wave;
    if (i >= g.row1) return;
    const int64_t rb = g.rowptr[i];
    const int deg = (int)(g.rowptr[i + 1] - rb);
    if (skip_hubs && deg > g.hub_threshold) return;  // left to first_hop_hub_kernel

    uint64_t a[PPL], b[PPL];
    uint32_t acc[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        a[q] = DO_MH ? pa[lane + kWave * q] : 0ULL;
        b[q] = DO_MH ? pb[lane + kWave * q] : 0ULL;
        acc[q] = 0xFFFFFFFFu;
    }
    uint32_t *my_row = hll_rows[wave];
    *reinterpret_cast<u32x4 *>(my_row + 4 * lane) = u32x4{0u, 0u, 0u, 0u};

    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;
    const int total = deg + (i < n_self ? 1 : 0);
    if (DO_MH && !DO_HLL && !kForceExactFirstHop) {
        // MinHash alone: two-phase walk (one multiply per neighbour and permutation); the rare ambiguous row is redone exactly
        const bool amb = total > 0 && first_hop_minhash_fast<PPL>(g.col + rb, deg, total, i, a, b, acc, lane);
        if (__any(amb)) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) acc[q] = 0xFFFFFFFFu;
            first_hop_walk<PPL, true, false>(g.col + rb, deg, total, i, 0, 1, p, a, b, acc, my_row, lane);
        }
    } else {
        first_hop_walk<PPL, DO_MH, DO_HLL>(g.col + rb, deg, total, i, 0, 1, p, a, b, acc, my_row, lane);
    }

    if (DO_MH) {
        if (total == 0) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) acc[q] = 0u;  // no in-edge, no self loop: all-zero row (PyG default)
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            mh_out[i * P + lane + kWave * q] = acc[q];
            mirror_mh1(g.mir, i * P + lane + kWave * q, acc[q]);
        }
    }
    if (!DO_HLL) return;
    // the wave's LDS row is only touched by this wave: a wave-level fence orders the atomics before the read
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t regs = pack_hll_quad(my_row, lane);  // HLL registers 4*lane .. 4*lane+3
    *reinterpret_cast<uint32_t *>(hll_out + i * 256 + 4 * lane) = regs;
    mirror_hll4(g.mir, i * 256 + 4 * lane, regs);

    if (want_cards) {
        int nonzero = 0;
        float hsum = 0.0f;
        hll_dword_stats(regs, nonzero, hsum);
        for (int off = 1; off < kWave; off <<= 1) {
            nonzero += __shfl_xor(nonzero, off);
            hsum += __shfl_xor(hsum, off);
        }
        if (lane == 0) {
            const float card = hll_estimate(est, 256 - nonzero, hsum);
            cards_out[i * cards_stride] = card;
            mirror_card(g.mir, i * cards_stride, card);
        }
    }
}

// ---- MinHash first hop, several consecutive rows per wavefront ---------------------------------------------------------
// first_hop_kernel spends about as many VALU instructions per row on fixed work (permutation parameters, neighbour-id
// load + splitmix64, winner fetch, exact evaluation: ~105) as on the walk itself at the bench graph's degree (9 per
// neighbour x 11), and the kernel is bound by VALU issue (VALUBusy 88 %).  Rows that are consecutive in the CSR are also
// consecutive in `col`, so a wavefront that owns kRowsPerWave consecutive rows
//   * loads a / b and sets up its lane constants once,
//   * fetches and hashes the neighbour ids of ALL its rows in batches of 64 - kRowsPerWave col entries with one coalesced
//     load + one splitmix64 per batch (the last kRowsPerWave lanes of every batch hold the hashes of the rows' own ids: the
//     implicit self loop of row r is slot 64 - kRowsPerWave + r of whatever batch is current),
//   * and then walks row after row over its slice of the batch (two-phase evaluation of ss_walks.hpp: phase 1 finds which
//     neighbour attains the minimum, phase 2 evaluates that one exactly).
// Rows flagged ambiguous (and rows that list themselves: duplicates of the implicit self loop) are redone by the exact walk.
// (The walk itself is MinhashRows of ss_walks.hpp, shared with the fused kernel of ss_fused_hop.hip.)
constexpr int kRowsPerWave = 8;

template <int PPL, int R, bool MIR>
__global__ __launch_bounds__(256) void first_hop_rows_kernel(GraphArgs g, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb,
                                                             uint32_t *__restrict__ mh_out, int p, bool skip_hubs)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    report_hub_rows(g);
    MinhashRows<PPL, R, MIR> m;
    if (!m.init(g, g.row0 + ((int64_t)blockIdx.x * (blockDim.x / kWave) + wave) * R, pa, pb, p, skip_hubs)) return;
    for (int r = 0; r < m.rows; ++r) m.row(r, mh_out);  // wave-uniform
}

// HLL-only first hop, latency-optimised: one 16-lane DPP row per destination (4 destinations in flight per wave).
// Used when the caller asks for the HLL sketch alone (the two-stream build runs the HLL chain beside the MinHash
// chain); the one-row-per-wave kernel above is a single dependent chain per wave and takes 4x longer for this.
#ifndef SS_HLL_ROWS
#define SS_HLL_ROWS 4
#endif
constexpr int kHllRows = SS_HLL_ROWS;  // (2, 6 and 8 measured the same 25 us on the bench graph: the kernel is not bound by its prefetch depth;
                                       // round 6, -DSS_HLL_ROWS=3 / 4 / 6 / 8: 25.5 / 25.7 / 25.4 / 37.9 us -- 8 spills 100 bytes at the 64-VGPR budget)

// PPL only matters to the leading workgroups (hub units, ss_hub.hpp): with hub_mh_out they also serve the hop-1 MinHash table
// (P = 64 * PPL) -- ss_fused_hop_stage's row kernel has no register to spare for them.  The register allocator is held to the 64
// VGPRs of the row path's eight wavefronts per SIMD; what the MinHash units spill they spill on their own, cold, path.
template <int PPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void hll_first_hop_kernel(
    GraphArgs g, int p, uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm, bool skip_hubs,
    int hub_blocks, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb, uint32_t *__restrict__ hub_mh_out, bool interleave)
{
    // only the linear-counting table is staged (this kernel is only launched for p = 8: 257 entries): a hop-1 row leaves the
    // linear-counting range at 147 neighbours, and those few read raw / bias from global memory.  17 KB of LDS and 60 VGPRs
    // instead of 24.5 KB and 73: 8 workgroups per CU instead of 6 (24.4-25.9 -> 22.9 us on the bench graph)
    __shared__ LcLds<257> lds;
    __shared__ __attribute__((aligned(16))) uint32_t rows[256 / kRow][256];  // one u32 per register and 16-lane group
    const bool want_cards = cards_out != nullptr;
    report_hub_rows(g);
    EstimatorTables est;
    if (want_cards) est = stage_lc_only(lds, prm);
    if ((int)blockIdx.x < hub_blocks) {  // workgroup-uniform: a leading workgroup serves hub units of the hop-1 HLL table (its rows
                                         // leave the linear-counting range: raw / bias come from global memory)
        static_assert(sizeof(FirstHopHubLds<PPL>) <= sizeof(rows), "the hub units' LDS rows live in the row image");
        FirstHopHubLds<PPL> &hub = *reinterpret_cast<FirstHopHubLds<PPL> *>(&rows[0][0]);
        // (with hub_mh_out the first QUARTER of the leading workgroups serves the HLL units, the rest the MinHash units -- tickets of
        // their own; a MinHash unit costs several times an HLL unit, and this launch is a short one to hide them in)
        const int first = hub_mh_out ? hub_blocks / 4 : hub_blocks;
        if ((int)blockIdx.x < first)
            first_hop_hub_units<PPL, kHubLeadWaves, false, true>(g, (int)blockIdx.x, first, nullptr, nullptr, nullptr, p, hll_out, cards_out,
                                                                 cards_stride, est, want_cards, hub, false);
        else
            first_hop_hub_units<PPL, kHubLeadWaves, true, false>(g, (int)blockIdx.x - first, hub_blocks - first, pa, pb, hub_mh_out, p, nullptr,
                                                                 nullptr, 0, est, false, hub, kForceExactFirstHop);
        return;
    }
    const int l = threadIdx.x & (kRow - 1);
    const int grp = threadIdx.x / kRow;
    // kHllRows rows per lane group, one after the other through the same LDS row image; the row bounds and the first
    // neighbour ids of ALL of them are requested up front, so the two dependent global round trips (rowptr -> col) of
    // the later rows hide under the work of the earlier ones
    // Row r of lane group `grp` is row r * 16 + grp of the workgroup's 64: at any moment the 16 lane groups work on 16 CONSECUTIVE rows
    // -- consecutive stretches of `col` and of the output (round 6; SS_HLL_ROW_MAP=0: the four consecutive rows per lane group of
    // rounds 2-5, whose shared boundary lines were fetched twice at ogbl-ppa size)
    const int groups = (int)(blockDim.x / kRow);
    const int64_t wg_first = g.row0 + (int64_t)((int)blockIdx.x - hub_blocks) * groups * kHllRows;
    const int64_t first = interleave ? wg_first + grp : wg_first + (int64_t)grp * kHllRows;
    const int64_t row_step = interleave ? groups : 1;
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;
    uint32_t *row = rows[grp];
    int64_t rbs[kHllRows];
    int degs[kHllRows], nid0[kHllRows];
    bool oks[kHllRows];
#pragma unroll
    for (int r = 0; r < kHllRows; ++r) {
        oks[r] = first + r * row_step < g.row1;
        const int64_t i = oks[r] ? first + r * row_step : g.row1 - 1;
        rbs[r] = g.rowptr[i];
        degs[r] = (int)(g.rowptr[i + 1] - rbs[r]);
    }
    // (unconditional loads from an address that is always valid -- lanes past the end of their row read the first word of
    // rowptr and never look at it: a load under `if (l < deg)` is awaited with vmcnt(0) at the end of its branch, which put
    // the four rows' id loads one round trip behind the other)
    const int32_t *always_valid = reinterpret_cast<const int32_t *>(g.rowptr);
    // ... of the first TWO rows; the first ids of row r + 2 are requested when row r is about to be walked.  (All four up front, as
    // rounds 2-4 had it: the line a later row's first ids sit in also holds its next ids, and by the time a lane group has walked
    // three rows of 74 neighbours -- ogbl-ppa -- that line has left the L2 again and is fetched a second time: PMC 0.398 GB for
    // 0.324 GB algorithmic at that size against 1.03x at collab size, where four rows span two lines; VERDICT r4 #5.)
#pragma unroll
    for (int r = 0; r < kHllRows; ++r) nid0[r] = r < 2 ? *(l < degs[r] ? g.col + rbs[r] + l : always_valid) : 0;
#pragma unroll
    for (int r = 0; r < kHllRows; ++r) {
        if (r + 2 < kHllRows) nid0[r + 2] = *(l < degs[r + 2] ? g.col + rbs[r + 2] + l : always_valid);
        const bool ok = oks[r];
        const int64_t i = ok ? first + r * row_step : g.row1 - 1;
        const int deg = degs[r];
        const bool hub = skip_hubs && deg > g.hub_threshold;
        const int total = hub ? 0 : deg + (i < n_self ? 1 : 0);
        const int32_t *nb = g.col + rbs[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4 *>(row + 64 * k + 4 * l) = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (the id of the lane's NEXT neighbour is requested before the current one is hashed -- unconditionally, from an address
        // that always exists: rows of more than 16 neighbours otherwise pay one exposed round trip per 16 neighbours)
        int cur = nid0[r];
        int fresh = 0;  // registers this lane was the FIRST to set (the LDS atomic returns what was there: 0 exactly once per register)
        for (int t = l; t < total; t += kRow) {
            const int nxt = *(t + kRow < deg ? nb + t + kRow : always_valid);
            const int64_t nid = t < deg ? (int64_t)cur : i;
            cur = nxt;
            const uint64_t hv = hash_u64((uint64_t)(nid + 1));
            const uint64_t bits = hv >> p;
            const int bl = bits ? 64 - __builtin_clzll(bits) : 0;
            fresh += atomicMax(&row[(uint32_t)hv & 255u], (uint32_t)((64 - p) - bl + 1)) == 0u ? 1 : 0;  // (ranks are >= 1)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // lane l owns registers 16l .. 16l+15
        u32x4 packed;
        uint32_t *pw = reinterpret_cast<uint32_t *>(&packed);
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[k] = pack_hll_quad(row + 16 * l, k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int nonzero = 0;
        float hsum = 0.0f;
        if (want_cards) {
            // A hop-1 row has at most deg + 1 non-zero registers, so nearly every row -- every one below 147 neighbours at p = 8 --
            // is estimated by linear counting, which needs the number of zero registers and nothing else (hashing.py:221-226):
            // that number comes out of the atomics above.  Only a wavefront holding a row that leaves the linear-counting range
            // digests the packed registers for the harmonic sum (12 instructions per dword, a third of this kernel's VALU work).
            nonzero = row16_sum_i(fresh);
            const int zeros = 256 - nonzero;
            if (__any(!(zeros > 0 && zeros >= est.lc_min_zeros))) {  // wave-uniform
                int nz2 = 0;
                hll_dword_stats(packed.x, nz2, hsum);
                hll_dword_stats(packed.y, nz2, hsum);
                hll_dword_stats(packed.z, nz2, hsum);
                hll_dword_stats(packed.w, nz2, hsum);
                hsum = row16_sum_f(hsum);
            }
        }
        if (ok && !hub) {
            *reinterpret_cast<u32x4 *>(hll_out + i * 256 + 16 * l) = packed;
            mirror_hll16(g.mir, i * 256 + 16 * l, packed);
            if (want_cards && l == 0) {
                const float card = hll_estimate(est, 256 - nonzero, hsum);
                cards_out[i * cards_stride] = card;
                mirror_card(g.mir, i * cards_stride, card);
            }
        }
    }
}

// ---- hub units as a launch of their own (16 wavefronts per workgroup) --------------------------------------------------------
constexpr int kHubThreads = 1024;
constexpr int kHubWaves = kHubThreads / kWave;
constexpr int kHubGrid = 256;  // one workgroup per CU; workgroups beyond the unit count exit at once

template <int PPL, bool DO_MH, bool DO_HLL>
__global__ __launch_bounds__(kHubThreads) void first_hop_hub_kernel(GraphArgs g, const uint64_t *__restrict__ pa,
                                                                    const uint64_t *__restrict__ pb, uint32_t *__restrict__ mh_out, int p,
                                                                    uint8_t *__restrict__ hll_out, float *__restrict__ cards_out,
                                                                    int64_t cards_stride, ss_hll_params prm)
{
    __shared__ EstimatorLds lds;
    __shared__ FirstHopHubLds<PPL> hub;
    const HubCounts n = hub_counts(g);
    if ((int)blockIdx.x >= n.hubs + n.slices) return;  // the common case (no hub rows) costs three scalar loads per workgroup
    const bool want_cards = DO_HLL && cards_out != nullptr;
    EstimatorTables est = {};
    if (want_cards) est = stage_tables(lds, prm);
    first_hop_hub_units<PPL, kHubWaves, DO_MH, DO_HLL>(g, (int)blockIdx.x, (int)gridDim.x, pa, pb, mh_out, p, hll_out, cards_out, cards_stride, est,
                                                       want_cards, hub, kForceExactFirstHop);
}

int launch_first_hop_hub_only(const GraphArgs &g, const uint64_t *a, const uint64_t *b, int P, uint32_t *mh_out, int p, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream);
template <int PPL>
void launch_minhash_rows(const GraphArgs &g, const uint64_t *a, const uint64_t *b, uint32_t *mh_out, int p, const ss_hll_params &prm,
                         bool hubs, hipStream_t s);

template <int PPL, bool DO_MH, bool DO_HLL>
int launch_first_hop_v(const GraphArgs &g, const uint64_t *a, const uint64_t *b, uint32_t *mh_out, int p, uint8_t *hll_out,
                       float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t s)
{
    const int64_t blocks = (g.rows() + 3) / 4;
    const bool hubs = g.hub_rows && g.hub_count;
    {
        ProfileSpan span(s, DO_MH && !DO_HLL ? SS_PROF_FIRST_HOP_MH : SS_PROF_TAGS);
        if constexpr (DO_MH && !DO_HLL)
            launch_minhash_rows<PPL>(g, a, b, mh_out, p, prm, hubs, s);
        else
            hipLaunchKernelGGL((first_hop_kernel<PPL, DO_MH, DO_HLL>), dim3((unsigned)blocks), dim3(256), 0, s, g, a, b, mh_out, p, hll_out,
                               cards_out, cards_stride, prm, hubs);
    }
    SS_LAUNCH_CHECK();
    if (hubs) {
        note_hub_call();
        hipLaunchKernelGGL((first_hop_hub_kernel<PPL, DO_MH, DO_HLL>), dim3(kHubGrid), dim3(kHubThreads), 0, s, g, a, b, mh_out, p,
                           hll_out, cards_out, cards_stride, prm);
        SS_LAUNCH_CHECK();
    }
    return SS_OK;
}

// MinHash rows of one launch: kRowsPerWave consecutive rows per wavefront; SS_FIRST_HOP_ROWS=0 selects the one-row-per-wave
// kernel (A/B measurements, and the build-time switch that forces the exact walk)
template <int PPL>
void launch_minhash_rows(const GraphArgs &g, const uint64_t *a, const uint64_t *b, uint32_t *mh_out, int p, const ss_hll_params &prm,
                         bool hubs, hipStream_t s)
{
    static const int rows_env = getenv("SS_FIRST_HOP_ROWS") ? atoi(getenv("SS_FIRST_HOP_ROWS")) : kRowsPerWave;
    if (kForceExactFirstHop || rows_env == 0) {
        hipLaunchKernelGGL((first_hop_kernel<PPL, true, false>), dim3((unsigned)((g.rows() + 3) / 4)), dim3(256), 0, s, g, a, b, mh_out, p,
                           (uint8_t *)nullptr, (float *)nullptr, (int64_t)0, prm, hubs);
        return;
    }
    auto go = [&](auto rows_tag) {
        constexpr int R = decltype(rows_tag)::value;
        constexpr int rows_per_block = 4 * R;
        const dim3 grid((unsigned)((g.rows() + rows_per_block - 1) / rows_per_block));
        if (g.mir.n > 0)
            hipLaunchKernelGGL((first_hop_rows_kernel<PPL, R, true>), grid, dim3(256), 0, s, g, a, b, mh_out, p, hubs);
        else
            hipLaunchKernelGGL((first_hop_rows_kernel<PPL, R, false>), grid, dim3(256), 0, s, g, a, b, mh_out, p, hubs);
    };
    if (rows_env == 4) go(std::integral_constant<int, 4>{});
    else if (rows_env == 16) go(std::integral_constant<int, 16>{});
    else if (rows_env == 2) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, kRowsPerWave>{});
}

// HLL first hop of the regular rows alone (ss_fused_hop_stage owns the hub pass that follows)
// (`lead` leading workgroups serve the hub units of the HLL table and -- with hub_mh_out, P = 64 * PPL values per row from the
// permutations a / b -- of the hop-1 MinHash table, ss_hub.hpp; 0: the caller launches first_hop_hub_kernel)
template <int PPL>
static int launch_hll_rows(const GraphArgs &g, int p, uint8_t *hll_out, float *cards_out, int64_t cards_stride, const ss_hll_params &prm,
                           bool skip_hubs, int lead, const uint64_t *a, const uint64_t *b, uint32_t *hub_mh_out, hipStream_t s)
{
    static const bool interleave = !(getenv("SS_HLL_ROW_MAP") && atoi(getenv("SS_HLL_ROW_MAP")) == 0);  // (measurement hook, see the kernel)
    ProfileSpan span(s, SS_PROF_FIRST_HOP_HLL);
    hipLaunchKernelGGL((hll_first_hop_kernel<PPL>), dim3((unsigned)((g.rows() + 16 * kHllRows - 1) / (16 * kHllRows) + lead)), dim3(256), 0, s, g,
                       p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, a, b, hub_mh_out, interleave);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int launch_hll_first_hop_rows(const GraphArgs &g, int p, uint8_t *hll_out, float *cards_out, int64_t cards_stride, const ss_hll_params &prm,
                              bool skip_hubs, int lead, const uint64_t *a, const uint64_t *b, uint32_t *hub_mh_out, int P, hipStream_t s)
{
    if (lead == 0 || !hub_mh_out) return launch_hll_rows<1>(g, p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, nullptr, nullptr, nullptr, s);
    lead *= 2;  // (a quarter for the HLL units, the rest for the MinHash units)
    switch (P / kWave) {
        case 1: return launch_hll_rows<1>(g, p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, a, b, hub_mh_out, s);
        case 2: return launch_hll_rows<2>(g, p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, a, b, hub_mh_out, s);
        case 3: return launch_hll_rows<3>(g, p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, a, b, hub_mh_out, s);
        default: return launch_hll_rows<4>(g, p, hll_out, cards_out, cards_stride, prm, skip_hubs, lead, a, b, hub_mh_out, s);
    }
}

template <int PPL>
int launch_first_hop(const GraphArgs &g, const uint64_t *a, const uint64_t *b, uint32_t *mh_out, int p, uint8_t *hll_out,
                     float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t s)
{
    if (mh_out && hll_out) {
        // both sketches: the latency-optimised HLL kernel + the MinHash kernel beat the combined kernel (37 + 134 us vs
        // 184 us on the bench graph); one hub pass serves both
        const bool hubs = g.hub_rows && g.hub_count;
        const int lead = hub_lead_blocks(hubs);
        const int rc = launch_hll_first_hop_rows(g, p, hll_out, cards_out, cards_stride, prm, hubs, lead, a, b, mh_out, PPL * kWave, s);
        if (rc != SS_OK) return rc;
        {
            ProfileSpan span(s, SS_PROF_FIRST_HOP_MH);
            launch_minhash_rows<PPL>(g, a, b, mh_out, p, prm, hubs, s);
        }
        SS_LAUNCH_CHECK();
        if (lead > 0) return SS_OK;  // (the hub units of both tables were hosted by the HLL launch)
        return launch_first_hop_hub_only(g, a, b, PPL * kWave, mh_out, p, hll_out, cards_out, cards_stride, prm, s);
    }
    if (mh_out) return launch_first_hop_v<PPL, true, false>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, prm, s);
    // HLL alone: 16-lane-per-row kernel for the regular rows, the cooperative hub kernel for the rest
    const bool hubs = g.hub_rows && g.hub_count;
    const int lead = hub_lead_blocks(hubs);
    const int rc = launch_hll_first_hop_rows(g, p, hll_out, cards_out, cards_stride, prm, hubs, lead, nullptr, nullptr, nullptr, 0, s);
    if (rc != SS_OK) return rc;
    if (hubs && lead == 0) {
        hipLaunchKernelGGL((first_hop_hub_kernel<PPL, false, true>), dim3(kHubGrid), dim3(kHubThreads), 0, s, g, a, b, mh_out, p,
                           hll_out, cards_out, cards_stride, prm);
        SS_LAUNCH_CHECK();
    }
    return SS_OK;
}

template <int PPL>
static int hub_only(const GraphArgs &g, const uint64_t *a, const uint64_t *b, uint32_t *mh_out, int p, uint8_t *hll_out, float *cards_out,
                    int64_t cards_stride, const ss_hll_params &prm, hipStream_t s)
{
    if (mh_out && hll_out)
        hipLaunchKernelGGL((first_hop_hub_kernel<PPL, true, true>), dim3(kHubGrid), dim3(kHubThreads), 0, s, g, a, b, mh_out, p, hll_out,
                           cards_out, cards_stride, prm);
    else if (mh_out)
        hipLaunchKernelGGL((first_hop_hub_kernel<PPL, true, false>), dim3(kHubGrid), dim3(kHubThreads), 0, s, g, a, b, mh_out, p, hll_out,
                           cards_out, cards_stride, prm);
    else
        hipLaunchKernelGGL((first_hop_hub_kernel<PPL, false, true>), dim3(kHubGrid), dim3(kHubThreads), 0, s, g, a, b, mh_out, p, hll_out,
                           cards_out, cards_stride, prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int launch_first_hop_hub_only(const GraphArgs &g, const uint64_t *a, const uint64_t *b, int P, uint32_t *mh_out, int p, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    if (!g.hub_rows || !g.hub_count || (!mh_out && !hll_out)) return SS_OK;
    ProfileSpan span(stream, SS_PROF_HUB);
    switch (P / kWave) {
        case 1: return hub_only<1>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, prm, stream);
        case 2: return hub_only<2>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, prm, stream);
        case 3: return hub_only<3>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, prm, stream);
        default: return hub_only<4>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, prm, stream);
    }
}

}  // namespace ss

extern "C" int ss_first_hop(const ss_csr_graph *graph, const uint64_t *a, const uint64_t *b, int32_t P,
                            uint32_t *mh_out, int32_t p, uint8_t *hll_out, float *cards_out, int64_t cards_stride,
                            const ss_hll_params *prm, void *stream)
{
    using namespace ss;
    if (!graph || graph->num_nodes < 0 || !graph->rowptr) return SS_ERR_INVALID_ARG;
    // (the MinHash rows do not depend on p: a MinHash-only call is served for every hll_p; the HLL row image is 256 registers)
    if ((hll_out && p != 8) || P <= 0 || P % kWave || P > 256) return SS_ERR_UNSUPPORTED;  // caller falls back to init + propagate
    const int64_t N = graph->num_nodes;
    if (N == 0) return SS_OK;
    if ((!mh_out && !hll_out) || (mh_out && (!a || !b)) || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;  // either sketch may be NULL; a / b only feed MinHash
    if ((graph->hub_rows == nullptr) != (graph->hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    ss_hll_params p0 = {};
    if (cards_out) {
        if (!hll_out) return SS_ERR_INVALID_ARG;
        const int rc = check_params(prm);
        if (rc != SS_OK) return rc;
        if (prm->p != p) return SS_ERR_INVALID_ARG;
        p0 = *prm;
    }
    if (!row_range_ok(*graph)) return SS_ERR_INVALID_ARG;
    const GraphArgs g = to_args(*graph);
    if (g.rows() == 0) return SS_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (P / kWave) {
        case 1: return launch_first_hop<1>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, p0, s);
        case 2: return launch_first_hop<2>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, p0, s);
        case 3: return launch_first_hop<3>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, p0, s);
        default: return launch_first_hop<4>(g, a, b, mh_out, p, hll_out, cards_out, cards_stride, p0, s);
    }
}
