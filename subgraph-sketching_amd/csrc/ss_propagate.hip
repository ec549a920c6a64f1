// ss_propagate.hip -- one hop of sketch propagation (CSR pull) + fused HLL++ cardinality.
//
// Replaces MinhashPropagation / HllPropagation (reference hashing.py:28-45) and the per-hop hll_count of
// build_hash_tables (hashing.py:160-163).  HBM-bound: per hop it streams (E' + N) sketch rows.
//
// Mapping: one wavefront per destination row.  A 16-byte chunk per lane; a MinHash row of P u32 is
// CM = P/4 chunks, an HLL row of M bytes is CH = M/16 chunks.  The wave is split into G = 64/SG
// sub-groups (SG = pow2 >= chunks per row, <= 64) that walk G neighbours concurrently
// (P=128: 2 neighbours per step, M=256: 4), so every global load is a coalesced dwordx4 and a step
// moves 1 KiB per wave.  Partial min / max are combined across sub-groups once per row with
// cross-lane shuffles.  The implicit self loop (i < n_self) is one extra virtual neighbour.
//
// Hub rows (in-degree > graph.hub_threshold, listed by ss_csr_build) would serialise tens of thousands of
// dependent 1 KiB loads on one wavefront (power-law graphs: ogbl-ppa, ogbl-citation2).  The row wavefronts
// skip them; the LEADING `hub_blocks` workgroups of the same launch walk them as hub units (ss_hub.hpp: a whole
// workgroup per row or slice of a row, partials combined through LDS).
#include "ss_hub.hpp"

namespace ss {

// TP / TM > 0: compile-time row sizes (fast path); 0: run-time.
template <int TP, int TM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void propagate_kernel(GraphArgs g, const uint32_t *__restrict__ mh_in, uint32_t *__restrict__ mh_out,
                                                        int P_rt, const uint8_t *__restrict__ hll_in, uint8_t *__restrict__ hll_out,
                                                        int M_rt, float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm,
                                                        bool skip_hubs, int hub_blocks, const uint8_t *__restrict__ hub_hll_in,
                                                        uint8_t *__restrict__ hub_hll_out, float *__restrict__ hub_cards_out)
{
    __shared__ EstimatorLds lds;
    // (workgroup-uniform; the row workgroups of a MinHash-only launch must not pay for the tables a leading workgroup needs: staged
    // by all 59 000 workgroups of the bench graph's launch they cost it 13 us)
    const bool hub_hll_block = TP == 128 && TM == 256 && hub_hll_out != nullptr && (int)blockIdx.x < hub_blocks && (int)blockIdx.x >= hub_blocks / 2;
    const bool want_cards = (cards_out != nullptr && hll_out != nullptr) || (hub_hll_block && hub_cards_out != nullptr);
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);
    if constexpr (TP == 128 && TM == 256) {  // (this instantiation is only ever launched for the MinHash rows alone)
        if ((int)blockIdx.x < hub_blocks) {
            // workgroup-uniform: a leading workgroup serves hub units (ss_hub.hpp) of this hop's MinHash table or -- with hub_hll_out,
            // the second half of the leading workgroups, tickets of their own -- of its HLL table (ss_fused_hop_stage: the kernel
            // that computes the HLL rows has no register to spare for them)
            __shared__ TableHubLds<kHubLeadWaves> hub;
            const int half = hub_hll_out ? hub_blocks / 2 : hub_blocks;
            if ((int)blockIdx.x < half)
                table_hub_units<kHubLeadWaves, true, false>(g, (int)blockIdx.x, half, mh_in, mh_out, nullptr, nullptr, nullptr, 0, est, false, hub);
            else
                table_hub_units<kHubLeadWaves, false, true>(g, (int)blockIdx.x - half, hub_blocks - half, nullptr, nullptr, hub_hll_in, hub_hll_out,
                                                            hub_cards_out, cards_stride, est, hub_cards_out != nullptr, hub);
            return;
        }
    }

    const bool row_cards = cards_out != nullptr && hll_out != nullptr;
    const int P = TP ? TP : P_rt;
    const int M = TM ? TM : M_rt;
    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform row id (readfirstlane makes the uniformity visible to the compiler: scalar loads, scalar loop control)
    int64_t i = g.row0 + (int64_t)((int)blockIdx.x - hub_blocks) * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    if (g.row_list) {  // ss_minhash_hop_rows: the q-th wavefront computes row row_list[q]
        const int64_t q = i - g.row0;
        if (q >= g.n_list) return;
        i = g.row_list[q];
        i = i < 0 ? i + g.N : i;
        if ((uint64_t)i >= (uint64_t)g.N) return;  // the query kernel reports such ids; nothing to compute here
    } else if (i >= g.row1) {
        return;
    }

    const int64_t rb = g.rowptr[i];
    const int deg = (int)(g.rowptr[i + 1] - rb);
    if (skip_hubs && deg > g.hub_threshold) return;  // a hub unit (ss_hub.hpp)
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;
    const int total = deg + (i < n_self ? 1 : 0);
    const int32_t *nb = g.col + rb;

    // ---------------- MinHash: min over neighbours ----------------
    if (mh_out) {
        const int CM = P >> 2;
        const int SG = TP ? (pow2_ceil(TP >> 2) > kWave ? kWave : pow2_ceil(TP >> 2)) : (pow2_ceil(CM) > kWave ? kWave : pow2_ceil(CM));
        const int G = kWave / SG;
        const int sg = lane / SG, cl = lane % SG;
        for (int cb = 0; cb < CM; cb += SG) {
            const int c = cb + cl;
            const bool act = c < CM;
            u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if constexpr (TP == 128) acc = minhash_walk128(mh_in, nb, deg, total, i, lane);
            else if (act) acc = minhash_walk(mh_in, nb, deg, total, i, sg, G, P, c);
            for (int off = SG; off < kWave; off <<= 1) acc = min4(acc, shfl_xor4(acc, off));
            if (total == 0) acc = u32x4{0u, 0u, 0u, 0u};
            if (act && sg == 0) {
                *reinterpret_cast<u32x4 *>(mh_out + i * P + 4 * c) = acc;
                mirror_mh4(g.mir, i * P + 4 * c, acc);
            }
        }
    }

    // ---------------- HLL: byte-wise max over neighbours, fused cardinality ----------------
    if (hll_out) {
        const int CH = M >> 4;
        const int SG = TM ? (pow2_ceil(TM >> 4) > kWave ? kWave : pow2_ceil(TM >> 4)) : (pow2_ceil(CH) > kWave ? kWave : pow2_ceil(CH));
        const int G = kWave / SG;
        const int sg = lane / SG, cl = lane % SG;
        int nonzero = 0;
        float hsum = 0.0f;
        for (int cb = 0; cb < CH; cb += SG) {
            const int c = cb + cl;
            const bool act = c < CH;
            u32x4 acc = {0u, 0u, 0u, 0u};
            if (act) acc = hll_walk(hll_in, nb, deg, total, i, sg, G, M, c);
            for (int off = SG; off < kWave; off <<= 1) acc = bytemax16(acc, shfl_xor4(acc, off));
            if (act && sg == 0) {
                *reinterpret_cast<u32x4 *>(hll_out + i * M + 16 * c) = acc;
                mirror_hll16(g.mir, i * M + 16 * c, acc);
                if (row_cards) {
                    hll_dword_stats(acc.x, nonzero, hsum);
                    hll_dword_stats(acc.y, nonzero, hsum);
                    hll_dword_stats(acc.z, nonzero, hsum);
                    hll_dword_stats(acc.w, nonzero, hsum);
                }
            }
        }
        if (row_cards) {
            // lanes of sub-group 0 hold partial stats; the rest hold 0
            if (SG == kRow) {  // the 16 lanes of sub-group 0 are one DPP row
                nonzero = row16_sum_i(nonzero);
                hsum = row16_sum_f(hsum);
            } else {
                for (int off = 1; off < kWave; off <<= 1) {
                    nonzero += __shfl_xor(nonzero, off);
                    hsum += __shfl_xor(hsum, off);
                }
            }
            if (lane == 0) {
                const float card = hll_estimate(est, M - nonzero, hsum);
                cards_out[i * cards_stride] = card;
                mirror_card(g.mir, i * cards_stride, card);
            }
        }
    }
}

// HLL-only hop, 4 destinations per wavefront (fast path of ss_propagate when the MinHash sketch is absent)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void hll_propagate_row16_kernel(GraphArgs g, const uint8_t *__restrict__ hll_in,
                                                                  uint8_t *__restrict__ hll_out, float *__restrict__ cards_out,
                                                                  int64_t cards_stride, ss_hll_params prm, bool skip_hubs, int hub_blocks)
{
    __shared__ EstimatorLds lds;
    const bool want_cards = cards_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);
    if ((int)blockIdx.x < hub_blocks) {  // workgroup-uniform: a leading workgroup serves hub units of this hop's HLL table
        __shared__ TableHubLds<kHubLeadWaves> hub;
        table_hub_units<kHubLeadWaves, false, true>(g, (int)blockIdx.x, hub_blocks, nullptr, nullptr, hll_in, hll_out, cards_out, cards_stride, est,
                                                    want_cards, hub);
        return;
    }
    // the 16 rows of this workgroup are dealt to its 16 lane groups in DEGREE order: the four rows that share a wavefront
    // then have similar degrees, and the walk of a wavefront lasts as long as its longest row
    __shared__ int s_deg[256 / kRow], s_owner[256 / kRow];
    const int grp = threadIdx.x / kRow, c16 = threadIdx.x & (kRow - 1);
    const int64_t first = g.row0 + (int64_t)((int)blockIdx.x - hub_blocks) * (blockDim.x / kRow);
    {
        const int64_t r = first + grp;
        const int d = r < g.row1 ? (int)(g.rowptr[r + 1] - g.rowptr[r]) : -1;
        if (c16 == 0) s_deg[grp] = d;
        __syncthreads();
        const int dj = s_deg[c16];
        const bool before = dj < d || (dj == d && c16 < grp);
        const unsigned long long b = __ballot(before);
        const int rank = __popcll((b >> (kRow * ((threadIdx.x & (kWave - 1)) / kRow))) & 0xFFFFull);
        if (c16 == 0) s_owner[rank] = grp;
        __syncthreads();
    }
    const int64_t row = first + s_owner[grp];
    hll_hop_row16(g, row < g.row1 ? row : -1, skip_hubs, hll_in, hll_out, cards_out, cards_stride, est, want_cards, threadIdx.x & (kRow - 1));
}

// ---- hub units as a launch of their own (16 wavefronts per workgroup): the shapes / calls whose row kernel does not host them,
// and SS_HUB_LAUNCHES=1; P = 128, M = 256 only
constexpr int kHubThreads = 1024;
constexpr int kHubWaves = kHubThreads / kWave;
constexpr int kHubGrid = 256;  // one workgroup per CU; workgroups beyond the unit count exit at once

template <bool DO_MH, bool DO_HLL>
__global__ __launch_bounds__(kHubThreads) void propagate_hub_kernel(GraphArgs g, const uint32_t *__restrict__ mh_in,
                                                                    uint32_t *__restrict__ mh_out, const uint8_t *__restrict__ hll_in,
                                                                    uint8_t *__restrict__ hll_out, float *__restrict__ cards_out,
                                                                    int64_t cards_stride, ss_hll_params prm)
{
    __shared__ EstimatorLds lds;
    __shared__ TableHubLds<kHubWaves> hub;
    const HubCounts n = hub_counts(g);
    if ((int)blockIdx.x >= n.hubs + n.slices) return;  // the common case (no hub rows) costs three scalar loads per workgroup
    const bool want_cards = DO_HLL && cards_out != nullptr;
    EstimatorTables est = {};
    if (want_cards) est = stage_tables(lds, prm);
    table_hub_units<kHubWaves, DO_MH, DO_HLL>(g, (int)blockIdx.x, (int)gridDim.x, mh_in, mh_out, hll_in, hll_out, cards_out, cards_stride, est,
                                              want_cards, hub);
}

template <int TP, int TM>
int launch_propagate(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, int P, const uint8_t *hll_in, uint8_t *hll_out, int M,
                     float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    const int rows_per_block = 256 / kWave;
    const int64_t blocks = (g.rows() + rows_per_block - 1) / rows_per_block;
    const bool hubs = TP == 128 && TM == 256 && g.hub_rows && g.hub_count;
    const int lead = hub_lead_blocks(hubs && !hll_out);  // (the <128, 256> row kernel hosts MinHash units only)
    {
        ProfileSpan span(stream, mh_out && !hll_out && TP == 128 ? SS_PROF_MINHASH_HOP : SS_PROF_TAGS);  // MinHash table hop
        hipLaunchKernelGGL((propagate_kernel<TP, TM>), dim3((unsigned)(blocks + lead)), dim3(256), 0, stream, g, mh_in, mh_out, P, hll_in,
                           hll_out, M, cards_out, cards_stride, prm, hubs, lead, (const uint8_t *)nullptr, (uint8_t *)nullptr, (float *)nullptr);
    }
    SS_LAUNCH_CHECK();
    if (hubs && lead == 0) return launch_propagate_hub_only(g, mh_in, mh_out, hll_in, hll_out, cards_out, cards_stride, prm, stream);
    return SS_OK;
}

// the MinHash table hop alone (P = 128): the regular rows and -- `lead` leading workgroups -- the hub units of the MinHash table
// and, with hub_hll_out, of the HLL table of the same hop (hub_cards_out / cards_stride / prm: its cardinalities)
int launch_minhash_hop(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, bool skip_hubs, int lead, const uint8_t *hub_hll_in,
                       uint8_t *hub_hll_out, float *hub_cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    ProfileSpan span(stream, SS_PROF_MINHASH_HOP, true);
    span.launch(propagate_kernel<128, 256>, dim3((unsigned)((g.rows() + 3) / 4 + lead)), dim3(256), g, mh_in, mh_out, 128,
                (const uint8_t *)nullptr, (uint8_t *)nullptr, 256, (float *)nullptr, cards_stride, prm, skip_hubs, lead, hub_hll_in, hub_hll_out,
                hub_cards_out);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// the hub units of a table hop as a launch of their own (16-wave workgroups): what the row launches above did not host
int launch_propagate_hub_only(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, const uint8_t *hll_in, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    if (!g.hub_rows || !g.hub_count || (!mh_out && !hll_out)) return SS_OK;
    ProfileSpan span(stream, SS_PROF_HUB);
    if (mh_out && hll_out)
        hipLaunchKernelGGL((propagate_hub_kernel<true, true>), dim3(kHubGrid), dim3(kHubThreads), 0, stream, g, mh_in, mh_out, hll_in, hll_out,
                           cards_out, cards_stride, prm);
    else if (mh_out)
        hipLaunchKernelGGL((propagate_hub_kernel<true, false>), dim3(kHubGrid), dim3(kHubThreads), 0, stream, g, mh_in, mh_out, hll_in, hll_out,
                           cards_out, cards_stride, prm);
    else
        hipLaunchKernelGGL((propagate_hub_kernel<false, true>), dim3(kHubGrid), dim3(kHubThreads), 0, stream, g, mh_in, mh_out, hll_in, hll_out,
                           cards_out, cards_stride, prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// the HLL table hop alone (M = 256): 4 destinations per wavefront + the hub units of the HLL table
static int launch_hll_hop(const GraphArgs &g, const uint8_t *hll_in, uint8_t *hll_out, float *cards_out, int64_t cards_stride,
                          const ss_hll_params &prm, bool hubs, int lead, hipStream_t stream)
{
    ProfileSpan span(stream, SS_PROF_HLL_HOP);
    hipLaunchKernelGGL(hll_propagate_row16_kernel, dim3((unsigned)((g.rows() + 15) / 16 + lead)), dim3(256), 0, stream, g, hll_in, hll_out,
                       cards_out, cards_stride, prm, hubs, lead);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

}  // namespace ss

// MinHash table hop of the listed rows only (+ the hub units, which always cover every hub row)
extern "C" int ss_minhash_hop_rows(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P, const int64_t *rows,
                                   int64_t n_rows, void *stream)
{
    using namespace ss;
    if (!graph || graph->num_nodes < 0 || !graph->rowptr || !mh_in || !mh_out || n_rows < 0 || (n_rows > 0 && !rows)) return SS_ERR_INVALID_ARG;
    if (P <= 0 || (P & 3) || P > 2048) return SS_ERR_INVALID_ARG;
    const int64_t N = graph->num_nodes;
    if (N == 0 || n_rows == 0) return SS_OK;
    if (N >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 33)) return SS_ERR_INVALID_ARG;
    if ((graph->hub_rows == nullptr) != (graph->hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    if (graph->row_begin != 0 || graph->row_end != 0) return SS_ERR_INVALID_ARG;  // a row list and a row range exclude each other
    GraphArgs g = to_args(*graph);
    g.row_list = rows;
    g.n_list = n_rows;
    hipStream_t s = (hipStream_t)stream;
    const bool hubs = P == 128 && g.hub_rows && g.hub_count;  // (the hub units of the table hops are P = 128 only, as in ss_propagate)
    const int lead = hub_lead_blocks(hubs);
    const unsigned blocks = (unsigned)((n_rows + 3) / 4 + lead);
    {
        ProfileSpan span(s, SS_PROF_MINHASH_ROWS);
        if (P == 128)
            hipLaunchKernelGGL((propagate_kernel<128, 256>), dim3(blocks), dim3(256), 0, s, g, mh_in, mh_out, 128, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, ss_hll_params{}, hubs, lead, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, (float *)nullptr);
        else
            hipLaunchKernelGGL((propagate_kernel<0, 0>), dim3(blocks), dim3(256), 0, s, g, mh_in, mh_out, P, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, ss_hll_params{}, false, 0, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, (float *)nullptr);
    }
    SS_LAUNCH_CHECK();
    if (!hubs || lead > 0) return SS_OK;
    GraphArgs all = to_args(*graph);
    return launch_propagate_hub_only(all, mh_in, mh_out, nullptr, nullptr, nullptr, 0, ss_hll_params{}, s);
}

extern "C" int ss_propagate(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                            const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                            float *cards_out, int64_t cards_stride, const ss_hll_params *prm, void *stream)
{
    using namespace ss;
    if (!graph || graph->num_nodes < 0 || !graph->rowptr) return SS_ERR_INVALID_ARG;
    const int64_t N = graph->num_nodes;
    if (N == 0) return SS_OK;
    if (!row_range_ok(*graph)) return SS_ERR_INVALID_ARG;
    if ((mh_in == nullptr) != (mh_out == nullptr) || (hll_in == nullptr) != (hll_out == nullptr)) return SS_ERR_INVALID_ARG;
    if (!mh_out && !hll_out) return SS_ERR_INVALID_ARG;
    if (mh_out && (P <= 0 || (P & 3) || P > 2048)) return SS_ERR_INVALID_ARG;
    if (hll_out && (M < 16 || (M & (M - 1)) || M > 65536)) return SS_ERR_INVALID_ARG;
    if (N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if ((graph->hub_rows == nullptr) != (graph->hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    ss_hll_params p0 = {};
    if (cards_out) {
        if (!hll_out) return SS_ERR_INVALID_ARG;
        const int rc = check_params(prm);
        if (rc != SS_OK) return rc;
        if ((1 << prm->p) != M) return SS_ERR_INVALID_ARG;
        p0 = *prm;
    }
    const GraphArgs g = to_args(*graph);
    if (g.rows() == 0) return SS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool hubs = g.hub_rows && g.hub_count;
    const int lead = hub_lead_blocks(hubs);
    if (!mh_out && M == 256) {  // HLL alone: 4 destinations per wavefront
        const int rc = launch_hll_hop(g, hll_in, hll_out, cards_out, cards_stride, p0, hubs, lead, s);
        if (rc != SS_OK || lead > 0) return rc;
        return launch_propagate_hub_only(g, nullptr, nullptr, hll_in, hll_out, cards_out, cards_stride, p0, s);
    }
    if (mh_out && hll_out && P == 128 && M == 256) {
        // both sketches: one launch per sketch is faster than the two-sketch kernel (192 + 111 us vs 326 us on the bench
        // graph): the HLL kernel keeps 4 destinations in flight per wavefront, the MinHash kernel one; each launch hosts the hub
        // units of its own sketch
        int rc = launch_hll_hop(g, hll_in, hll_out, cards_out, cards_stride, p0, hubs, lead, s);
        if (rc != SS_OK) return rc;
        rc = launch_minhash_hop(g, mh_in, mh_out, hubs, lead, nullptr, nullptr, nullptr, 0, p0, s);
        if (rc != SS_OK || lead > 0) return rc;
        return launch_propagate_hub_only(g, mh_in, mh_out, hll_in, hll_out, cards_out, cards_stride, p0, s);
    }
    // the fast path needs both sketches (or the absent one's size irrelevant): P == 128 and M == 256
    const bool fast = (!mh_out || P == 128) && (!hll_out || M == 256);
    if (fast)
        return launch_propagate<128, 256>(g, mh_in, mh_out, 128, hll_in, hll_out, 256, cards_out, cards_stride, p0, s);
    return launch_propagate<0, 0>(g, mh_in, mh_out, P, hll_in, hll_out, M, cards_out, cards_stride, p0, s);
}
