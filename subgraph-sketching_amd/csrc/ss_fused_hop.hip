// ss_fused_hop.hip -- the MinHash first hop and the HLL table hop of hop 2 in ONE launch, interleaved inside every wavefront.
//
// Why: of the kernels of a build the MinHash first hop (hop 1 recomputed from node ids, ss_first_hop.hip) is bound by VALU issue
// (VALUBusy 88 %) and the HLL table hop (ss_propagate.hip) by memory (VALUBusy 47 %); they use different pipes but as separate
// kernels they cannot overlap -- on two streams they share the wave slots and each runs at half rate (both are latency-bound by
// occupancy; tools/probe_overlap.py: 385 us against 392 us back to back).  Inside one wavefront they can: the wave POSTS the HLL
// row loads of four of its rows, walks the MinHash first hop of those rows while the loads travel, and only then folds the HLL
// rows -- the memory pipe works under the VALU pipe.
//
// Data dependencies make this the only such pair of a build: hop-2 HLL rows need the COMPLETE hop-1 HLL table (so the HLL first
// hop and its hub pass run before this launch); hop-1 MinHash rows need nothing but the graph.  A build of h >= 2 hops is
//     ss_fused_hop_stage [this file]  ->  ss_propagate for hops 3..h
// where ss_fused_hop_stage = HLL first hop of the regular rows, ONE hub pass for both hop-1 sketches, this kernel, the MinHash
// table hop of hop 2 (propagate_kernel<128,256>) and ONE hub pass for both hop-2 sketches -- as many hub launches as the unfused
// schedule.  (With cards1_out == NULL the hop-1 HLL table is an input -- the deferred first hop of the ELPH call sequence, hashing.py
// DEFER_FIRST_HOP -- and the hop-1 MinHash hub rows get a pass of their own after the kernel.)
// Results are bit-identical to the unfused sequence: the MinHash side is MinhashRows (ss_walks.hpp, shared with
// first_hop_rows_kernel), the HLL side folds the same rows with the same byte-wise max and runs the same cardinality epilogue.
//
// Mapping: a wavefront owns kFusedRows = 4 consecutive destination rows.  MinHash side: as first_hop_rows_kernel (with 4 rows the
// 60 col entries of a batch almost always cover the whole chunk: a batch reload in the middle of the walk would have to wait for
// its ids with vmcnt(0) -- vmcnt retires in order -- and so for every HLL row posted before it).  HLL side: one 16-lane DPP row
// per destination (lane c = 16-byte chunk c of the 256-byte row); the first kHllInFlight = 7 neighbour chunks per lane are
// requested into registers up front and the next kHllLds = 7 into LDS (global_load_lds_dwordx4; ids by one coalesced load per
// lane group, handed out by DPP row_newbcast), the rest of rows longer than 14 after the MinHash walk (hll_row16_finish).  Hub rows are skipped by both sides:
// they are hub units (ss_hub.hpp) hosted by the launches before and after this one -- not by this one: the kernel sits exactly at its
// 96-VGPR budget, and a hub branch, whatever it contained, cost the row path 64 bytes of scratch and 30 us (round 4).
// P = 64 * PPL, M = 256 (p = 8) only -- the shapes ss_first_hop has a kernel for.
#include <cstdlib>

#include "ss_hub.hpp"

namespace ss {

constexpr int kFusedRows = 4;
#ifndef SS_HLL_INFLIGHT
#define SS_HLL_INFLIGHT 7
#define SS_HLL_LDS 7
#endif
constexpr int kHllInFlight = SS_HLL_INFLIGHT;

struct HllPosted {  // one lane group's view of its row while the row's first chunks are in flight
    int64_t i;
    const int32_t *nb;
    int deg, total;
    bool write;
    u32x4 x[kHllInFlight];
};

// requests chunk c of the first min(total, kHllInFlight) neighbour rows (the implicit self loop is neighbour `deg`)
template <int T0>
__device__ __forceinline__ void hll_post(HllPosted &h, const uint8_t *__restrict__ hll_in, int my_nb, int c)
{
    if constexpr (T0 < kHllInFlight) {
        const int nbt = __builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T0, 0xF, 0xF, false);  // row_newbcast: lane T0 of the 16-lane row
        const int64_t j = T0 < h.deg ? (int64_t)nbt : h.i;
        h.x[T0] = u32x4{0u, 0u, 0u, 0u};
        if (T0 < h.total) h.x[T0] = *reinterpret_cast<const u32x4 *>(hll_in + j * 256 + 16 * c);
        hll_post<T0 + 1>(h, hll_in, my_nb, c);
    }
}

// neighbour rows kHllInFlight .. kHllInFlight + kHllLds - 1 of every lane group travel into LDS instead of registers
// (global_load_lds_dwordx4: no VGPR destination; one instruction = one neighbour of each of the wavefront's four rows, landing as
// 64 lanes x 16 B = 1 KiB at the wave-uniform LDS address): rows of up to 16 neighbours need no load after the MinHash walk
constexpr int kHllLds = SS_HLL_LDS;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) const uint32_t global_u32;

template <int T>
__device__ __forceinline__ void hll_post_lds(const HllPosted &h, const uint8_t *__restrict__ hll_in, int my_nb, int c, uint32_t lds_wave)
{
    if constexpr (T < kHllInFlight + kHllLds) {
        const int nbt = __builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T, 0xF, 0xF, false);
        const int64_t j = T < h.deg ? (int64_t)nbt : h.i;
        if (T < h.total)
            __builtin_amdgcn_global_load_lds((global_u32 *)(uintptr_t)(hll_in + j * 256 + 16 * c),
                                             (lds_u32 *)(uintptr_t)(lds_wave + 1024u * (T - kHllInFlight)), 16, 0, 0);
        hll_post_lds<T + 1>(h, hll_in, my_nb, c, lds_wave);
    }
}

__device__ __forceinline__ void hll_fold(const HllPosted &h, u32x4 &ae, u32x4 &ao)
{
#pragma unroll
    for (int k = 0; k < kHllInFlight; ++k) hll_acc(ae, ao, h.x[k]);
}

// ---- the kernel: persistent, software-pipelined -------------------------------------------------------------------------------
// Four or five wavefronts per SIMD are too few to hide the three dependent round trips at the head of a chunk (row
// bounds -> neighbour ids -> HLL rows): a one-chunk-per-wavefront form of this kernel gained 6 % over the two separate launches
// (174 us against 184).  Here a wavefront keeps walking chunks (chunk = kFusedRows rows; chunk q, q + waves, ...) and the loads of the NEXT chunks are posted
// before the MinHash walk of the current one:
//     ids(k) arrive  ->  post HLL rows(k)  ->  post ids(k+1) [bounds(k+1) arrived an iteration ago]  ->  post bounds(k+2)
//     ->  MinHash walk(k)  [VALU; everything above travels]  ->  fold HLL rows(k), finish, store
// vmcnt retires in order and that is exactly the order of use, so no wait ever covers a younger load.
// Measured (bench graph): two separate launches 82 + 102 = 184 us; one chunk per wavefront 174 us; this kernel 166-170 us with 12
// chunks in registers and nothing in LDS (step 0.494 -> 0.478 ms), 151 us as it stands.  The VALU work alone would be
// ~100-120 us: what is left are the loads of rows with more than 16 neighbours, issued and awaited after the walk.  Register-only
// attempts at more coverage: a rolling window (fold four posted chunks after every MinHash row and re-post their registers with the
// row's next four neighbours) covered 24 neighbours but cost 157-167 VGPRs (three wavefronts per SIMD): 181-196 us; a single re-post of
// 8 chunks before the last MinHash row (coverage 16, 134 VGPRs): 184 us.  Not shipped: the kernel lives on its occupancy.
// Round 3: 11 register + 5 LDS landings at ~125 VGPRs (four wavefronts per SIMD) was 151-152 us; held to 96 VGPRs by
// amdgpu_waves_per_eu (no scratch) with 7 + 7 landings and a 3 KB estimator image (31 KB of LDS: five workgroups per CU) it is
// 146 us -- and 1 946 -> 1 730 / 3 650 -> 3 410 us at ppa / citation2 size, where the table hop's gathers come from HBM and a fifth
// wavefront hides more of them.  (6 + 7, 7 + 6: the same within noise; 4 + 6: 154-158 us -- coverage still matters; six wavefronts spill.)
// A second use of the register landings -- folded before the LAST MinHash row and sent out again for neighbours 14 .. 20 (ids from a
// second id word per lane), coverage 21 -- needs the accumulators alive through that row's walk: 96 / 64 / 32 bytes of scratch with
// 7 / 6 / 5 register landings and 197 / 177 / 154 us; with 4 (no scratch, coverage 15) 146 us, the same as without.  Not shipped.
template <int PPL>
__device__ __forceinline__ void fused_hop_body(const GraphArgs &g, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb,
                                               uint32_t *__restrict__ mh_out, int p, const uint8_t *__restrict__ hll_in,
                                               uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride,
                                               const ss_hll_params &prm, bool skip_hubs)
{
    constexpr int R = kFusedRows, kNb = kWave - R;
    // 3 KB of estimator tables (the kernel is p = 8 only: 257 linear-counting entries; raw / bias up to 256 entries, longer tables stay
    // in global memory) + 7 KB of landings per wavefront = 31 KB: five workgroups per CU
    __shared__ CompactEstimatorLds<257, 256> lds;
    __shared__ __attribute__((aligned(16))) uint8_t landing[256 / kWave][kHllLds > 0 ? kHllLds : 1][1024];
    const bool want_cards = cards_out != nullptr;
    EstimatorTables est = {};
    if (want_cards) est = stage_tables_compact(lds, prm);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t lds_wave = (uint32_t)(uintptr_t)&landing[wave][0][0];  // LDS byte address (low half of the generic pointer)
    const int64_t n_chunks = (g.rows() + R - 1) / R;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x / kWave);
    int64_t chunk = (int64_t)blockIdx.x * (blockDim.x / kWave) + wave;
    if (chunk >= n_chunks) return;
    MinhashRows<PPL, R> m;
    m.setup(g, pa, pb, p, skip_hubs);
    const int lane = m.lane, grp = lane >> 4, c = lane & (kRow - 1);

    auto chunk_rows = [&](int64_t q) -> int { return q < n_chunks ? (int)(g.row1 - (g.row0 + q * R) < R ? g.row1 - (g.row0 + q * R) : R) : 0; };
    auto load_bounds = [&](int64_t q) -> int64_t {  // lane l: rowptr[first row of chunk q + l]
        const int nr = chunk_rows(q);
        return (nr > 0 && lane <= nr) ? g.rowptr[g.row0 + q * R + lane] : 0;
    };
    // what a lane fetches for a chunk whose bounds have arrived: its batch id (MinHash side) and its lane group's neighbour id (HLL side)
    // (nid stays 32 bits wide until it is used: widening it here would be a use of the load's result, and hipcc would wait
    // -- vmcnt(0): for every HLL row posted just before -- right behind the load instead of after the MinHash walk)
    struct Ids { int nid; int my_nb; };
    auto load_ids = [&](int64_t q, int64_t rp) -> Ids {
        const int nr = chunk_rows(q);
        if (nr == 0) return Ids{0, 0};
        const int64_t first = g.row0 + q * R;
        const int64_t c_lo = ((int64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)rp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rp);
        const int rel = (int)(rp - c_lo);
        const int c_n = __builtin_amdgcn_readlane(rel, nr);
        const int32_t *nb = g.col + c_lo;
        Ids out;
        out.nid = lane >= kNb ? (int)(first + (lane - kNb)) : 0;  // node ids are < 2^31 (checked by the entry point)
        if (lane < kNb && lane < c_n) out.nid = nb[lane];
        const bool ok = grp < nr;
        const int rel0 = __shfl(rel, ok ? grp : 0), rel1 = __shfl(rel, ok ? grp + 1 : 0);
        out.my_nb = (ok && c < rel1 - rel0) ? nb[rel0 + c] : 0;
        return out;
    };

    // prologue: bounds of the first two chunks, ids of the first
    int64_t rp_cur = load_bounds(chunk);
    int64_t rp_next = load_bounds(chunk + stride);
    Ids ids_cur = load_ids(chunk, rp_cur);

    for (; chunk < n_chunks; chunk += stride) {  // wave-uniform
        const int64_t first = g.row0 + chunk * R;
        m.begin(g, first, chunk_rows(chunk), rp_cur);
        m.set_batch((int64_t)ids_cur.nid);
        // HLL side of this chunk: the row of this lane group
        HllPosted h;
        const bool ok = grp < m.rows;
        const int rel0 = __shfl(m.rel, ok ? grp : 0), rel1 = __shfl(m.rel, ok ? grp + 1 : 0);
        h.i = m.i0 + (ok ? grp : 0);
        h.nb = m.nb + rel0;
        h.deg = ok ? rel1 - rel0 : 0;
        const bool hub = skip_hubs && h.deg > g.hub_threshold;
        h.write = ok && !hub;
        h.total = h.write ? h.deg + (h.i < m.n_self ? 1 : 0) : 0;
        if constexpr (!(SS_FUSED_ABLATE & 2)) {
        hll_post<0>(h, hll_in, ids_cur.my_nb, c);              // HLL rows of chunk k
        hll_post_lds<kHllInFlight>(h, hll_in, ids_cur.my_nb, c, lds_wave);
        }
        const Ids ids_next = load_ids(chunk + stride, rp_next); // ids of chunk k + 1 (its bounds arrived during the last walk)
        const int64_t rp_after = load_bounds(chunk + 2 * stride);  // bounds of chunk k + 2
        if constexpr (!(SS_FUSED_ABLATE & 1)) {
#pragma unroll 1
        for (int r = 0; r < m.rows; ++r) m.row(r, mh_out);       // MinHash first hop of chunk k: VALU work under all of the above
        }
        if constexpr (SS_FUSED_ABLATE & 2) {
            rp_cur = rp_next;
            rp_next = rp_after;
            ids_cur = ids_next;
            continue;
        }
        u32x4 ae = {0u, 0u, 0u, 0u}, ao = {0u, 0u, 0u, 0u};
        hll_fold(h, ae, ao);
        if constexpr (kHllLds > 0) {
            // (hll_fold waited for h.x; the LDS landings were issued after them and vmcnt retires in order -- but hipcc does not
            // count a global_load_lds against the ds_read below, so the wait is spelled out; everything posted before the walk
            // is long back by now)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < kHllLds; ++k) {
                u32x4 v = *reinterpret_cast<const u32x4 *>(&landing[wave][k][16 * lane]);
                if (!(kHllInFlight + k < h.total)) v = u32x4{0u, 0u, 0u, 0u};  // slot not written this round (stale bytes)
                hll_acc(ae, ao, v);
            }
        }
        const u32x4 acc = hll_acc_result(ae, ao);
        hll_row16_finish<false>(h.i, h.write, h.nb, h.deg, h.total, acc, kHllInFlight + kHllLds, hll_in, hll_out, cards_out, cards_stride, est, want_cards, c, Mirrors{});  // (the fused stage is the unsharded build's: no peers)
        rp_cur = rp_next;
        rp_next = rp_after;
        ids_cur = ids_next;
    }
}

// P = 64 / 128: the register allocator is held to 96 VGPRs (amdgpu_waves_per_eu: at least FIVE wavefronts per SIMD; it gets there
// without scratch -- left alone it spreads to ~125 and four wavefronts).  P = 192 / 256 would spill at 96 and keep the default budget.
template <int PPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PPL <= 2 ? 5 : 1))) void fused_hop_persistent_kernel(
    GraphArgs g, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb, uint32_t *__restrict__ mh_out, int p,
    const uint8_t *__restrict__ hll_in, uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm,
    bool skip_hubs)
{
    fused_hop_body<PPL>(g, pa, pb, mh_out, p, hll_in, hll_out, cards_out, cards_stride, prm, skip_hubs);
}

}  // namespace ss

extern "C" int ss_fused_hop_stage(const ss_csr_graph *graph, const uint64_t *a, const uint64_t *b, int32_t P, uint32_t *mh1_out,
                                  uint32_t *mh2_out, int32_t p, uint8_t *hll1, float *cards1_out, uint8_t *hll2_out, float *cards2_out,
                                  int64_t cards_stride, const ss_hll_params *prm, void *stream)
{
    const uint8_t *hll1_in = hll1;
    using namespace ss;
    if (!graph || graph->num_nodes < 0 || !graph->rowptr) return SS_ERR_INVALID_ARG;
    if (p != 8 || P <= 0 || P % kWave || P > 256) return SS_ERR_UNSUPPORTED;  // caller uses ss_first_hop + ss_propagate
    if (mh2_out && P != 128) return SS_ERR_UNSUPPORTED;                        // the hub pass of the table hop is P = 128 only
    const int64_t N = graph->num_nodes;
    if (N == 0) return SS_OK;
    if (!mh1_out || !a || !b || !hll1_in || !hll2_out || N >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if ((graph->hub_rows == nullptr) != (graph->hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    ss_hll_params p0 = {};
    if (cards2_out) {
        const int rc = check_params(prm);
        if (rc != SS_OK) return rc;
        if (prm->p != p) return SS_ERR_INVALID_ARG;
        p0 = *prm;
    }
    if (!row_range_ok(*graph)) return SS_ERR_INVALID_ARG;
    const GraphArgs g = to_args(*graph);
    if (g.rows() == 0) return SS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool hubs = g.hub_rows && g.hub_count;
    // hub units: `lead` leading workgroups of the HLL first-hop launch serve both hop-1 tables, of the MinHash table hop both hop-2
    // tables (0: launches of their own, as in rounds 1-3)
    const int lead = hub_lead_blocks(hubs);
    const bool own_hop1 = cards1_out != nullptr;  // the hop-1 HLL table is computed here as well (build_hash_tables)
    if (own_hop1) {
        if (!cards2_out) return SS_ERR_INVALID_ARG;
        int rc1 = launch_hll_first_hop_rows(g, p, hll1, cards1_out, cards_stride, p0, hubs, lead, a, b, lead > 0 ? mh1_out : nullptr, P, s);
        if (rc1 != SS_OK) return rc1;
        // ONE hub pass from node ids for both hop-1 sketches: the MinHash hub rows are not needed before the table hop below,
        // but computing them here saves the hub launch after the fused kernel
        if (lead == 0) rc1 = launch_first_hop_hub_only(g, a, b, P, mh1_out, p, hll1, cards1_out, cards_stride, p0, s);
        if (rc1 != SS_OK) return rc1;
    }
    constexpr int rows_per_block = 4 * kFusedRows;
    const unsigned blocks = (unsigned)((g.rows() + rows_per_block - 1) / rows_per_block);
    // 5 workgroups are resident per CU; several times that many balance the tail.  Bench graph (58 blocks per CU): 12 / 15 / 20 / 25 / 30 /
    // 40 per CU: 148.7 / 147.8 / 146.2-146.9 / 149.9 / 149.2 / 151.2 us.  ppa / citation2 size (140 / 715 blocks per CU): 15 / 20 / 30 /
    // 48 / 64 / 100 / 200 / all: 1 832 / 1 819 / 1 774 / 1 724 / 1 738 / 1 728 / 1 751 / 1 748 us and 3 540 / 3 499 / 3 448 / 3 418 / 3 408 /
    // 3 419 / 3 457 / 3 814 us
    static const int wg_per_cu_env = getenv("SS_FUSED_WG_PER_CU") ? atoi(getenv("SS_FUSED_WG_PER_CU")) : 0;
    const int wg_per_cu = (wg_per_cu_env > 0 && wg_per_cu_env <= 4096) ? wg_per_cu_env  // (a bad value falls back to the default)
                          : (blocks > 256u * 128u ? 64 : 20);
    const unsigned grid = blocks < (unsigned)(256 * wg_per_cu) ? blocks : (unsigned)(256 * wg_per_cu);
    {
        ProfileSpan span(s, SS_PROF_FUSED, true);
        switch (P / kWave) {
            case 1: span.launch(fused_hop_persistent_kernel<1>, dim3(grid), dim3(256), g, a, b, mh1_out, p, hll1_in, hll2_out, cards2_out, cards_stride, p0, hubs); break;
            case 2: span.launch(fused_hop_persistent_kernel<2>, dim3(grid), dim3(256), g, a, b, mh1_out, p, hll1_in, hll2_out, cards2_out, cards_stride, p0, hubs); break;
            case 3: span.launch(fused_hop_persistent_kernel<3>, dim3(grid), dim3(256), g, a, b, mh1_out, p, hll1_in, hll2_out, cards2_out, cards_stride, p0, hubs); break;
            default: span.launch(fused_hop_persistent_kernel<4>, dim3(grid), dim3(256), g, a, b, mh1_out, p, hll1_in, hll2_out, cards2_out, cards_stride, p0, hubs); break;
        }
    }
    SS_LAUNCH_CHECK();
    // hub rows of the hop-1 MinHash table (from node ids): they must be in place before anything reads that table
    int rc = own_hop1 ? SS_OK : launch_first_hop_hub_only(g, a, b, P, mh1_out, p, nullptr, nullptr, 0, p0, s);
    if (rc != SS_OK) return rc;
    if (!mh2_out)  // hop-2 HLL hub rows alone
        return launch_propagate_hub_only(g, nullptr, nullptr, hll1_in, hll2_out, cards2_out, cards_stride, p0, s);
    // MinHash table hop of hop 2; its launch hosts the hub units of both hop-2 tables (or ONE hub pass for both follows)
    if (lead > 0) return launch_minhash_hop(g, mh1_out, mh2_out, hubs, lead, hll1_in, hll2_out, cards2_out, cards_stride, p0, s);
    rc = launch_minhash_hop(g, mh1_out, mh2_out, hubs, 0, nullptr, nullptr, nullptr, 0, p0, s);
    if (rc != SS_OK) return rc;
    return launch_propagate_hub_only(g, mh1_out, mh2_out, hll1_in, hll2_out, cards2_out, cards_stride, p0, s);
}
