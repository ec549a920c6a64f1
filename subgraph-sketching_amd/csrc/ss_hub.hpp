// ss_hub.hpp -- hub units: the rows the row kernels leave out (in-degree > graph.hub_threshold, listed by ss_csr_build), walked by
// whole workgroups.
//
// A row of tens of thousands of neighbours on one wavefront is tens of thousands of dependent loads (power-law graphs: ogbl-ppa,
// ogbl-citation2).  The CSR build lists such rows -- `hub_rows` (at most SS_MEGA_SLICE neighbours: one unit each) and `mega_rows`
// (more: one unit per slice of SS_MEGA_SLICE neighbours; the partial rows go through mega_scratch and the workgroup that finishes a
// row's LAST slice -- ticket counter -- combines them) -- and a unit is walked by all wavefronts of a workgroup, 64-neighbour chunks
// dealt round robin, partials combined through LDS.
//
// Round 4: the units are served by the LEADING workgroups of a row kernel's own launch (`hub_blocks` of them, 256 threads = 4
// wavefronts each; the row workgroups follow).  As launches of their own (rounds 1-3: 256 workgroups of 16 wavefronts after every
// row kernel) the passes were a serial chain of ~15-40 us per hop that nothing overlapped -- 35 us of a 0.49 ms step at rank^-0.5,
// 113 us of 0.52 ms at rank^-0.9 --, and on a second stream they only started in the row kernel's tail.  Inside the
// launch they start FIRST (workgroups are dispatched in index order) and the row workgroups fill the chip around them.
// Which launch hosts what: the HLL first-hop launch both hop-1 tables' units, the MinHash table-hop launch both tables' units of
// its hop (ss_fused_hop_stage) or its own sketch's (ss_propagate); the fused kernel hosts none -- it has no register to spare.
// The stand-alone kernels remain for the shapes without a specialised row kernel and for A/B runs (SS_HUB_LAUNCHES=1).
//
// Tickets: a mega row has one counter per sketch side (descriptor words 3 / 4), because MinHash units and HLL units of the same
// rows are served by different workgroups of ONE launch, at the same time.
#pragma once
#include <cstdlib>

#include "ss_walks.hpp"

namespace ss {

constexpr int kHubLeadBlocks = 1024;  // leading workgroups of a row launch (four per CU); those beyond the unit count exit at once
constexpr int kHubLeadWaves = 4;
constexpr int kMegaDesc = SS_MEGA_DESC_WORDS;  // int32 words per mega-row descriptor: {row, first slice, slices, ticket (MinHash side), ticket (HLL side), 0, 0, 0}

// host: how many leading workgroups a row launch gets (0: the stand-alone hub kernels follow the launch instead)
void note_hub_call();  // ss_debug.hip: ss_debug_hub_calls counts the library calls that served hub units
inline int hub_lead_blocks(bool hubs)
{
    if (hubs) note_hub_call();
    static const bool launches = getenv("SS_HUB_LAUNCHES") && atoi(getenv("SS_HUB_LAUNCHES")) != 0;
    static const int n = getenv("SS_HUB_LEAD_BLOCKS") ? atoi(getenv("SS_HUB_LEAD_BLOCKS")) : kHubLeadBlocks;
    // (at least 4: the launches split the leading workgroups by integer division -- a quarter of them for the hop-1 HLL units, half
    // for the MinHash units -- and every share must hold a workgroup)
    return hubs && !launches ? (n >= 4 && n <= 65536 ? n : kHubLeadBlocks) : 0;
}

struct HubCounts {
    int hubs, mega, slices;
};
__device__ __forceinline__ HubCounts hub_counts(const GraphArgs &g)
{
    return HubCounts{*g.hub_count, g.mega_count ? g.mega_count[0] : 0, g.mega_count ? g.mega_count[1] : 0};
}

// the mega row global slice `gs` belongs to: every thread looks at some descriptors (a per-row loop over all mega rows cost every
// workgroup three dependent loads per ROW, slices or not).  Called by all threads; ends with a barrier.
template <int THREADS>
__device__ __forceinline__ int mega_row_of_slice(const GraphArgs &g, int gs, int n_mega, int &s_m)
{
    for (int t = threadIdx.x; t < n_mega; t += THREADS) {
        const int4 d = *reinterpret_cast<const int4 *>(g.mega_rows + (int64_t)kMegaDesc * t);
        if (gs >= d.y && gs < d.y + d.z) s_m = t;
    }
    __syncthreads();
    return s_m;
}

// ---- table hops (P = 128, M = 256): min / byte-wise max over the neighbours' rows of the previous hop ------------------------
template <int WAVES>
struct TableHubLds {
    u32x4 part_mh[WAVES][32];
    u32x4 part_hll[WAVES][16];
    int s_last, s_m;
};

// worker / n_workers: this workgroup's place among the workgroups that serve units (units are dealt round robin: hub row h to
// worker h % n_workers, global slice gs to (hubs + gs) % n_workers)
template <int WAVES, bool DO_MH, bool DO_HLL>
__device__ __forceinline__ void table_hub_units(const GraphArgs &g, int worker, int n_workers, const uint32_t *__restrict__ mh_in,
                                                uint32_t *__restrict__ mh_out, const uint8_t *__restrict__ hll_in,
                                                uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride,
                                                const EstimatorTables &est, bool want_cards, TableHubLds<WAVES> &s)
{
    constexpr int P = 128, M = 256, CM = 32, CH = 16, THREADS = WAVES * kWave;
    constexpr int kTicket = DO_MH ? 3 : 4;
    const HubCounts n = hub_counts(g);
    if (worker >= n.hubs + n.slices) return;  // the common case (few or no hub rows) costs three scalar loads per workgroup
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;

    // all waves walk the neighbours t in [lo, hi) of row i; wave 0 ends up with the combined partial rows
    // (MinHash chunk `lane` in lanes 0..31, HLL chunk `lane - 32` in lanes 32..47)
    // (wave w takes the CONTIGUOUS 64-neighbour chunks w, w + WAVES, ... of [lo, hi): ids by one coalesced load per chunk, handed
    // out with v_readlane (MinHash) / DPP row broadcasts (HLL: 16 neighbours per lane group) -- one round trip per chunk instead
    // of the generic walk's two per batch of four)
    auto walk = [&](int64_t i, const int32_t *nb, int deg, int lo, int hi, u32x4 &mh_acc, u32x4 &hll_acc) {
        if constexpr (DO_MH) {
            const int sg = lane >> 5, c = lane & 31;
            u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (int b = lo + wave * kWave; b < hi; b += WAVES * kWave)  // wave-uniform
                acc = min4(acc, minhash_chunk64(mh_in, nb + b, deg - b, hi - b < kWave ? hi - b : kWave, i, lane));
            acc = min4(acc, shfl_xor4(acc, 32));
            if (sg == 0) s.part_mh[wave][c] = acc;
        }
        if constexpr (DO_HLL) {
            const int sg = lane >> 4, c = lane & 15;
            u32x4 acc = {0u, 0u, 0u, 0u};
            for (int b = lo + wave * kWave; b < hi; b += WAVES * kWave) {
                const int bg = b + kRow * sg;  // this lane group's 16 neighbours of the chunk
                acc = bytemax16(acc, hll_walk_first16(hll_in, nb + bg, deg - bg, hi - bg, i, c));
            }
            acc = bytemax16(acc, shfl_xor4(acc, 16));
            acc = bytemax16(acc, shfl_xor4(acc, 32));
            if (sg == 0) s.part_hll[wave][c] = acc;
        }
        __syncthreads();
        if (wave == 0) {
            if (DO_MH && lane < CM) {
                mh_acc = s.part_mh[0][lane];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) mh_acc = min4(mh_acc, s.part_mh[w][lane]);
            }
            if (DO_HLL && lane >= 32 && lane < 32 + CH) {
                hll_acc = s.part_hll[0][lane - 32];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) hll_acc = bytemax16(hll_acc, s.part_hll[w][lane - 32]);
            }
        }
    };
    // wave 0 stores the finished row (+ its cardinality)
    auto finish = [&](int64_t i, u32x4 mh_acc, u32x4 hll_acc) {
        if (DO_MH && lane < CM) {
            *reinterpret_cast<u32x4 *>(mh_out + i * P + 4 * lane) = mh_acc;
            mirror_mh4(g.mir, i * P + 4 * lane, mh_acc);
        }
        if (DO_HLL && lane >= 32 && lane < 32 + CH) {  // lanes 32..47 = one DPP row
            const int c = lane - 32;
            *reinterpret_cast<u32x4 *>(hll_out + i * M + 16 * c) = hll_acc;
            mirror_hll16(g.mir, i * M + 16 * c, hll_acc);
            if (want_cards) {
                int nonzero = 0;
                float hsum = 0.0f;
                hll_dword_stats(hll_acc.x, nonzero, hsum);
                hll_dword_stats(hll_acc.y, nonzero, hsum);
                hll_dword_stats(hll_acc.z, nonzero, hsum);
                hll_dword_stats(hll_acc.w, nonzero, hsum);
                nonzero = row16_sum_i(nonzero);
                hsum = row16_sum_f(hsum);
                if (c == 0) {
                    const float card = hll_estimate(est, M - nonzero, hsum);
                    cards_out[i * cards_stride] = card;
                    mirror_card(g.mir, i * cards_stride, card);
                }
            }
        }
    };

    for (int h = worker; h < n.hubs; h += n_workers) {
        const int64_t i = g.hub_rows[h];
        if (!g.owns(i)) continue;  // workgroup-uniform
        const int64_t rb = g.rowptr[i];
        const int deg = (int)(g.rowptr[i + 1] - rb);
        const int total = deg + (i < n_self ? 1 : 0);
        u32x4 mh_acc = {0u, 0u, 0u, 0u}, hll_acc = {0u, 0u, 0u, 0u};
        walk(i, g.col + rb, deg, 0, total, mh_acc, hll_acc);
        if (wave == 0) finish(i, mh_acc, hll_acc);
        __syncthreads();
    }

    // ---- mega rows: the slices of ALL mega rows form one list (slice s of the row with first slice f is global slice f + s) that
    // continues the round robin of the hub rows above.  A row with a million neighbours is spread over the whole chip instead of
    // being one workgroup's serial walk.
    for (int gs = (int)((unsigned)(worker + n_workers - n.hubs % n_workers) % (unsigned)n_workers); gs < n.slices; gs += n_workers) {
        const int m = mega_row_of_slice<THREADS>(g, gs, n.mega, s.s_m);
        int32_t *desc = g.mega_rows + (int64_t)kMegaDesc * m;
        const int4 e = *reinterpret_cast<const int4 *>(desc);  // {row, first slice, slices, -}
        const int64_t i = e.x;
        if (!g.owns(i)) { __syncthreads(); continue; }  // workgroup-uniform
        const int64_t rb = g.rowptr[i];
        const int deg = (int)(g.rowptr[i + 1] - rb);
        const int total = deg + (i < n_self ? 1 : 0);
        const int sl = gs - e.y;
        const int lo = sl * SS_MEGA_SLICE < total ? sl * SS_MEGA_SLICE : total;
        const int hi = lo + SS_MEGA_SLICE < total ? lo + SS_MEGA_SLICE : total;
        u32x4 mh_acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hll_acc = {0u, 0u, 0u, 0u};
        walk(i, g.col + rb, deg, lo, hi, mh_acc, hll_acc);
        uint8_t *mine = g.mega_scratch + (int64_t)(e.y + sl) * kMegaSlot;
        if (wave == 0) {
            if (DO_MH && lane < CM) coherent_store4(mine + 16 * lane, mh_acc);
            if (DO_HLL && lane >= 32 && lane < 32 + CH) coherent_store4(mine + kMegaHllOffset + 16 * (lane - 32), hll_acc);
        }
        publish_drain();  // every wave: the slot stores are acknowledged before the barrier that precedes the ticket
        __syncthreads();
        if (threadIdx.x == 0) {
            const int prev = take_ticket(desc + kTicket);
            s.s_last = prev == e.z - 1;
            if (s.s_last) reset_ticket(desc + kTicket);  // every slice has arrived: ready for the next hop
        }
        __syncthreads();
        if (s.s_last) {  // workgroup-uniform.  All waves read the slots (wave w: slots w, w + WAVES, ...: a row of 70 000 neighbours has
                         // 69 of them), wave 0 combines and stores
            u32x4 mh_all = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hll_all = {0u, 0u, 0u, 0u};
            constexpr int kSlotsInFlight = WAVES >= 16 ? 1 : 2;  // (more would cost the hosting row kernels registers they do not have)
            for (int q = wave; q < e.z; q += kSlotsInFlight * WAVES) {
                u32x4 v[kSlotsInFlight];
#pragma unroll
                for (int k = 0; k < kSlotsInFlight; ++k) {
                    const int qq = q + k * WAVES < e.z ? q + k * WAVES : q;  // (past the end: slot q again -- min / max do not mind)
                    const uint8_t *part = g.mega_scratch + (int64_t)(e.y + qq) * kMegaSlot;
                    v[k] = u32x4{0u, 0u, 0u, 0u};
                    if (DO_MH && lane < CM) v[k] = coherent_load4(part + 16 * lane);
                    if (DO_HLL && lane >= 32 && lane < 32 + CH) v[k] = coherent_load4(part + kMegaHllOffset + 16 * (lane - 32));
                }
#pragma unroll
                for (int k = 0; k < kSlotsInFlight; ++k) {
                    if (DO_MH && lane < CM) mh_all = min4(mh_all, v[k]);
                    if (DO_HLL && lane >= 32 && lane < 32 + CH) hll_all = bytemax16(hll_all, v[k]);
                }
            }
            if (DO_MH && lane < CM) s.part_mh[wave][lane] = mh_all;
            if (DO_HLL && lane >= 32 && lane < 32 + CH) s.part_hll[wave][lane - 32] = hll_all;
            __syncthreads();
            if (wave == 0) {
                if (DO_MH && lane < CM) {
#pragma unroll
                    for (int w = 1; w < WAVES; ++w) mh_all = min4(mh_all, s.part_mh[w][lane]);
                }
                if (DO_HLL && lane >= 32 && lane < 32 + CH) {
#pragma unroll
                    for (int w = 1; w < WAVES; ++w) hll_all = bytemax16(hll_all, s.part_hll[w][lane - 32]);
                }
                finish(i, mh_all, hll_all);
            }
        }
        __syncthreads();
    }
}

// ---- first hop (hop 1 from node ids, ss_first_hop.hip): P = 64 * PPL, M = 256 ---------------------------------------------------
template <int PPL>
struct FirstHopHubLds {
    __attribute__((aligned(16))) uint32_t hll_row[256];
    uint32_t mh_row[PPL * kWave];
    int s_last, s_m;
};

template <int PPL, int WAVES, bool DO_MH, bool DO_HLL>
__device__ __forceinline__ void first_hop_hub_units(const GraphArgs &g, int worker, int n_workers, const uint64_t *__restrict__ pa,
                                                    const uint64_t *__restrict__ pb, uint32_t *__restrict__ mh_out, int p,
                                                    uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride,
                                                    const EstimatorTables &est, bool want_cards_in, FirstHopHubLds<PPL> &s, bool force_exact)
{
    constexpr int P = PPL * kWave, THREADS = WAVES * kWave;
    constexpr int kTicket = DO_MH ? 3 : 4;
    const HubCounts n = hub_counts(g);
    if (worker >= n.hubs + n.slices) return;
    const bool want_cards = DO_HLL && want_cards_in;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    uint64_t a[PPL], b[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        a[q] = DO_MH ? pa[lane + kWave * q] : 0ULL;
        b[q] = DO_MH ? pb[lane + kWave * q] : 0ULL;
    }
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;

    // all waves hash the neighbours t in [lo, hi) of row i (batches of 64, one batch per wave and step); the combined partial rows
    // are left in s.mh_row / s.hll_row
    auto walk = [&](int64_t i, const int32_t *nb, int deg, int lo, int hi) {
        for (int t = threadIdx.x; t < 256; t += THREADS) s.hll_row[t] = 0u;
        for (int t = threadIdx.x; t < P; t += THREADS) s.mh_row[t] = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t acc[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) acc[q] = 0xFFFFFFFFu;
        // slice = the same walk over nb + lo with the degree counted from lo (slot deg - lo is the implicit self loop)
        if (DO_HLL) first_hop_walk<PPL, false, true>(nb + lo, deg - lo, hi - lo, i, wave, WAVES, p, a, b, acc, s.hll_row, lane);
        // MinHash: the two-phase walk (ss_walks.hpp: 4-5 instead of 10 VALU per neighbour and permutation) over this wavefront's
        // batches; a wavefront whose share is ambiguous (duplicated minimum, key collision) redoes its share exactly
        if (DO_MH && wave * kWave < hi - lo) {  // (wave-uniform) the wavefront has at least one batch
            const bool amb = force_exact || first_hop_minhash_fast<PPL>(nb + lo, deg - lo, hi - lo, i, a, b, acc, lane, wave, WAVES);
            if (__any(amb)) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) acc[q] = 0xFFFFFFFFu;
                first_hop_walk<PPL, true, false>(nb + lo, deg - lo, hi - lo, i, wave, WAVES, p, a, b, acc, s.hll_row, lane);
            }
        }
        if (DO_MH) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) atomicMin(&s.mh_row[lane + kWave * q], acc[q]);
        }
        __syncthreads();
    };
    // wave 0 stores the finished row (+ its cardinality): lane l holds MinHash values l, l + 64, .. and HLL registers 4l .. 4l+3
    auto finish = [&](int64_t i, const uint32_t (&mh)[PPL], uint32_t regs) {
        if (DO_MH) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                mh_out[i * P + lane + kWave * q] = mh[q];
                mirror_mh1(g.mir, i * P + lane + kWave * q, mh[q]);
            }
        }
        if (DO_HLL) {
            *reinterpret_cast<uint32_t *>(hll_out + i * 256 + 4 * lane) = regs;
            mirror_hll4(g.mir, i * 256 + 4 * lane, regs);
        }
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
    };

    for (int h = worker; h < n.hubs; h += n_workers) {
        const int64_t i = g.hub_rows[h];
        if (!g.owns(i)) continue;  // workgroup-uniform
        const int64_t rb = g.rowptr[i];
        const int deg = (int)(g.rowptr[i + 1] - rb);
        const int total = deg + (i < n_self ? 1 : 0);
        walk(i, g.col + rb, deg, 0, total);
        if (wave == 0) {
            uint32_t mh[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) mh[q] = DO_MH ? s.mh_row[lane + kWave * q] : 0u;
            finish(i, mh, DO_HLL ? pack_hll_quad(s.hll_row, lane) : 0u);
        }
        __syncthreads();
    }

    // ---- mega rows (see table_hub_units); scratch layout per slice: MinHash u32[P] (P <= 256: the first 1024 B) then the packed
    // HLL row at kMegaHllOffset
    for (int gs = (int)((unsigned)(worker + n_workers - n.hubs % n_workers) % (unsigned)n_workers); gs < n.slices; gs += n_workers) {
        const int m = mega_row_of_slice<THREADS>(g, gs, n.mega, s.s_m);
        int32_t *desc = g.mega_rows + (int64_t)kMegaDesc * m;
        const int4 e = *reinterpret_cast<const int4 *>(desc);
        const int64_t i = e.x;
        if (!g.owns(i)) { __syncthreads(); continue; }  // workgroup-uniform
        const int64_t rb = g.rowptr[i];
        const int deg = (int)(g.rowptr[i + 1] - rb);
        const int total = deg + (i < n_self ? 1 : 0);
        const int sl = gs - e.y;
        const int lo = sl * SS_MEGA_SLICE < total ? sl * SS_MEGA_SLICE : total;
        const int hi = lo + SS_MEGA_SLICE < total ? lo + SS_MEGA_SLICE : total;
        walk(i, g.col + rb, deg, lo, hi);
        uint8_t *mine = g.mega_scratch + (int64_t)(e.y + sl) * kMegaSlot;
        if (wave == 0) {
            if (DO_MH) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) coherent_store(reinterpret_cast<uint32_t *>(mine) + lane + kWave * q, s.mh_row[lane + kWave * q]);
            }
            if (DO_HLL) coherent_store(reinterpret_cast<uint32_t *>(mine + kMegaHllOffset) + lane, pack_hll_quad(s.hll_row, lane));
        }
        publish_drain();  // every wave: the slot stores are acknowledged before the barrier that precedes the ticket
        __syncthreads();
        if (threadIdx.x == 0) {
            const int prev = take_ticket(desc + kTicket);
            s.s_last = prev == e.z - 1;
            if (s.s_last) reset_ticket(desc + kTicket);
        }
        __syncthreads();
        if (s.s_last) {  // workgroup-uniform.  All waves read the slots (wave w: slots w, w + WAVES, ...), combined through the
                         // LDS rows of the walk (which the barrier above has released), wave 0 stores
            uint32_t mh[PPL], regs = 0u;
#pragma unroll
            for (int q = 0; q < PPL; ++q) mh[q] = 0xFFFFFFFFu;
            for (int s2 = wave; s2 < e.z; s2 += 2 * WAVES) {
                const int s3 = s2 + WAVES < e.z ? s2 + WAVES : s2;  // two slots requested together (past the end: s2 again)
                const uint8_t *part0 = g.mega_scratch + (int64_t)(e.y + s2) * kMegaSlot;
                const uint8_t *part1 = g.mega_scratch + (int64_t)(e.y + s3) * kMegaSlot;
                uint32_t v0[PPL], v1[PPL], h0 = 0u, h1 = 0u;
                if (DO_MH) {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        v0[q] = coherent_load(reinterpret_cast<const uint32_t *>(part0) + lane + kWave * q);
                        v1[q] = coherent_load(reinterpret_cast<const uint32_t *>(part1) + lane + kWave * q);
                    }
                }
                if (DO_HLL) {
                    h0 = coherent_load(reinterpret_cast<const uint32_t *>(part0 + kMegaHllOffset) + lane);
                    h1 = coherent_load(reinterpret_cast<const uint32_t *>(part1 + kMegaHllOffset) + lane);
                }
                if (DO_MH) {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const uint32_t v = v0[q] < v1[q] ? v0[q] : v1[q];
                        mh[q] = v < mh[q] ? v : mh[q];
                    }
                }
                if (DO_HLL) {
                    const uint32_t v = bytemax4(h0, h1);
                    regs = bytemax4(regs, v);
                }
            }
            for (int t = threadIdx.x; t < 256; t += THREADS) s.hll_row[t] = 0u;
            for (int t = threadIdx.x; t < P; t += THREADS) s.mh_row[t] = 0xFFFFFFFFu;
            __syncthreads();
            if (DO_MH) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) atomicMin(&s.mh_row[lane + kWave * q], mh[q]);
            }
            if (DO_HLL) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicMax(&s.hll_row[4 * lane + k], (regs >> (8 * k)) & 0xFFu);
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) mh[q] = DO_MH ? s.mh_row[lane + kWave * q] : 0u;
                finish(i, mh, DO_HLL ? pack_hll_quad(s.hll_row, lane) : 0u);
            }
        }
        __syncthreads();
    }
}

}  // namespace ss
