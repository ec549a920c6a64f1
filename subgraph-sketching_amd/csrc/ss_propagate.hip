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
// dependent 1 KiB loads on one wavefront (power-law graphs: ogbl-ppa, ogbl-citation2).  The row kernel
// skips them and propagate_hub_kernel gives each of them a whole 16-wave workgroup: 32 MinHash / 64 HLL
// neighbours per step, partials combined through LDS.
#include "ss_walks.hpp"

namespace ss {

// TP / TM > 0: compile-time row sizes (fast path); 0: run-time.
template <int TP, int TM>
__global__ __launch_bounds__(256) void propagate_kernel(GraphArgs g, const uint32_t *__restrict__ mh_in, uint32_t *__restrict__ mh_out,
                                                        int P_rt, const uint8_t *__restrict__ hll_in, uint8_t *__restrict__ hll_out,
                                                        int M_rt, float *__restrict__ cards_out, int64_t cards_stride, ss_hll_params prm,
                                                        bool skip_hubs)
{
    __shared__ EstimatorLds lds;
    const bool want_cards = cards_out != nullptr && hll_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);

    const int P = TP ? TP : P_rt;
    const int M = TM ? TM : M_rt;
    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform row id (readfirstlane makes the uniformity visible to the compiler: scalar loads, scalar loop control)
    int64_t i = g.row0 + (int64_t)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
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
    if (skip_hubs && deg > g.hub_threshold) return;  // left to propagate_hub_kernel
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
                if (want_cards) {
                    hll_dword_stats(acc.x, nonzero, hsum);
                    hll_dword_stats(acc.y, nonzero, hsum);
                    hll_dword_stats(acc.z, nonzero, hsum);
                    hll_dword_stats(acc.w, nonzero, hsum);
                }
            }
        }
        if (want_cards) {
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
__global__ __launch_bounds__(256) void hll_propagate_row16_kernel(GraphArgs g, const uint8_t *__restrict__ hll_in,
                                                                  uint8_t *__restrict__ hll_out, float *__restrict__ cards_out,
                                                                  int64_t cards_stride, ss_hll_params prm, bool skip_hubs)
{
    __shared__ EstimatorLds lds;
    const bool want_cards = cards_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);
    // the 16 rows of this workgroup are dealt to its 16 lane groups in DEGREE order: the four rows that share a wavefront
    // then have similar degrees, and the walk of a wavefront lasts as long as its longest row
    __shared__ int s_deg[256 / kRow], s_owner[256 / kRow];
    const int grp = threadIdx.x / kRow, c16 = threadIdx.x & (kRow - 1);
    const int64_t first = g.row0 + (int64_t)blockIdx.x * (blockDim.x / kRow);
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

// ---- hub rows: one 1024-thread workgroup (16 wavefronts) per row; P = 128, M = 256 only -------------------------
constexpr int kHubThreads = 1024;
constexpr int kHubWaves = kHubThreads / kWave;
constexpr int kHubGrid = 256;  // one workgroup per CU; workgroups beyond the hub count exit at once

__global__ __launch_bounds__(kHubThreads) void propagate_hub_kernel(GraphArgs g, const uint32_t *__restrict__ mh_in,
                                                                    uint32_t *__restrict__ mh_out, const uint8_t *__restrict__ hll_in,
                                                                    uint8_t *__restrict__ hll_out, float *__restrict__ cards_out,
                                                                    int64_t cards_stride, ss_hll_params prm)
{
    constexpr int P = 128, M = 256, CM = 32, CH = 16;
    __shared__ EstimatorLds lds;
    __shared__ u32x4 part_mh[kHubWaves][CM];
    __shared__ u32x4 part_hll[kHubWaves][CH];
    __shared__ int s_last, s_m;
    const int n_hubs = *g.hub_count;
    const int n_mega = g.mega_count ? g.mega_count[0] : 0;
    const int n_slices = g.mega_count ? g.mega_count[1] : 0;
    if ((int)blockIdx.x >= n_hubs && n_mega == 0) return;  // the common case (no hub rows) costs two scalar loads per workgroup
    const bool want_cards = cards_out != nullptr && hll_out != nullptr;
    EstimatorTables est;
    if (want_cards) est = stage_tables(lds, prm);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;

    // all 16 waves walk the neighbours t in [lo, hi) of row i; wave 0 ends up with the combined partial rows
    // (MinHash chunk `lane` in lanes 0..31, HLL chunk `lane - 32` in lanes 32..47)
    // (wave w takes the CONTIGUOUS 64-neighbour chunks w, w + 16, ... of [lo, hi): ids by one coalesced load per chunk, handed
    // out with v_readlane (MinHash) / DPP row broadcasts (HLL: 16 neighbours per lane group) -- one round trip per chunk instead
    // of the generic walk's two per batch of four)
    auto walk = [&](int64_t i, const int32_t *nb, int deg, int lo, int hi, u32x4 &mh_acc, u32x4 &hll_acc) {
        if (mh_out) {
            const int sg = lane >> 5, c = lane & 31;
            u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (int b = lo + wave * kWave; b < hi; b += kHubWaves * kWave)  // wave-uniform
                acc = min4(acc, minhash_chunk64(mh_in, nb + b, deg - b, hi - b < kWave ? hi - b : kWave, i, lane));
            acc = min4(acc, shfl_xor4(acc, 32));
            if (sg == 0) part_mh[wave][c] = acc;
        }
        if (hll_out) {
            const int sg = lane >> 4, c = lane & 15;
            u32x4 acc = {0u, 0u, 0u, 0u};
            for (int b = lo + wave * kWave; b < hi; b += kHubWaves * kWave) {
                const int bg = b + kRow * sg;  // this lane group's 16 neighbours of the chunk
                acc = bytemax16(acc, hll_walk_first16(hll_in, nb + bg, deg - bg, hi - bg, i, c));
            }
            acc = bytemax16(acc, shfl_xor4(acc, 16));
            acc = bytemax16(acc, shfl_xor4(acc, 32));
            if (sg == 0) part_hll[wave][c] = acc;
        }
        __syncthreads();
        if (wave == 0) {
            if (mh_out && lane < CM) {
                mh_acc = part_mh[0][lane];
#pragma unroll
                for (int w = 1; w < kHubWaves; ++w) mh_acc = min4(mh_acc, part_mh[w][lane]);
            }
            if (hll_out && lane >= 32 && lane < 32 + CH) {
                hll_acc = part_hll[0][lane - 32];
#pragma unroll
                for (int w = 1; w < kHubWaves; ++w) hll_acc = bytemax16(hll_acc, part_hll[w][lane - 32]);
            }
        }
    };
    // wave 0 stores the finished row (+ its cardinality)
    auto finish = [&](int64_t i, u32x4 mh_acc, u32x4 hll_acc) {
        if (mh_out && lane < CM) {
            *reinterpret_cast<u32x4 *>(mh_out + i * P + 4 * lane) = mh_acc;
            mirror_mh4(g.mir, i * P + 4 * lane, mh_acc);
        }
        if (hll_out && lane >= 32 && lane < 32 + CH) {  // lanes 32..47 = one DPP row
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

    for (int h = blockIdx.x; h < n_hubs; h += gridDim.x) {
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

    // ---- mega rows: every workgroup takes slices of SS_MEGA_SLICE neighbours of every mega row; the partial rows go
    // through mega_scratch and the workgroup that finishes a row's LAST slice (ticket counter) combines them.  A row
    // with a million neighbours is spread over the whole chip instead of being one workgroup's serial walk.
    // The slices of ALL mega rows form one list (slice s of the row with first slice f is global slice f + s) that continues the
    // round robin of the hub rows above: hub row h went to workgroup h % grid, global slice gs goes to (n_hubs + gs) % grid.
    // Whose slice gs is: every thread looks at some descriptors (a per-row loop over all mega rows cost every workgroup three
    // dependent loads per ROW, slices or not).
    for (int gs = (int)((blockIdx.x + gridDim.x - (unsigned)n_hubs % gridDim.x) % gridDim.x); gs < n_slices; gs += gridDim.x) {
        for (int t = threadIdx.x; t < n_mega; t += kHubThreads) {
            const int4 d = reinterpret_cast<const int4 *>(g.mega_rows)[t];
            if (gs >= d.y && gs < d.y + d.z) s_m = t;
        }
        __syncthreads();
        const int m = s_m;
        const int4 e = reinterpret_cast<const int4 *>(g.mega_rows)[m];  // {row, first slice, slices, ticket}
        const int64_t i = e.x;
        if (!g.owns(i)) { __syncthreads(); continue; }  // workgroup-uniform
        const int64_t rb = g.rowptr[i];
        const int deg = (int)(g.rowptr[i + 1] - rb);
        const int total = deg + (i < n_self ? 1 : 0);
        {
            const int sl = gs - e.y;
            const int lo = sl * SS_MEGA_SLICE < total ? sl * SS_MEGA_SLICE : total;
            const int hi = lo + SS_MEGA_SLICE < total ? lo + SS_MEGA_SLICE : total;
            u32x4 mh_acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hll_acc = {0u, 0u, 0u, 0u};
            walk(i, g.col + rb, deg, lo, hi, mh_acc, hll_acc);
            uint8_t *mine = g.mega_scratch + (int64_t)(e.y + sl) * kMegaSlot;
            if (wave == 0) {
                if (mh_out && lane < CM) coherent_store4(mine + 16 * lane, mh_acc);
                if (hll_out && lane >= 32 && lane < 32 + CH) coherent_store4(mine + kMegaHllOffset + 16 * (lane - 32), hll_acc);
            }
            publish_drain();  // every wave: the slot stores are acknowledged before the barrier that precedes the ticket
            __syncthreads();
            if (threadIdx.x == 0) {
                const int prev = take_ticket(&g.mega_rows[4 * m + 3]);
                s_last = prev == e.z - 1;
                if (s_last) reset_ticket(&g.mega_rows[4 * m + 3]);  // every slice has arrived: ready for the next hop
            }
            __syncthreads();
            if (s_last) {  // workgroup-uniform.  All 16 waves read the slots (wave w: slots w, w + 16, ...: a row of 70 000
                           // neighbours has 69 of them, one wave reading them in turn was a 50 us tail), wave 0 combines and stores
                u32x4 mh_all = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hll_all = {0u, 0u, 0u, 0u};
                for (int q = wave; q < e.z; q += kHubWaves) {
                    const uint8_t *part = g.mega_scratch + (int64_t)(e.y + q) * kMegaSlot;
                    if (mh_out && lane < CM) mh_all = min4(mh_all, coherent_load4(part + 16 * lane));
                    if (hll_out && lane >= 32 && lane < 32 + CH)
                        hll_all = bytemax16(hll_all, coherent_load4(part + kMegaHllOffset + 16 * (lane - 32)));
                }
                if (mh_out && lane < CM) part_mh[wave][lane] = mh_all;
                if (hll_out && lane >= 32 && lane < 32 + CH) part_hll[wave][lane - 32] = hll_all;
                __syncthreads();
                if (wave == 0) {
                    if (mh_out && lane < CM) {
#pragma unroll
                        for (int w = 1; w < kHubWaves; ++w) mh_all = min4(mh_all, part_mh[w][lane]);
                    }
                    if (hll_out && lane >= 32 && lane < 32 + CH) {
#pragma unroll
                        for (int w = 1; w < kHubWaves; ++w) hll_all = bytemax16(hll_all, part_hll[w][lane - 32]);
                    }
                    finish(i, mh_all, hll_all);
                }
            }
            __syncthreads();
        }
    }
}

template <int TP, int TM>
int launch_propagate(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, int P, const uint8_t *hll_in, uint8_t *hll_out, int M,
                     float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    const int rows_per_block = 256 / kWave;
    const int64_t blocks = (g.rows() + rows_per_block - 1) / rows_per_block;
    const bool hubs = TP == 128 && TM == 256 && g.hub_rows && g.hub_count;
    {
        ProfileSpan span(stream, mh_out && !hll_out && TP == 128 ? SS_PROF_MINHASH_HOP : SS_PROF_TAGS);  // MinHash table hop
        hipLaunchKernelGGL((propagate_kernel<TP, TM>), dim3((unsigned)blocks), dim3(256), 0, stream, g, mh_in, mh_out, P, hll_in, hll_out,
                           M, cards_out, cards_stride, prm, hubs);
    }
    SS_LAUNCH_CHECK();
    if (hubs) {
        hipLaunchKernelGGL(propagate_hub_kernel, dim3(kHubGrid), dim3(kHubThreads), 0, stream, g, mh_in, mh_out, hll_in, hll_out,
                           cards_out, cards_stride, prm);
        SS_LAUNCH_CHECK();
    }
    return SS_OK;
}

// the MinHash table hop of the regular rows alone (P = 128), for ss_fused_hop_stage
int launch_minhash_hop(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, bool skip_hubs, hipStream_t stream)
{
    ProfileSpan span(stream, SS_PROF_MINHASH_HOP);
    hipLaunchKernelGGL((propagate_kernel<128, 256>), dim3((unsigned)((g.rows() + 3) / 4)), dim3(256), 0, stream, g, mh_in, mh_out, 128,
                       (const uint8_t *)nullptr, (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, ss_hll_params{}, skip_hubs);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int launch_propagate_hub_only(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, const uint8_t *hll_in, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream)
{
    if (!g.hub_rows || !g.hub_count) return SS_OK;
    ProfileSpan span(stream, SS_PROF_HUB);
    hipLaunchKernelGGL(propagate_hub_kernel, dim3(kHubGrid), dim3(kHubThreads), 0, stream, g, mh_in, mh_out, hll_in, hll_out, cards_out,
                       cards_stride, prm);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

}  // namespace ss

// MinHash table hop of the listed rows only (+ the hub pass, which always serves every hub row)
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
    const bool hubs = P == 128 && g.hub_rows && g.hub_count;  // (the hub pass of the table hops is P = 128 only, as in ss_propagate)
    const unsigned blocks = (unsigned)((n_rows + 3) / 4);
    {
        ProfileSpan span(s, SS_PROF_MINHASH_ROWS);
        if (P == 128)
            hipLaunchKernelGGL((propagate_kernel<128, 256>), dim3(blocks), dim3(256), 0, s, g, mh_in, mh_out, 128, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, ss_hll_params{}, hubs);
        else
            hipLaunchKernelGGL((propagate_kernel<0, 0>), dim3(blocks), dim3(256), 0, s, g, mh_in, mh_out, P, (const uint8_t *)nullptr,
                               (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, ss_hll_params{}, false);
    }
    SS_LAUNCH_CHECK();
    if (!hubs) return SS_OK;
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
    const int64_t R = g.rows();
    if (!mh_out && M == 256) {  // HLL alone: 4 destinations per wavefront
        const bool hubs = g.hub_rows && g.hub_count;
        {
            ProfileSpan span((hipStream_t)stream, SS_PROF_HLL_HOP);
            hipLaunchKernelGGL(hll_propagate_row16_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, (hipStream_t)stream, g, hll_in,
                               hll_out, cards_out, cards_stride, p0, hubs);
        }
        SS_LAUNCH_CHECK();
        return launch_propagate_hub_only(g, nullptr, nullptr, hll_in, hll_out, cards_out, cards_stride, p0, (hipStream_t)stream);
    }
    if (mh_out && hll_out && P == 128 && M == 256) {
        // both sketches: one launch per sketch is faster than the two-sketch kernel (192 + 111 us vs 326 us on the bench
        // graph): the HLL kernel keeps 4 destinations in flight per wavefront, the MinHash kernel one; a single hub pass
        // serves both
        const bool hubs = g.hub_rows && g.hub_count;
        {
            ProfileSpan span((hipStream_t)stream, SS_PROF_HLL_HOP);
            hipLaunchKernelGGL(hll_propagate_row16_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, (hipStream_t)stream, g, hll_in,
                               hll_out, cards_out, cards_stride, p0, hubs);
        }
        SS_LAUNCH_CHECK();
        {
            ProfileSpan span((hipStream_t)stream, SS_PROF_MINHASH_HOP);
            hipLaunchKernelGGL((propagate_kernel<128, 256>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, mh_in,
                               mh_out, 128, (const uint8_t *)nullptr, (uint8_t *)nullptr, 256, (float *)nullptr, (int64_t)0, p0, hubs);
        }
        SS_LAUNCH_CHECK();
        return launch_propagate_hub_only(g, mh_in, mh_out, hll_in, hll_out, cards_out, cards_stride, p0, (hipStream_t)stream);
    }
    // the fast path needs both sketches (or the absent one's size irrelevant): P == 128 and M == 256
    const bool fast = (!mh_out || P == 128) && (!hll_out || M == 256);
    if (fast)
        return launch_propagate<128, 256>(g, mh_in, mh_out, 128, hll_in, hll_out, 256, cards_out, cards_stride, p0, (hipStream_t)stream);
    return launch_propagate<0, 0>(g, mh_in, mh_out, P, hll_in, hll_out, M, cards_out, cards_stride, p0, (hipStream_t)stream);
}
