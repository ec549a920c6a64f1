// ss_walks.hpp -- the neighbour walks shared by the propagation kernels (ss_propagate.hip, ss_first_hop.hip): table-reading walks (one 16-byte chunk per lane) and the table-free first-hop walk.
#pragma once
#include "ss_common.hpp"

// Measurement build only (SS_EXTRA_FLAGS=-DSS_FUSED_ABLATE=<mask>, tools/ablate_fused.sh): parts of the fused stage's wavefront are left out so
// that what each costs can be read off the launch's duration -- the results of such a build are WRONG by construction.
//   1: no MinHash rows at all   2: no HLL side (post / fold / finish)   4: no exact evaluation of the winners (phase 2 + ambiguity)
//   8: no two-phase walk (the update loop)   16: neighbour ids are not hashed   32: no winner fetch (the two ds_bpermute per permutation and segment)
#ifndef SS_FUSED_ABLATE
#define SS_FUSED_ABLATE 0
#endif

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

// The generic walks below take the neighbours t = first, first + stride, ... < total four at a time: four id loads, then four
// row loads, then the four folds.  Two rules keep the loads of a batch in flight together (both learnt from the ISA): an id is
// loaded UNCONDITIONALLY (lanes past the row's end read a word that always exists and ignore it) -- a load inside `if (t < deg)`
// is awaited with vmcnt(0) at the end of its branch -- and nothing touches a loaded id (not even the widening to int64) before
// every load of the batch is issued.  The plain `#pragma unroll 4` loop they replace cost two dependent round trips per neighbour.
constexpr int kWalkBatch = 4;

// min over the neighbours t = first, first + stride, ... < total of MinHash chunk c (16 bytes per lane)
__device__ __forceinline__ u32x4 minhash_walk(const uint32_t *__restrict__ mh_in, const int32_t *__restrict__ nb, int deg, int total,
                                              int64_t self_row, int first, int stride, int P, int c)
{
    u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const int32_t *always_valid = reinterpret_cast<const int32_t *>(mh_in);
    for (int t = first; t < total; t += kWalkBatch * stride) {
        int id[kWalkBatch];
        u32x4 x[kWalkBatch];
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) id[k] = *(t + k * stride < deg ? nb + t + k * stride : always_valid);
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) {
            const int tk = t + k * stride;
            const int64_t j = tk < deg ? (int64_t)id[k] : self_row;
            x[k] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (tk < total) x[k] = *reinterpret_cast<const u32x4 *>(mh_in + j * P + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) acc = min4(acc, x[k]);
    }
    return acc;
}

// P = 128 (two 32-lane subgroups, subgroup sg takes the neighbours t = sg, sg + 2, ...): the first <= 64 neighbour ids of
// the row arrive with ONE coalesced load (lane l holds id l) and are handed out with v_readlane -- scalar, no memory
// pipeline -- instead of one broadcast load per neighbour and subgroup; the rest of a longer row is walked as before
__device__ __forceinline__ u32x4 minhash_walk128(const uint32_t *__restrict__ mh_in, const int32_t *__restrict__ nb, int deg, int total,
                                                 int64_t self_row, int lane)
{
    const int sg = lane >> 5, c = lane & 31;
    const int my_nb = lane < deg ? nb[lane] : 0;
    const int head = total < kWave ? total : kWave;
    u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // one pair of neighbours per iteration, NOT batched: batches of 2 / 4 pairs (masked or clamped) measured 194-198 us
    // against 187 us for this loop -- eight waves per SIMD already cover the latency, extra instructions only cost issue slots.
    // Round 3, all three shapes (SS_MH_BATCH = 1 / 2 / 3 / 4 pairs per iteration): collab size 191.4 / 193.1 / 196.2 / 199.4 us, ppa size
    // 2 936 / 2 977 / 2 972 / 2 975 us, citation2 size (1.5 GB table, HBM-resident) 5 661 / 5 566-5 591 / 5 572-5 604 / 5 581-5 593 us:
    // -1.5 % where the table lives in HBM, +1 % where it does not.  Not worth a second instantiation.
#ifndef SS_MH_BATCH
#define SS_MH_BATCH 1
#endif
#if SS_MH_BATCH == 1
    for (int t0 = 0; t0 < head; t0 += 2) {  // wave-uniform trip count
        const int s0 = __builtin_amdgcn_readlane(my_nb, t0), s1 = __builtin_amdgcn_readlane(my_nb, (t0 + 1) & (kWave - 1));
        const int t = t0 + sg;
        const int64_t j = t < deg ? (int64_t)(sg ? s1 : s0) : self_row;
        if (t < head) acc = min4(acc, *reinterpret_cast<const u32x4 *>(mh_in + j * 128 + 4 * c));
    }
#else
    for (int t0 = 0; t0 < head; t0 += 2 * SS_MH_BATCH) {  // wave-uniform trip count
        u32x4 x[SS_MH_BATCH];
#pragma unroll
        for (int k = 0; k < SS_MH_BATCH; ++k) {
            const int s0 = __builtin_amdgcn_readlane(my_nb, (t0 + 2 * k) & (kWave - 1));
            const int s1 = __builtin_amdgcn_readlane(my_nb, (t0 + 2 * k + 1) & (kWave - 1));
            const int t = t0 + 2 * k + sg;
            const int64_t j = t < deg ? (int64_t)(sg ? s1 : s0) : self_row;
            x[k] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (t < head) x[k] = *reinterpret_cast<const u32x4 *>(mh_in + j * 128 + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < SS_MH_BATCH; ++k) acc = min4(acc, x[k]);
    }
#endif
    if (total > kWave) acc = min4(acc, minhash_walk(mh_in, nb, deg, total, self_row, kWave + sg, 2, 128, c));
    return acc;
}

// hub / mega rows (propagate_hub_kernel): ONE wavefront folds a CONTIGUOUS chunk of <= 64 neighbours, nb[0 .. total) with
// nb[t >= deg] = the implicit self loop.  Ids arrive with one coalesced load and are handed out with v_readlane, the two
// 32-lane halves take alternating neighbours, four row loads per lane are requested before the first is folded (the generic
// batched walk pays a second, dependent, round trip per batch for its broadcast id loads)
__device__ __forceinline__ u32x4 minhash_chunk64(const uint32_t *__restrict__ mh_in, const int32_t *__restrict__ nb, int deg, int total,
                                                 int64_t self_row, int lane)
{
    const int sg = lane >> 5, c = lane & 31;
    const int my_nb = lane < deg ? nb[lane] : 0;
    u32x4 acc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int t0 = 0; t0 < total; t0 += 8) {  // wave-uniform trip count
        u32x4 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s0 = __builtin_amdgcn_readlane(my_nb, (t0 + 2 * k) & (kWave - 1));
            const int s1 = __builtin_amdgcn_readlane(my_nb, (t0 + 2 * k + 1) & (kWave - 1));
            const int t = t0 + 2 * k + sg;
            const int64_t j = t < deg ? (int64_t)(sg ? s1 : s0) : self_row;
            x[k] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (t < total) x[k] = *reinterpret_cast<const u32x4 *>(mh_in + j * 128 + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = min4(acc, x[k]);
    }
    return acc;
}

// byte-wise max over the same neighbour walk of HLL chunk c; even / odd bytes accumulated as packed u16
__device__ __forceinline__ u32x4 hll_walk(const uint8_t *__restrict__ hll_in, const int32_t *__restrict__ nb, int deg, int total,
                                          int64_t self_row, int first, int stride, int M, int c)
{
    u32x4 ae = {0u, 0u, 0u, 0u}, ao = {0u, 0u, 0u, 0u};
    const int32_t *always_valid = reinterpret_cast<const int32_t *>(hll_in);
    for (int t = first; t < total; t += kWalkBatch * stride) {
        int id[kWalkBatch];
        u32x4 x[kWalkBatch];
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) id[k] = *(t + k * stride < deg ? nb + t + k * stride : always_valid);
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) {
            const int tk = t + k * stride;
            const int64_t j = tk < deg ? (int64_t)id[k] : self_row;
            x[k] = u32x4{0u, 0u, 0u, 0u};
            if (tk < total) x[k] = *reinterpret_cast<const u32x4 *>(hll_in + j * M + 16 * c);
        }
#pragma unroll
        for (int k = 0; k < kWalkBatch; ++k) hll_acc(ae, ao, x[k]);
    }
    return hll_acc_result(ae, ao);
}

// the first <= 16 neighbours of a row walked by ONE 16-lane DPP row: lane c fetches neighbour id c (one coalesced 64-byte
// load per row instead of one broadcast load per neighbour), the ids reach the other lanes through DPP row_newbcast, and
// four row chunks are requested before the first is consumed
template <int T0>
__device__ __forceinline__ void hll_visit16(const uint8_t *__restrict__ hll_in, int my_nb, int deg, int count, int64_t self_row, int c,
                                            u32x4 &ae, u32x4 &ao)
{
    if constexpr (T0 < 16) {
        if (__any(T0 < count)) {  // wave-uniform: skip batches no lane group needs
            u32x4 x[4];
            const int nbt[4] = {__builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T0, 0xF, 0xF, false),  // row_newbcast:T0 ..
                                __builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T0 + 1, 0xF, 0xF, false),
                                __builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T0 + 2, 0xF, 0xF, false),
                                __builtin_amdgcn_update_dpp(0, my_nb, 0x150 + T0 + 3, 0xF, 0xF, false)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t j = T0 + k < deg ? (int64_t)nbt[k] : self_row;
                x[k] = u32x4{0u, 0u, 0u, 0u};
                if (T0 + k < count) x[k] = *reinterpret_cast<const u32x4 *>(hll_in + j * 256 + 16 * c);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hll_acc(ae, ao, x[k]);
            }
            hll_visit16<T0 + 4>(hll_in, my_nb, deg, count, self_row, c, ae, ao);
        }
    }
}

__device__ __forceinline__ u32x4 hll_walk_first16(const uint8_t *__restrict__ hll_in, const int32_t *__restrict__ nb, int deg, int total,
                                                  int64_t self_row, int c)
{
    const int count = total < 16 ? total : 16;
    const int my_nb = c < deg ? nb[c] : 0;
    u32x4 ae = {0u, 0u, 0u, 0u}, ao = {0u, 0u, 0u, 0u};
    hll_visit16<0>(hll_in, my_nb, deg, count, self_row, c, ae, ao);
    return hll_acc_result(ae, ao);
}

__device__ __forceinline__ uint32_t permuted_hash(uint64_t a, uint64_t b, uint64_t hv)
{
    return (uint32_t)mod_mersenne61(a * hv + b);
}

// processes neighbour batches base = first_batch*64, += batch_stride*64 of one row; acc / hll_row accumulate
template <int PPL, bool DO_MH, bool DO_HLL>
__device__ __forceinline__ void first_hop_walk(const int32_t *__restrict__ nb, int deg, int total, int64_t self_row, int first_batch,
                                               int batch_stride, int p, const uint64_t (&a)[PPL], const uint64_t (&b)[PPL],
                                               uint32_t (&acc)[PPL], uint32_t *hll_row, int lane)
{
    for (int base = first_batch * kWave; base < total; base += batch_stride * kWave) {
        const int t = base + lane;
        const int64_t nid = t < deg ? (int64_t)nb[t] : self_row;  // t == deg is the implicit self loop; t > deg unused
        const uint64_t hv = hash_u64((uint64_t)(nid + 1));
        const uint32_t hv_lo = (uint32_t)hv, hv_hi = (uint32_t)(hv >> 32);
        // HLL (hashing.py:126-137): every lane scatters ITS neighbour's single register into the LDS row
        if (DO_HLL && t < total) {
            const uint64_t bits = hv >> p;
            const int bl = bits ? 64 - __builtin_clzll(bits) : 0;
            atomicMax(&hll_row[hv_lo & 255u], (uint32_t)((64 - p) - bl + 1));
        }
        // MinHash: walk the batch; the neighbour's hash is wave-uniform, each lane evaluates its own permutations
        if (!DO_MH) continue;
        const int cnt = total - base < kWave ? total - base : kWave;
        int k = 0;
        // 4 neighbours per iteration: 4 * PPL independent multiply chains per lane (the integer multiplies have long
        // issue + latency; a single chain per wave leaves the VALU idle)
        for (; k + 3 < cnt; k += 4) {
            uint64_t hh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                hh[u] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv_hi, k + u) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k + u);
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const uint32_t v0 = permuted_hash(a[q], b[q], hh[0]);
                const uint32_t v1 = permuted_hash(a[q], b[q], hh[1]);
                const uint32_t v2 = permuted_hash(a[q], b[q], hh[2]);
                const uint32_t v3 = permuted_hash(a[q], b[q], hh[3]);
                const uint32_t m01 = v0 < v1 ? v0 : v1, m23 = v2 < v3 ? v2 : v3;
                const uint32_t v = m01 < m23 ? m01 : m23;
                acc[q] = v < acc[q] ? v : acc[q];
            }
        }
        for (; k + 1 < cnt; k += 2) {
            const uint64_t h0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv_hi, k) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k);
            const uint64_t h1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv_hi, k + 1) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k + 1);
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const uint32_t v0 = permuted_hash(a[q], b[q], h0);
                const uint32_t v1 = permuted_hash(a[q], b[q], h1);
                const uint32_t v = v0 < v1 ? v0 : v1;
                acc[q] = v < acc[q] ? v : acc[q];
            }
        }
        if (k < cnt) {
            const uint64_t h0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv_hi, k) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k);
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const uint32_t v = permuted_hash(a[q], b[q], h0);
                acc[q] = v < acc[q] ? v : acc[q];
            }
        }
    }
}

// one lane-quad of HLL registers (u32 each in LDS) -> packed bytes, stored + optional stats for the cardinality
__device__ __forceinline__ uint32_t pack_hll_quad(const uint32_t *row, int lane)
{
    const u32x4 r4 = *reinterpret_cast<const u32x4 *>(row + 4 * lane);
    return r4.x | (r4.y << 8) | (r4.z << 16) | (r4.w << 24);
}

// ---- two-phase MinHash first hop ----------------------------------------------------------------------------------------
// With x = (a*h + b) mod 2^64, the permuted hash is r = x mod (2^61 - 1) = (x & M) + (x >> 61) [- M], so unless the
// low word of x is within 8 of 2^32 its low 32 bits are simply   r32 = lo32(x) + (x >> 61),   0 <= x >> 61 <= 7.
// The first-hop kernel is bound by its VALU instruction count (measured: ~5 cycles per wave64 instruction whatever the
// opcode), and the exact hash costs 10 instructions per (neighbour, permutation).  Phase 1 therefore only finds, per
// permutation, WHICH neighbour attains the minimum, with 4 instructions per pair: x' = lo32(a_lo*h_lo + b_lo + 8)
// (v_mad_u64_u32; the +8 makes any low word that could wrap show up as x' < 8), key = x' with its low 6 bits replaced
// by the neighbour's slot in the 64-neighbour batch (v_bfi_b32), and the two smallest keys (v_med3_u32, v_min_u32).
// After each batch the hash of the slot holding the smallest key is fetched across lanes; phase 2 evaluates the
// exact hash once, for that neighbour.  The result is exact unless the two smallest keys fall into the same or
// adjacent 64-wide buckets (the order inside a bucket is by slot, not by value, and r32 may add up to 7) or a wrap is
// possible (bucket 0): probability ~4e-6 per (row, permutation); such rows are reported and redone by the exact walk.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// first_batch / batch_stride: the wavefront takes the 64-neighbour batches first_batch, first_batch + batch_stride, ... of the row
// (first_hop_hub_kernel: 16 wavefronts share a hub row or a slice of a mega row; each one's minimum is exact -- or flagged --
// on its own, the workgroup combines them with an exact min)
template <int PPL>
__device__ __forceinline__ bool first_hop_minhash_fast(const int32_t *__restrict__ nb, int deg, int total, int64_t self_row,
                                                       const uint64_t (&a)[PPL], const uint64_t (&b)[PPL], uint32_t (&acc)[PPL], int lane,
                                                       int first_batch = 0, int batch_stride = 1)
{
    uint32_t m1[PPL], m2[PPL], h1_lo[PPL], h1_hi[PPL], a_lo[PPL], b8[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        m1[q] = m2[q] = 0xFFFFFFFFu;
        h1_lo[q] = h1_hi[q] = 0u;
        a_lo[q] = (uint32_t)a[q];
        b8[q] = (uint32_t)b[q] + 8u;
    }
    bool seen_self = false;  // the row lists its own node explicitly (graphs that already carry self loops)
    // (the ids of the NEXT batch are requested before the current one is walked -- unconditionally, from a word that always exists
    // for the lanes past the row's end: a hub unit's wavefront walks 4 - 16 batches, each of which began with an exposed round trip)
    const int32_t *always_valid = nb;  // (deg >= 1 whenever a batch exists beyond the self loop; nb[0] is never out of bounds then)
    int t_first = first_batch * kWave + lane;
    int cur = deg > 0 ? *(t_first < deg ? nb + t_first : always_valid) : 0;
    for (int base = first_batch * kWave; base < total; base += batch_stride * kWave) {
        const int t = base + lane;
        const int tn = t + batch_stride * kWave;
        const int nxt = deg > 0 ? *(tn < deg ? nb + tn : always_valid) : 0;
        const int64_t nid = t < deg ? (int64_t)cur : self_row;
        cur = nxt;
        const uint64_t hv = hash_u64((uint64_t)(nid + 1));
        const uint32_t hv_lo = (uint32_t)hv, hv_hi = (uint32_t)(hv >> 32);
        int cnt = total - base < kWave ? total - base : kWave;
        // a duplicated neighbour makes the two smallest keys collide for EVERY permutation (exact, but through the slow
        // path); the common duplicate is an explicit self edge + the implicit self loop (slot `deg`, always the last):
        // drop the implicit one, it adds nothing to a min
        seen_self |= __any(t < deg && nid == self_row);
        if (seen_self && total > deg && base + cnt == total) --cnt;
        uint32_t before[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) before[q] = m1[q];
        auto update = [&](uint32_t h_lo, uint32_t slot) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const uint32_t x = a_lo[q] * h_lo + b8[q];
                // key = (x & ~63) | slot as ONE v_bfi_b32 with the inline constant 63 as the mask: left to itself the compiler
                // keeps ~63 in an SGPR, and a VOP3 instruction may read only one scalar register, so the wave-uniform
                // slot costs a second instruction (and + add) -- 11 instead of 9 VALU per neighbour in the unrolled walk
                uint32_t key;
                asm("v_bfi_b32 %0, 63, %1, %2" : "=v"(key) : "s"(slot), "v"(x));
                m2[q] = umed3(m1[q], m2[q], key);       // second smallest key so far
                m1[q] = key < m1[q] ? key : m1[q];
            }
        };
        int k = 0;
        for (; k + 3 < cnt; k += 4) {
            uint32_t hl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) hl[u] = (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k + u);
#pragma unroll
            for (int u = 0; u < 4; ++u) update(hl[u], (uint32_t)(k + u));
        }
        for (; k < cnt; ++k) update((uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k), (uint32_t)k);
        // the batch's hashes still sit one per lane: fetch the one whose slot now holds the smallest key
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int slot = (int)(m1[q] & 63u);
            const uint32_t cand_lo = (uint32_t)__shfl((int)hv_lo, slot), cand_hi = (uint32_t)__shfl((int)hv_hi, slot);
            const bool changed = m1[q] != before[q];
            h1_lo[q] = changed ? cand_lo : h1_lo[q];
            h1_hi[q] = changed ? cand_hi : h1_hi[q];
        }
    }
    bool ambiguous = false;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        acc[q] = permuted_hash(a[q], b[q], ((uint64_t)h1_hi[q] << 32) | h1_lo[q]);
        ambiguous |= (m1[q] < 64u) | ((m2[q] >> 6) - (m1[q] >> 6) <= 1u);
    }
    return ambiguous;
}

// ---- MinHash first hop of R consecutive rows by one wavefront (first_hop_rows_kernel, fused_hop_persistent_kernel) ------
// init() reads the bounds of all rows with one load (lane l holds rowptr[first + l] as an offset from the first row's start),
// the permutation parameters, and the first batch; load_batch() fetches 64 - R consecutive `col` entries of the chunk and
// hashes them, one per lane -- lanes 64 - R .. 63 always carry the hashes of the rows' OWN ids, so the implicit self loop of
// row r is slot 64 - R + r of whatever batch is current; row(r) walks row r over its slice of the batch(es) with the
// two-phase evaluation (first_hop_minhash_fast above), evaluates the winner exactly and stores the row.  Rows flagged
// ambiguous (and rows that list themselves: duplicates of the implicit self loop) are redone by the exact walk.
// MIR: rows also go to the peers' tables (peer-write build); a compile-time switch -- the store loop and the pointer it needs
// cost first_hop_rows_kernel 14 registers and fused_hop_persistent_kernel a wavefront per SIMD
template <int PPL, int R, bool MIR = false>
struct MinhashRows {
    static constexpr int P = PPL * kWave;
    static constexpr int kNb = kWave - R;  // col entries per batch
    int lane, rows, rel, c_n, base, p, hub_threshold;
    const Mirrors *mir;  // peers' tables (kernel-argument memory: the pointer stays valid for the kernel's lifetime)
    bool skip_hubs;
    int64_t i0, n_self, nid;
    const int32_t *nb;
    uint64_t a[PPL], b[PPL];
    uint32_t a_lo[PPL], b8[PPL], hv_lo, hv_hi;

    __device__ __forceinline__ void load_batch() { set_batch(batch_id()); }

    // once per wavefront: lane constants (permutation parameters, self-loop count, hub rule)
    __device__ __forceinline__ void setup(const GraphArgs &g, const uint64_t *__restrict__ pa, const uint64_t *__restrict__ pb, int p_, bool skip)
    {
        lane = threadIdx.x & (kWave - 1);
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            a[q] = pa[lane + kWave * q];
            b[q] = pb[lane + kWave * q];
            a_lo[q] = (uint32_t)a[q];
            b8[q] = (uint32_t)b[q] + 8u;
        }
        n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;
        p = p_;
        skip_hubs = skip;
        hub_threshold = g.hub_threshold;
        mir = MIR ? &g.mir : nullptr;
    }

    // rows [first_row, first_row + n_rows) become the current chunk; rp: lane l holds rowptr[first_row + l] (l <= n_rows)
    __device__ __forceinline__ void begin(const GraphArgs &g, int64_t first_row, int n_rows, int64_t rp)
    {
        i0 = first_row;
        rows = n_rows;
        const int64_t c_lo = ((int64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)rp >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)rp);
        rel = (int)(rp - c_lo);
        c_n = __builtin_amdgcn_readlane(rel, rows);  // col entries of the whole chunk
        nb = g.col + c_lo;
        base = 0;  // chunk-relative position of the current batch's first col entry
    }

    // the id lane `lane` contributes to the batch at `base`: a col entry, or (lanes kNb..) the id of one of the chunk's own rows
    __device__ __forceinline__ int64_t batch_id() const
    {
        const int t = base + lane;
        return lane >= kNb ? i0 + (lane - kNb) : (t < c_n ? (int64_t)nb[t] : 0);
    }
    __device__ __forceinline__ void set_batch(int64_t id)
    {
        nid = id;
        const uint64_t hv = (SS_FUSED_ABLATE & 16) ? (uint64_t)(nid + 1) : hash_u64((uint64_t)(nid + 1));
        hv_lo = (uint32_t)hv;
        hv_hi = (uint32_t)(hv >> 32);
    }

    // false: this wavefront owns no row of [g.row0, g.row1)
    __device__ __forceinline__ bool init(const GraphArgs &g, int64_t first_row, const uint64_t *__restrict__ pa,
                                         const uint64_t *__restrict__ pb, int p_, bool skip)
    {
        if (first_row >= g.row1) return false;
        const int n_rows = (int)(g.row1 - first_row < R ? g.row1 - first_row : R);
        const int l = threadIdx.x & (kWave - 1);
        const int64_t rp = l <= n_rows ? g.rowptr[first_row + l] : 0;
        setup(g, pa, pb, p_, skip);
        begin(g, first_row, n_rows, rp);
        load_batch();
        return true;
    }

    __device__ __forceinline__ void row(int r, uint32_t *__restrict__ mh_out)
    {
        const int64_t i = i0 + r;
        const int p0 = __builtin_amdgcn_readlane(rel, r), p1 = __builtin_amdgcn_readlane(rel, r + 1);
        const int deg = p1 - p0;
        if (skip_hubs && deg > hub_threshold) return;  // left to first_hop_hub_kernel
        const bool self = i < n_self;
        uint32_t acc[PPL];
        bool redo = false;
        if (deg + (self ? 1 : 0) == 0) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) acc[q] = 0u;  // no in-edge, no self loop: all-zero row (PyG default)
        } else {
            uint32_t m1[PPL], m2[PPL], h1_lo[PPL], h1_hi[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                m1[q] = m2[q] = 0xFFFFFFFFu;
                h1_lo[q] = h1_hi[q] = 0u;
            }
            auto update = [&](uint32_t h_lo, uint32_t slot) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    const uint32_t x = a_lo[q] * h_lo + b8[q];
                    uint32_t key;
                    asm("v_bfi_b32 %0, 63, %1, %2" : "=v"(key) : "s"(slot), "v"(x));  // (x & ~63) | slot, see first_hop_minhash_fast
                    m2[q] = umed3(m1[q], m2[q], key);
                    m1[q] = key < m1[q] ? key : m1[q];
                }
            };
            int pos = p0;
            bool seen_self = false;
            for (;;) {
                if (pos < p1 && pos >= base + kNb) {  // the row starts (or continues) beyond the current batch
                    base += (pos - base) / kNb * kNb;  // (a skipped hub row may lie in between: jump, do not step)
                    load_batch();
                }
                const int s_lo = pos - base;
                const int s_hi = p1 - base < kNb ? p1 - base : kNb;
                uint32_t before[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) before[q] = m1[q];
                // a row that lists itself would meet its implicit self loop as a duplicate (ambiguous for every permutation)
                seen_self |= __any(lane >= s_lo && lane < s_hi && nid == i);
                int k = s_lo;
                if constexpr (!(SS_FUSED_ABLATE & 8)) {
                for (; k + 3 < s_hi; k += 4) {
                    uint32_t hl[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) hl[u] = (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k + u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) update(hl[u], (uint32_t)(k + u));
                }
                for (; k < s_hi; ++k) update((uint32_t)__builtin_amdgcn_readlane((int)hv_lo, k), (uint32_t)k);
                } else if (s_hi > s_lo) update((uint32_t)__builtin_amdgcn_readlane((int)hv_lo, s_lo), (uint32_t)s_lo);
                pos = base + (s_hi > s_lo ? s_hi : s_lo);
                const bool last = pos >= p1;
                if (last && self && !seen_self) update((uint32_t)__builtin_amdgcn_readlane((int)hv_lo, kNb + r), (uint32_t)(kNb + r));
                // the batch's hashes still sit one per lane: fetch the one whose slot now holds the smallest key
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    const int slot = (int)(m1[q] & 63u);
                    const uint32_t cand_lo = (SS_FUSED_ABLATE & 32) ? hv_lo : (uint32_t)__shfl((int)hv_lo, slot);
                    const uint32_t cand_hi = (SS_FUSED_ABLATE & 32) ? hv_hi : (uint32_t)__shfl((int)hv_hi, slot);
                    const bool changed = m1[q] != before[q];
                    h1_lo[q] = changed ? cand_lo : h1_lo[q];
                    h1_hi[q] = changed ? cand_hi : h1_hi[q];
                }
                if (last) break;
            }
            bool ambiguous = false;
            if constexpr (SS_FUSED_ABLATE & 4) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) acc[q] = m1[q] ^ h1_lo[q] ^ h1_hi[q] ^ m2[q];
            } else
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                // x = a * h + b (mod 2^64); the permuted hash is x mod (2^61 - 1) = (x & M) + (x >> 61) [- M], whose low word
                // is lo32(x) + (x >> 61) unless the sum reaches M -- only possible when bits 32..60 of x are all ones: such
                // a row (2^-29 per evaluation) is flagged and redone exactly like the key collisions
                const uint64_t lo = (uint64_t)a_lo[q] * h1_lo[q] + b[q];
                const uint32_t hi = (uint32_t)(lo >> 32) + a_lo[q] * h1_hi[q] + (uint32_t)(a[q] >> 32) * h1_lo[q];
                acc[q] = (uint32_t)lo + (hi >> 29);
                ambiguous |= ((hi & 0x1FFFFFFFu) == 0x1FFFFFFFu) | (m1[q] < 64u) | ((m2[q] >> 6) - (m1[q] >> 6) <= 1u);
            }
            redo = __any(ambiguous);
        }
        if (redo) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) acc[q] = 0xFFFFFFFFu;
            first_hop_walk<PPL, true, false>(nb + p0, deg, deg + (self ? 1 : 0), i, 0, 1, p, a, b, acc, nullptr, lane);
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            mh_out[i * P + lane + kWave * q] = acc[q];
            if constexpr (MIR) mirror_mh1(*mir, i * P + lane + kWave * q, acc[q]);
        }
    }
};

// everything after the first `walked` neighbours of the four rows of a wavefront have been folded into `acc` (one 16-lane
// group per row, lane c = chunk c): the rest of rows up to kSolo neighbours by their own group, what is left of longer rows
// by all four groups together, then the row's statistics, its store and its cardinality.  `total` = 0 marks a group without
// a row to write (past the end, or a hub row left to the hub pass).
template <bool MIR = true>
__device__ __forceinline__ void hll_row16_finish(int64_t i, bool write, const int32_t *__restrict__ nb, int deg, int total, u32x4 acc,
                                                 int walked, const uint8_t *__restrict__ hll_in, uint8_t *__restrict__ hll_out,
                                                 float *__restrict__ cards_out, int64_t cards_stride, const EstimatorTables &est,
                                                 bool want_cards, int c /* lane & 15 */, const Mirrors &mir)
{
    constexpr int M = 256;
    // every lane group walks the first kSolo neighbours of its own row; what is left of longer rows is walked by the whole
    // wavefront, one row at a time (group g takes every 4th neighbour), so a wavefront lasts about sum(excess)/4 instead
    // of max(degree) iterations -- skewed graphs put rows of 10 and of 500 neighbours into the same wavefront
    constexpr int kSolo = 32;
    if (__any(total > walked)) acc = bytemax16(acc, hll_walk(hll_in, nb, deg, total < kSolo ? total : kSolo, i, walked, 1, M, c));
    const unsigned long long long_rows = __ballot(total > kSolo);
    if (long_rows) {
        const int grp = (threadIdx.x & (kWave - 1)) / kRow;
#pragma unroll
        for (int gg = 0; gg < kWave / kRow; ++gg) {
            if (!((long_rows >> (kRow * gg)) & 1ull)) continue;  // wave-uniform
            const uint64_t nb_bits = (uint64_t)(uintptr_t)nb;
            const int32_t *nb_g = (const int32_t *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(nb_bits >> 32), kRow * gg) << 32) |
                                                             (uint32_t)__builtin_amdgcn_readlane((int)nb_bits, kRow * gg));
            const int deg_g = __builtin_amdgcn_readlane(deg, kRow * gg), total_g = __builtin_amdgcn_readlane(total, kRow * gg);
            const int64_t i_g = ((int64_t)__builtin_amdgcn_readlane((int)((uint64_t)i >> 32), kRow * gg) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)i, kRow * gg);
            u32x4 part = hll_walk(hll_in, nb_g, deg_g, total_g, i_g, kSolo + grp, kWave / kRow, M, c);
            part = bytemax16(part, shfl_xor4(part, 16));
            part = bytemax16(part, shfl_xor4(part, 32));
            if (grp == gg) acc = bytemax16(acc, part);
        }
    }
    int nonzero = 0;
    float hsum = 0.0f;
    if (want_cards) {
        hll_dword_stats(acc.x, nonzero, hsum);
        hll_dword_stats(acc.y, nonzero, hsum);
        hll_dword_stats(acc.z, nonzero, hsum);
        hll_dword_stats(acc.w, nonzero, hsum);
        nonzero = row16_sum_i(nonzero);
        hsum = row16_sum_f(hsum);
    }
    if (write) {
        *reinterpret_cast<u32x4 *>(hll_out + i * M + 16 * c) = acc;
        if constexpr (MIR) mirror_hll16(mir, i * M + 16 * c, acc);
        if constexpr (MIR) {
            if (want_cards && c == 0) {
                const float card = hll_estimate(est, M - nonzero, hsum);
                cards_out[i * cards_stride] = card;
                mirror_card(mir, i * cards_stride, card);
            }
        } else {
            if (want_cards && c == 0) cards_out[i * cards_stride] = hll_estimate(est, M - nonzero, hsum);
        }
    }
}

// HLL table hop for FOUR destination rows per wavefront: one 16-lane DPP row per destination, lane c owns the 16-byte
// chunk c of the 256-byte HLL row.  Compared with one destination per wave (hll_walk + two cross-group shuffles +
// an epilogue that uses 16 of 64 lanes) this keeps 4x the loads in flight per wave and runs the cardinality
// epilogue for 4 rows at once.  `row` < 0 marks an inactive group.  M = 256 only.
__device__ __forceinline__ void hll_hop_row16(const GraphArgs &g, int64_t row, bool skip_hubs, const uint8_t *__restrict__ hll_in,
                                              uint8_t *__restrict__ hll_out, float *__restrict__ cards_out, int64_t cards_stride,
                                              const EstimatorTables &est, bool want_cards, int c /* lane & 15 */)
{
    const bool ok = row >= 0;
    const int64_t i = ok ? row : 0;
    const int64_t rb = g.rowptr[i];
    const int deg = (int)(g.rowptr[i + 1] - rb);
    const bool hub = skip_hubs && deg > g.hub_threshold;
    const int64_t n_self = g.n_self_dev ? *g.n_self_dev : g.n_self;
    const int total = (!ok || hub) ? 0 : deg + (i < n_self ? 1 : 0);
    const int32_t *nb = g.col + rb;
    const u32x4 acc = hll_walk_first16(hll_in, nb, deg, total, i, c);
    hll_row16_finish(i, ok && !hub, nb, deg, total, acc, 16, hll_in, hll_out, cards_out, cards_stride, est, want_cards, c, g.mir);
}

// launches defined in ss_first_hop.hip / ss_propagate.hip that ss_fused_hop_stage strings together (`lead`, `skip_hubs`: ss_hub.hpp)
int launch_first_hop_hub_only(const GraphArgs &g, const uint64_t *a, const uint64_t *b, int P, uint32_t *mh_out, int p, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream);
int launch_propagate_hub_only(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, const uint8_t *hll_in, uint8_t *hll_out,
                              float *cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream);
int launch_minhash_hop(const GraphArgs &g, const uint32_t *mh_in, uint32_t *mh_out, bool skip_hubs, int lead, const uint8_t *hub_hll_in,
                       uint8_t *hub_hll_out, float *hub_cards_out, int64_t cards_stride, const ss_hll_params &prm, hipStream_t stream);
int launch_hll_first_hop_rows(const GraphArgs &g, int p, uint8_t *hll_out, float *cards_out, int64_t cards_stride, const ss_hll_params &prm,
                              bool skip_hubs, int lead, const uint64_t *a, const uint64_t *b, uint32_t *hub_mh_out, int P, hipStream_t stream);

}  // namespace ss
