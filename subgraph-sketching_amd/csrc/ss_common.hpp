// ss_common.hpp -- shared device helpers for the gfx950 subgraph-sketching kernels.
// Wavefront = 64 lanes; a DPP "row" = 16 lanes, which is the unit one node pair / one HLL row
// segment is mapped to throughout (see DESIGN.md "lane mapping").
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "subgraph_sketch.h"
#include "subgraph_sketch_debug.h"

#define SS_LAUNCH_CHECK()                                        \
    do {                                                         \
        if (hipGetLastError() != hipSuccess) return SS_ERR_LAUNCH; \
    } while (0)

namespace ss {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kRow = 16;  // lanes per DPP row

// pandas.util.hash_array on int64 data == splitmix64 finaliser (reference hashing.py:121,128)
__device__ __forceinline__ uint64_t hash_u64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

// x mod (2^61 - 1) for any 64-bit x
__device__ __forceinline__ uint64_t mod_mersenne61(uint64_t x)
{
    const uint64_t M = (1ULL << 61) - 1;
    uint64_t r = (x & M) + (x >> 61);
    return r >= M ? r - M : r;
}

// ---- byte-parallel helpers on HLL registers (4 registers per dword) --------------------------------
// even/odd byte split: both halves are valid packed-u16 operands (odd bytes stay in the high byte of
// each u16 lane, so unsigned 16-bit max orders them like the bytes).
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
__device__ __forceinline__ uint32_t bytemax4(uint32_t a, uint32_t b)
{
    const uint32_t e = pk_max_u16(a & 0x00FF00FFu, b & 0x00FF00FFu);
    const uint32_t o = pk_max_u16(a & 0xFF00FF00u, b & 0xFF00FF00u);
    return e | o;
}
// running byte-wise max of many 16-byte chunks as packed u16: `ae` holds the even bytes (masked in), `ao` the odd ones -- for those
// the chunk goes in UNMASKED: a u16 max is decided by its high byte, so the high byte of every half of `ao` is the max of the odd
// bytes seen, whatever its low byte has become (12 instead of 16 instructions per chunk; hll_acc_result masks once at the end)
__device__ __forceinline__ void hll_acc(u32x4 &ae, u32x4 &ao, const u32x4 &x)
{
    ae.x = pk_max_u16(ae.x, x.x & 0x00FF00FFu); ao.x = pk_max_u16(ao.x, x.x);
    ae.y = pk_max_u16(ae.y, x.y & 0x00FF00FFu); ao.y = pk_max_u16(ao.y, x.y);
    ae.z = pk_max_u16(ae.z, x.z & 0x00FF00FFu); ao.z = pk_max_u16(ao.z, x.z);
    ae.w = pk_max_u16(ae.w, x.w & 0x00FF00FFu); ao.w = pk_max_u16(ao.w, x.w);
}
__device__ __forceinline__ u32x4 hll_acc_result(const u32x4 &ae, const u32x4 &ao)
{
    return u32x4{ae.x | (ao.x & 0xFF00FF00u), ae.y | (ao.y & 0xFF00FF00u), ae.z | (ao.z & 0xFF00FF00u), ae.w | (ao.w & 0xFF00FF00u)};
}
__device__ __forceinline__ u32x4 bytemax16(u32x4 a, u32x4 b)
{
    u32x4 r;
    r.x = bytemax4(a.x, b.x);
    r.y = bytemax4(a.y, b.y);
    r.z = bytemax4(a.z, b.z);
    r.w = bytemax4(a.w, b.w);
    return r;
}
__device__ __forceinline__ u32x4 min4(u32x4 a, u32x4 b)
{
    u32x4 r;
    r.x = a.x < b.x ? a.x : b.x;
    r.y = a.y < b.y ? a.y : b.y;
    r.z = a.z < b.z ? a.z : b.z;
    r.w = a.w < b.w ? a.w : b.w;
    return r;
}

// ---- HLL register statistics (zero count V and harmonic sum S = sum_j 2^-reg_j of hashing.py:221,228) ----------
// 2^-r in bf16 is the bit pattern 0x3F80 - (r << 7): the registers of a dword are turned into two packed-bf16
// words (even / odd bytes) with shift-and-subtract, and summed by v_dot2c_f32_bf16 against (1.0, 1.0) -- exact
// products, fp32 accumulation.  Full-rate VALU costs 4 cycles per wave64 instruction on this part, so the
// instruction count of this helper (11 per dword instead of ~22 with per-byte extraction) is what matters.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t regs_even_to_bf16(uint32_t x) { return 0x3F803F80u - ((x << 7) & 0x7F807F80u); }
__device__ __forceinline__ uint32_t regs_odd_to_bf16(uint32_t x) { return 0x3F803F80u - ((x >> 1) & 0x7F807F80u); }

__device__ __forceinline__ float dot2_ones(uint32_t packed_bf16, float acc)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed_bf16), __builtin_bit_cast(bf16x2, 0x3F803F80u), acc,
                                           false);
}

// bit 7 of every byte set iff the byte (a register value <= 127) is non-zero
__device__ __forceinline__ uint32_t nonzero_byte_flags(uint32_t x) { return (x + 0x7F7F7F7Fu) & 0x80808080u; }

// accumulates the number of NON-zero registers and the harmonic sum of the 4 registers of one dword
__device__ __forceinline__ void hll_dword_stats(uint32_t w, int &nonzero, float &sum)
{
    sum = dot2_ones(regs_even_to_bf16(w), sum);
    sum = dot2_ones(regs_odd_to_bf16(w), sum);
    nonzero += __builtin_popcount(nonzero_byte_flags(w));
}

// ---- DPP reductions inside a 16-lane row: every lane ends with the row total ----------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: i <-> 7-i inside each 8 lanes
constexpr int kDppMirror = 0x140;     // row_mirror: i <-> 15-i inside the row

__device__ __forceinline__ int row16_sum_i(int v)
{
    v += dpp_i<kDppXor1>(v);
    v += dpp_i<kDppXor2>(v);
    v += dpp_i<kDppHalfMirror>(v);
    v += dpp_i<kDppMirror>(v);
    return v;
}
__device__ __forceinline__ float row16_sum_f(float v)
{
    v += __int_as_float(dpp_i<kDppXor1>(__float_as_int(v)));
    v += __int_as_float(dpp_i<kDppXor2>(__float_as_int(v)));
    v += __int_as_float(dpp_i<kDppHalfMirror>(__float_as_int(v)));
    v += __int_as_float(dpp_i<kDppMirror>(__float_as_int(v)));
    return v;
}

// ---- HLL++ estimator (reference hashing.py:194-232) ------------------------------------------------
// Tables are staged by the caller: raw/bias (sorted ascending by raw) and lc (linear-counting values)
// may point to LDS or global memory.
struct EstimatorTables {
    const float *raw;
    const float *bias;
    const float *lc;
    int n_tbl;
    int lc_min_zeros;
    float alpha_mm;
    float five_m;
};

__device__ __forceinline__ float sqdist(float e, float r)
{
    const float d = e - r;
    return d * d;
}

// mean of the bias entries at the 6 nearest raw estimates (fp32 squared distance, ties -> lower index):
// hashing.py:197-204.  raw is sorted, so the answer is a window of 6 consecutive entries.
__device__ __forceinline__ float bias_of_6nn(const EstimatorTables &t, float e)
{
    int lo = 0, hi = t.n_tbl;  // lower_bound: first index with raw >= e
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t.raw[mid] < e) lo = mid + 1; else hi = mid;
    }
    int w = lo - 3;
    w = w < 0 ? 0 : w;
    w = w > t.n_tbl - 6 ? t.n_tbl - 6 : w;
    while (w + 6 < t.n_tbl && sqdist(e, t.raw[w + 6]) < sqdist(e, t.raw[w])) ++w;
    while (w > 0 && sqdist(e, t.raw[w - 1]) <= sqdist(e, t.raw[w + 5])) --w;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += t.bias[w + k];
    return s / 6.0f;
}

// zeros = number of zero registers, hsum = sum_j 2^-reg_j
__device__ __forceinline__ float hll_estimate(const EstimatorTables &t, int zeros, float hsum)
{
    if (zeros > 0 && zeros >= t.lc_min_zeros) return t.lc[zeros];  // linear counting (:221-226)
    float e = t.alpha_mm / hsum;                                    // raw estimate    (:228)
    if (e <= t.five_m) e -= bias_of_6nn(t, e);                      // bias correction (:206-210)
    return e;
}

// 5 * 2^p as a float assembled from integer (scalar) arithmetic: 1.25 * 2^(p + 2).  `5.0f * (float)(1 << p)` is a VALU multiply, and its
// wave-uniform result then occupies a VECTOR register for the whole kernel -- in kernels that sit at their register budget that
// one register was spilled to scratch (round 4)
__device__ __forceinline__ float five_times_two_to(int p) { return __uint_as_float(((uint32_t)(129 + p) << 23) | 0x200000u); }

// cooperative staging of the estimator tables into LDS (all threads of the block call this)
constexpr int kLcLdsMax = 1025;  // lc table staged in LDS when m + 1 <= kLcLdsMax (p <= 10)
struct EstimatorLds {
    float raw[SS_MAX_TABLE];
    float bias[SS_MAX_TABLE];
    float lc[kLcLdsMax];
};

__device__ __forceinline__ EstimatorTables stage_tables(EstimatorLds &lds, const ss_hll_params &prm)
{
    const int m1 = (1 << prm.p) + 1;
    for (int i = threadIdx.x; i < prm.n_tbl; i += blockDim.x) {
        lds.raw[i] = prm.raw_est[i];
        lds.bias[i] = prm.bias[i];
    }
    const bool lc_in_lds = m1 <= kLcLdsMax;
    if (lc_in_lds)
        for (int i = threadIdx.x; i < m1; i += blockDim.x) lds.lc[i] = prm.lc_table[i];
    __syncthreads();
    EstimatorTables t;
    t.raw = lds.raw;
    t.bias = lds.bias;
    t.lc = lc_in_lds ? lds.lc : prm.lc_table;
    t.n_tbl = prm.n_tbl;
    t.lc_min_zeros = prm.lc_min_zeros;
    t.alpha_mm = prm.alpha_mm;
    t.five_m = five_times_two_to(prm.p);
    return t;
}

// The linear-counting table alone in LDS; raw / bias stay in global memory.  For kernels whose rows are nearly all in the
// linear-counting range (a hop-1 row of fewer than 147 neighbours at p = 8): 1 KB of LDS per workgroup instead of 8.
template <int M1>
struct LcLds {
    float lc[M1];
};

template <int M1>
__device__ __forceinline__ EstimatorTables stage_lc_only(LcLds<M1> &lds, const ss_hll_params &prm)
{
    const int m1 = (1 << prm.p) + 1;
    const bool lc_in_lds = m1 <= M1;
    if (lc_in_lds)
        for (int i = threadIdx.x; i < m1; i += blockDim.x) lds.lc[i] = prm.lc_table[i];
    __syncthreads();
    EstimatorTables t;
    t.raw = prm.raw_est;
    t.bias = prm.bias;
    t.lc = lc_in_lds ? lds.lc : prm.lc_table;
    t.n_tbl = prm.n_tbl;
    t.lc_min_zeros = prm.lc_min_zeros;
    t.alpha_mm = prm.alpha_mm;
    t.five_m = five_times_two_to(prm.p);
    return t;
}

// A compact image for kernels that are tight on LDS: LC entries of the linear-counting table and TBL entries of raw / bias; a
// table that does not fit stays in global memory (datasketch ships 200 entries for p = 8; SS_MAX_TABLE = 512 is the ABI's bound).
template <int LC, int TBL>
struct CompactEstimatorLds {
    float raw[TBL];
    float bias[TBL];
    float lc[LC];
};

template <int LC, int TBL>
__device__ __forceinline__ EstimatorTables stage_tables_compact(CompactEstimatorLds<LC, TBL> &lds, const ss_hll_params &prm)
{
    const int m1 = (1 << prm.p) + 1;
    const bool tbl_in_lds = prm.n_tbl <= TBL, lc_in_lds = m1 <= LC;
    if (tbl_in_lds)
        for (int i = threadIdx.x; i < prm.n_tbl; i += blockDim.x) {
            lds.raw[i] = prm.raw_est[i];
            lds.bias[i] = prm.bias[i];
        }
    if (lc_in_lds)
        for (int i = threadIdx.x; i < m1; i += blockDim.x) lds.lc[i] = prm.lc_table[i];
    __syncthreads();
    EstimatorTables t;
    t.raw = tbl_in_lds ? lds.raw : prm.raw_est;
    t.bias = tbl_in_lds ? lds.bias : prm.bias;
    t.lc = lc_in_lds ? lds.lc : prm.lc_table;
    t.n_tbl = prm.n_tbl;
    t.lc_min_zeros = prm.lc_min_zeros;
    t.alpha_mm = prm.alpha_mm;
    t.five_m = five_times_two_to(prm.p);
    return t;
}

constexpr int kMegaSlot = SS_MEGA_SLOT_BYTES, kMegaHllOffset = 1024;

// Cross-workgroup hand-off of the mega-row partials WITHOUT cache-wide fences: the 8 XCD L2s are not coherent with each
// other, and a device-scope release / acquire fence (__threadfence) writes back and invalidates a whole L2 -- measured
// 1.6 ms for 135 slices, slower than not slicing at all.  The protocol instead is the write-through one of
// MI355X_MICROARCH.md "Inter-workgroup visibility" / cdna_hip_programming.md Guideline 16 (R1):
//   producer  slot words leave as agent-scope relaxed atomic stores (global_store ... sc1: written through the L2, access by
//             access) -> EVERY storing wave drains them (publish_drain: s_waitcnt vmcnt(0); gfx9 counts stores in vmcnt, and
//             a store is only counted down once the memory side has acknowledged it) -> __syncthreads() -> one lane takes
//             the row's ticket with an agent-scope atomic add.
//   consumer  the workgroup that draws the last ticket waits for the atomic's return value (a data dependency), passes a
//             barrier, and reads every slot with agent-scope relaxed atomic loads (global_load ... sc1: L1 bypassed).
// A workgroup-scope fence in the producer is NOT enough (round-1 bug): on gfx950 it emits no s_waitcnt, so the ticket could
// reach the memory side before the slot stores and the last finisher combine a stale slot.  The ISA of both hub kernels
// (profiles/round2_handoff_isa.txt) shows `s_waitcnt vmcnt(0)` between the sc1 stores and the ticket's global_atomic_add.
__device__ __forceinline__ void coherent_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t coherent_load(const uint32_t *p)
{
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store4(uint8_t *p, u32x4 v)
{
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
    coherent_store(q, v.x); coherent_store(q + 1, v.y); coherent_store(q + 2, v.z); coherent_store(q + 3, v.w);
}
__device__ __forceinline__ u32x4 coherent_load4(const uint8_t *p)
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    return u32x4{coherent_load(q), coherent_load(q + 1), coherent_load(q + 2), coherent_load(q + 3)};
}
// every vector-memory operation this wave has issued (the sc1 slot stores in particular) has been acknowledged by the
// memory side.  Inline asm, placed AFTER the stores: the compiler can neither drop nor move it (a builtin wait can be
// elided when its scoreboard believes nothing is outstanding -- Guideline 16 pitfall 12).
__device__ __forceinline__ void publish_drain() { asm volatile("s_waitcnt vmcnt(0) ; SS_HANDOFF drain before ticket" ::: "memory"); }
// the ticket of a mega row: agent-scope RMW executed at the coherence point; returns the number of slices that arrived before
__device__ __forceinline__ int take_ticket(int32_t *counter) { return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void reset_ticket(int32_t *counter) { __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// tables of the other ranks that receive every finished row as well (ss_csr_graph.mirror_*)
struct Mirrors {
    int n;
    uint32_t *mh[SS_MAX_MIRRORS];
    uint8_t *hll[SS_MAX_MIRRORS];
    float *cards[SS_MAX_MIRRORS];
};

struct GraphArgs {  // device-side view of ss_csr_graph
    const int64_t *rowptr;
    const int32_t *col;
    int64_t N;
    int64_t n_self;
    const int64_t *n_self_dev;
    int hub_threshold;
    const int32_t *hub_rows;
    const int32_t *hub_count;
    int32_t *mega_rows;        // int4 {row, first_slice, n_slices, done} per mega row (done: ticket counter, self-resetting)
    const int32_t *mega_count;  // {mega rows, slices}
    uint8_t *mega_scratch;      // kMegaSlot bytes per slice: partial MinHash row at 0, partial HLL row at kMegaHllOffset
    int64_t row0, row1;  // destination rows [row0, row1) are computed by this launch
    // ss_minhash_hop_rows: the launch computes the rows listed here instead (n_list entries, torch-style negative ids allowed,
    // out-of-range ids ignored, duplicates harmless); nullptr for every other launch
    const int64_t *row_list = nullptr;
    int64_t n_list = 0;
    Mirrors mir = {};
    int32_t *hub_report = nullptr;  // ss_csr_graph.hub_report and the two device counters it is formed from
    const int32_t *report_hub_count = nullptr, *report_mega_count = nullptr;
    __host__ __device__ int64_t rows() const { return row1 - row0; }
    __device__ bool owns(int64_t i) const { return i >= row0 && i < row1; }
};

inline GraphArgs to_args(const ss_csr_graph &g)
{
    const bool all = g.row_end == 0 && g.row_begin == 0;
    const bool mega = g.mega_rows && g.mega_count && g.mega_scratch;
    GraphArgs a{g.rowptr, g.col, g.num_nodes, g.n_self_loops, g.n_self_loops_dev, g.hub_threshold, g.hub_rows, g.hub_count,
                mega ? const_cast<int32_t *>(g.mega_rows) : nullptr, mega ? g.mega_count : nullptr,
                mega ? static_cast<uint8_t *>(g.mega_scratch) : nullptr, all ? 0 : g.row_begin, all ? g.num_nodes : g.row_end};
    a.hub_report = g.report_hub_count ? g.hub_report : nullptr;
    a.report_hub_count = g.report_hub_count;
    a.report_mega_count = g.report_mega_count;
    a.mir.n = g.n_mirrors > 0 && g.n_mirrors <= SS_MAX_MIRRORS ? g.n_mirrors : 0;
    for (int m = 0; m < a.mir.n; ++m) {
        a.mir.mh[m] = g.mirror_mh[m];
        a.mir.hll[m] = g.mirror_hll[m];
        a.mir.cards[m] = g.mirror_cards[m];
    }
    return a;
}

// ss_csr_graph.hub_report: one thread of a first-hop launch, at its very start (the launch then runs for tens of microseconds:
// a store to pinned host memory issued here is long acknowledged when the kernel ends -- the same store by the LAST workgroup of
// the CSR build's finish launch added 10 us to that launch)
__device__ __forceinline__ void report_hub_rows(const GraphArgs &g)
{
    if (g.hub_report && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(g.hub_report, *g.report_hub_count + (g.report_mega_count ? g.report_mega_count[0] : 0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// a finished piece of a row also goes to the peers' tables (same element offset; a launch that does not produce a sketch never
// calls that sketch's helper, so its mirror entries may be null)
__device__ __forceinline__ void mirror_mh4(const Mirrors &mir, int64_t off, u32x4 v)
{
    for (int m = 0; m < mir.n; ++m) *reinterpret_cast<u32x4 *>(mir.mh[m] + off) = v;
}
__device__ __forceinline__ void mirror_mh1(const Mirrors &mir, int64_t off, uint32_t v)
{
    for (int m = 0; m < mir.n; ++m) mir.mh[m][off] = v;
}
__device__ __forceinline__ void mirror_hll16(const Mirrors &mir, int64_t off, u32x4 v)
{
    for (int m = 0; m < mir.n; ++m) *reinterpret_cast<u32x4 *>(mir.hll[m] + off) = v;
}
__device__ __forceinline__ void mirror_hll4(const Mirrors &mir, int64_t off, uint32_t v)
{
    for (int m = 0; m < mir.n; ++m) *reinterpret_cast<uint32_t *>(mir.hll[m] + off) = v;
}
__device__ __forceinline__ void mirror_card(const Mirrors &mir, int64_t off, float v)
{
    for (int m = 0; m < mir.n; ++m) mir.cards[m][off] = v;
}

// optional HIP-event bracket around a launch (ss_debug.hip; a no-op unless ss_profile_enable selected `tag`)
// `attached`: the span records nothing itself -- its two events are handed to ONE hipExtLaunchKernelGGL, which fills them from the
// dispatch's own completion signal (no marker packets in the queue: the two hipEventRecord of a recorded span cost a 0.43 ms step
// about 5 us, and they bracket the dispatch gap together with the kernel).  Used for the launches bench.py times inside its timed
// region (the MinHash table hop, the fused stage, the query); `launch()` below picks the right call.
struct ProfileSpan {
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream;
    int tag;
    bool attached;
    ProfileSpan(hipStream_t s, int tag, bool attached = false);
    ~ProfileSpan();
    template <typename... Args, typename F = void (*)(Args...)>
    void launch(F kernel, dim3 grid, dim3 block, Args... args)
    {
        if (attached && start) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, start, stop, 0, args...);
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
    }
    ProfileSpan(const ProfileSpan &) = delete;
    ProfileSpan &operator=(const ProfileSpan &) = delete;
};

inline bool row_range_ok(const ss_csr_graph &g)
{
    if (g.row_end == 0 && g.row_begin == 0) return true;
    return g.row_begin >= 0 && g.row_begin <= g.row_end && g.row_end <= g.num_nodes;
}

inline int check_params(const ss_hll_params *prm)
{
    if (!prm) return SS_ERR_INVALID_ARG;
    if (prm->p < 4 || prm->p > 16) return SS_ERR_UNSUPPORTED;
    if (prm->n_tbl < 6 || prm->n_tbl > SS_MAX_TABLE) return SS_ERR_INVALID_ARG;
    if (!prm->raw_est || !prm->bias || !prm->lc_table) return SS_ERR_INVALID_ARG;
    return SS_OK;
}

}  // namespace ss
