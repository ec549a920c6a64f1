// ss_csr.hip -- CSR-by-destination construction on the device.
//
// The reference materialises one message x[src] per edge and scatter-maxes it (PyG propagate,
// hashing.py:34,44).  The MI355X engine instead pulls: edges are grouped by destination once and every
// hop streams whole neighbour rows.  The order of sources inside a row is unspecified (min / max do not
// care), which lets the build be an MSD radix sort with
//   * NO per-edge global atomics -- device-scope atomics from 8 non-coherent XCD L2s are served at the memory
//     side (first version: 0.23 ms for 2.4 M edges),
//   * NO scattered small stores -- an 8-byte store per edge into hundreds of open segments reaches the memory side
//     as 3-5x its useful bytes (second version, PMC: 1.09 GB written for 0.34 GB on a ppa-sized graph): every
//     step sorts a 4096-edge tile by key in LDS first and writes it back as ONE contiguous tile, and
//   * NO counting pass over the edges and NO scan over counters (third version: two counting kernels that re-read the
//     edges and three scans per partition pass, 3.6x the algorithmic traffic at ogbl-ppa / -citation2 size): a sorted tile
//     needs no global offsets -- it is located through its exclusive key offsets off[key][tile] ("run descriptors").
//
// A "fine bucket" is 2^node_shift consecutive destination nodes (<= 1024, fewer for dense graphs so that a bucket's
// edges fit the LDS image of the last step), reached by 1 - 3 tile-sort LEVELS of at most 256 keys each:
//   level 0   tile_sort_kernel     edges (16 B) -> tiles of (src, dst) int2 keyed by the top bits of dst, off0
//   level l   regroup_sort_kernel  the input of group G = (parent group g, key k) is the VIRTUAL concatenation of the runs
//                                  (tile, k) over the tiles of g, read through the descriptors and cut into 4096-edge chunks:
//                                  chunk c of G becomes tile (G, c) of this level, keyed by the next bits of dst
//   finish    finish_runs_kernel   one workgroup per fine bucket: its runs gathered from the tiles of its parent group, LDS
//                                  histogram over its nodes, LDS scan -> rowptr, sources placed into an LDS image of the
//                                  bucket's col segment and streamed out.  The start of the bucket in col is
//                                  base[g] + sum over those tiles of off[k][tile]: no scan kernel, no global counters.
//                                  Buckets above the image (node ids correlated with degree) are split over several workgroups.
// The records of the LAST level are packed, src | (dst & (2^node_shift - 1)) << src_bits (4 bytes), when the ids fit.
// What is needed between two levels is small: per (g, k) a prefix of the run lengths over the tiles and the first tile of
// every chunk (level_scan_kernel), the tile ranges of the groups (level_tiles_kernel) and a header per tile
// (level_fill_kernel) -- descriptor-sized arrays.  One level up to ogbl-collab size (2 launches), two up to 65 536 fine
// buckets (ogbl-ppa, ogbl-citation2), three beyond (N > 67 M nodes, or a few million nodes of a very dense graph).
// Traffic per edge at two levels: 16 + 8 (level 0) + 8 + 4 (level 1) + 4 + 4 (finish) = 44 B against 20 B algorithmic.
// The first level also validates ids and reduces max(id)+1 (the self-loop count of add_self_loops, hashing.py:148);
// the finish step lists hub rows.  Nothing synchronises with the host.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "ss_common.hpp"

namespace ss {

constexpr int kTile = 4096;            // edges sorted in LDS at a time by a tile-sort level
constexpr int kMaxKeys = 256;          // fan-out of a tile-sort level
constexpr int kMaxLevels = 3;
constexpr int kSortThreads = 512;      // level 0: 8 edges per thread (256 threads x 16 edges: 19.3 us on the collab-like graph, 512 x 8: 16.1 us)
constexpr int kRegroupThreads = 512;   // levels >= 1
constexpr int kRegroupRuns = 512;      // run descriptors a regroup workgroup holds at a time (more: further rounds)
constexpr int kRunThreads = 512;       // finish: two workgroups per CU at 128 VGPRs, positions in steps of 512 (less padding than 1024; round 6: 1 024 threads
                                       // for one-level plans -- one workgroup per CU there -- measured 38 -> 44 us at ogbl-collab size, 72 -> 81 at rank^-0.9: not shipped)
constexpr int kDenseThreads = 1024;    // upper bound of the workgroup size of anything that walks shares (LDS arrays are sized for it)
constexpr int kFinishCap = 16384;      // edges of a bucket's col image in LDS (64 KiB)
constexpr int kRunCap = 1024;          // run descriptors the finish step keeps in LDS (two workgroups per CU: 80 KB each)
// Dense fine buckets (node ids correlated with degree: power-law graphs put 5 - 40 % of the edges into the first 1024 nodes) are
// NOT finished by their one workgroup -- a single CU walking such a segment was the whole build on those graphs (532 us of a
// 1.23 ms step at rank^-0.9 endpoints).  The finish launch only registers every bucket that does not fit the image; two further
// steps split each over several workgroups by TILE ranges of about kDensePart edges (a run is at most a tile long, so tile
// granularity balances the shares): dense_count (LDS histogram of a share, ONE global atomic per touched node and share: its
// return value is the share's offset inside the node's row) and dense_place (row starts from the summed counters, sources stored
// at start + share offset + LDS cursor).  (Walking a bucket of up to two images in node sub-ranges, re-reading it for each, as
// round 3 did: 61 us instead of 23 for the finish launch of a collab-size graph with rank^-0.5 endpoints.)
constexpr int kDenseMin = kFinishCap;  // a fine bucket with more edges than this is split ...
constexpr int kDensePart = 8192;       // ... into shares of about this many edges
// dense_count words: [0] registered (dense) buckets, [1] shares, [2] the hint (registered buckets before it have no unclaimed share), [3] spent buckets,
// [kArriveBase + 16 k] (k < kArriveWords, one cache line each) fine buckets whose workgroup has decided -- 64 sharded words: one word
// takes ~90 atomics per microsecond
constexpr int kArriveBase = 16, kArriveWords = 64, kDenseSyncInts = kArriveBase + 16 * kArriveWords;
struct DenseSync {     // per registered bucket, zeroed by the tile sort of the build
    int32_t pub;       // != 0: descriptor, share bounds and zeroed counters are visible; the value is the bucket's number of shares
    int32_t next;      // count step: next share to claim (>= shares: none left)
    int32_t counted;   // shares whose count step is complete
    int32_t pnext;     // place step: next ticket for the shares whose counting workgroup moved on (orphans)
    int32_t orphans;   // number of those
    int32_t pad[3];
};

// average edges of a fine bucket the plans aim for: 3/4 of the image (a bucket above it goes to the dense steps, at some cost;
// half as many finish workgroups as at 1/2, each with the same fixed latencies).  SS_CSR_BUCKET_EDGES: tuning hook
inline int64_t bucket_edges_target()
{
    static const int64_t env = getenv("SS_CSR_BUCKET_EDGES") ? atoll(getenv("SS_CSR_BUCKET_EDGES")) : 0;
    return env > 0 ? env : kFinishCap * 3 / 4;
}

struct LevelPlan {
    int levels;
    int node_shift;                 // fine bucket = dst >> node_shift
    int shift[kMaxLevels];          // level l keys on dst >> shift[l]; shift[levels - 1] == node_shift
    int keys[kMaxLevels];           // fan-out (level 0: ceil(N / 2^shift[0]) <= 256; below: powers of two <= 256)
    int log2_keys[kMaxLevels];      // -1 at level 0
    int64_t groups[kMaxLevels + 1]; // groups[l]: groups at the INPUT of level l (groups[0] = 1); groups[levels]: fine buckets
    int64_t tmax[kMaxLevels];       // tile slots of level l (row stride of its descriptor arrays)
    bool packed;                    // records of the last level are 4 bytes
    int src_bits;                   // 32 - node_shift
    bool packed0;                   // two-level plans: the records of level 0 are 4 bytes too, src | (dst & (2^shift[0] - 1)) << src_bits0
    int src_bits0;                  // 32 - shift[0]
};

inline int ceil_log2_i64(int64_t x)
{
    int b = 0;
    while (((int64_t)1 << b) < x) ++b;
    return b;
}

// max_src: sources are < max_src (N for edge lists, B for link lists)
inline bool make_plan(int64_t N, int64_t E, int64_t max_src, LevelPlan &p)
{
    if (N < 0 || E < 0 || N >= ((int64_t)1 << 31)) return false;
    if (E >= ((int64_t)1 << 32) - 2 * kTile) return false;  // record indices are 32-bit
    const int64_t n = N > 0 ? N : 1;
    // fine bucket = 1024 nodes when that gives <= 256 buckets (one level); otherwise a bucket sized for about 3/4 kFinishCap
    // edges on average, between 64 and 1024 nodes
    int node_shift = 10;
    if (((n + 1023) >> 10) > kMaxKeys)
        while (node_shift > 6 && (E / n) * ((int64_t)1 << node_shift) > bucket_edges_target()) --node_shift;
    if (const char *forced = getenv("SS_CSR_NODE_SHIFT")) {  // test hook: reach the three-level plans with small graphs
        const int f = atoi(forced);
        if (f >= 4 && f <= 10) node_shift = f;
    }
    const int64_t fine = (n + ((int64_t)1 << node_shift) - 1) >> node_shift;
    const int tb = ceil_log2_i64(fine);
    if (tb > 24) return false;
    p.node_shift = node_shift;
    p.levels = fine <= kMaxKeys ? 1 : (tb <= 16 ? 2 : 3);
    int below[kMaxLevels] = {0, 0, 0};  // key bits of the levels under level 0: split evenly (run lengths 4096 / keys per level)
    if (p.levels == 2) below[1] = tb / 2;
    // two levels: inside a run of level 0 the top bits of dst ARE the run's key -- a record only needs src and the bits of dst below
    // that key.  If those fit 32 bits with level 0 taking a few more key bits than half (<= 256 keys still), the level-0 records are
    // written and read as 4 bytes instead of 8: 44 -> 36 bytes moved per edge (ogbl-ppa size: 20 + 12 bits)
    bool packed0 = false;
    if (p.levels == 2 && !getenv("SS_CSR_NO_PACK")) {
        const int need = ceil_log2_i64(max_src > 1 ? max_src : 2);
        for (int b1 = below[1]; b1 >= 1; --b1)  // fewer key bits at level 1 = more at level 0 = fewer low bits of dst in a record
            if (need + node_shift + b1 <= 32 && ((n + ((int64_t)1 << (node_shift + b1)) - 1) >> (node_shift + b1)) <= kMaxKeys) {
                below[1] = b1;
                packed0 = true;
                break;
            }
    }
    if (p.levels == 3) { below[2] = tb / 3; below[1] = (tb - below[2]) / 2; }
    int s = node_shift;
    for (int l = p.levels - 1; l >= 1; --l) {
        p.shift[l] = s;
        p.keys[l] = 1 << below[l];
        p.log2_keys[l] = below[l];
        s += below[l];
    }
    p.shift[0] = s;
    p.keys[0] = (int)((n + ((int64_t)1 << s) - 1) >> s);
    p.log2_keys[0] = -1;
    if (p.keys[0] > kMaxKeys) return false;
    const int64_t tiles0 = (E + kTile - 1) / kTile;
    p.groups[0] = 1;
    p.tmax[0] = tiles0 > 0 ? tiles0 : 1;
    for (int l = 1; l <= p.levels; ++l) {
        p.groups[l] = (n + ((int64_t)1 << p.shift[l - 1]) - 1) >> p.shift[l - 1];
        if (l < p.levels) p.tmax[l] = tiles0 + p.groups[l];
    }
    p.src_bits = 32 - node_shift;
    p.packed = max_src <= ((int64_t)1 << p.src_bits) && !getenv("SS_CSR_NO_PACK");  // (SS_CSR_NO_PACK: test hook, 8-byte records)
    p.src_bits0 = 32 - p.shift[0];
    p.packed0 = packed0 && p.packed && max_src <= ((int64_t)1 << p.src_bits0);
    return true;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int64_t max_dense_buckets(int64_t E) { return E / kDenseMin + 1; }
inline int64_t max_dense_shares(int64_t E) { return E / kDensePart + max_dense_buckets(E) + 1; }

// ---- block-wide exclusive scan of the 256 key counters inside a workgroup of THREADS >= 256 threads: thread k < 256 owns key k, the other
// wavefronts contribute zeros (tile sort / regroup run 512 threads per 4096-edge tile: 8 edges per thread)
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan_keys(uint32_t x /* 0 for threads >= 256 */, uint32_t *wave_tot /* LDS [THREADS / 64] */,
                                                              uint32_t *total)
{
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    uint32_t inc = x;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int w = 0; w < kMaxKeys / kWave; ++w) {
        if (w < wv) pre += wave_tot[w];
        tot += wave_tot[w];
    }
    if (total) *total = tot;
    __syncthreads();
    return pre + inc - x;
}

// first statement of every kernel of a build (workgroup-uniform: one word): *skip != 0 -> the outputs already hold this CSR
// (ss_csr_build_cached), every kernel exits
#define SS_CSR_SKIP(word)            \
    do {                             \
        if ((word) && *(word)) return; \
    } while (0)

#ifdef SS_CSR_TIMING  // measurement build only (SS_EXTRA_FLAGS=-DSS_CSR_TIMING): where a finish workgroup's time goes
__device__ unsigned long long csr_phase_ticks[16];
#define SS_TICK(i)                                                                                   \
    do {                                                                                             \
        if (threadIdx.x == 0) {                                                                      \
            const unsigned long long now_ = wall_clock64();                                          \
            atomicAdd(&csr_phase_ticks[i], now_ - tick_);                                            \
            tick_ = now_;                                                                            \
        }                                                                                            \
    } while (0)
#define SS_TICK_START() unsigned long long tick_ = wall_clock64()
// the finish launch's timeline: slot 8 = earliest start of a workgroup, slots 9.. = latest time any workgroup passed mark i
#define SS_MARK_START()                                                          \
    do {                                                                         \
        if (threadIdx.x == 0) atomicMin(&csr_phase_ticks[8], wall_clock64());    \
    } while (0)
#define SS_MARK(i)                                                               \
    do {                                                                         \
        if (threadIdx.x == 0) atomicMax(&csr_phase_ticks[i], wall_clock64());    \
    } while (0)
#else
#define SS_TICK(i)
#define SS_TICK_START()
#define SS_MARK_START()
#define SS_MARK(i)
#endif

// ---- level 0: every 4096-edge tile of the caller's list is sorted by key in LDS and written back as ONE contiguous tile (no global
// offsets are needed for that), together with the tile's exclusive key offsets off[key][tile] (key-major, so that a group's row is
// contiguous) and the tile's max(id) + 1.  A LINK list (ss_group_links_by_source: the pairs of a query grouped by their first node):
// src == nullptr, dst = links [B, 2] -- the key is the pair's first node, torch-style negative ids wrapped, ids out of range keyed
// to node 0 (the query kernel itself reports them and writes their NaN rows: nothing may be dropped here), the "source" is the
// pair's index.  The same with key_stride = 1 groups the entries of any id array (ss_csr_group_ids).
// PACKED (the only level of a one-level plan whose ids fit): records are src | (dst & (2^shift - 1)) << src_bits, 4 bytes
template <bool PACKED>
__global__ __launch_bounds__(kSortThreads) void tile_sort_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t E,
                                                                 int64_t N, int key_stride, int shift, int src_bits, int keys, int tiles,
                                                                 void *__restrict__ staged_,
                                                                 uint32_t *__restrict__ tile_off, unsigned long long *__restrict__ tile_max,
                                                                 int32_t *__restrict__ err, int32_t *__restrict__ hub_count,
                                                                 int32_t *__restrict__ mega_count, int32_t *__restrict__ dense_count,
                                                                 DenseSync *__restrict__ dense_sync, int dense_cap,
                                                                 const int32_t *__restrict__ skip, int32_t *__restrict__ bad_record)
{
    __shared__ int2 sorted[kTile];
    __shared__ uint32_t tile_hist[kMaxKeys], tile_offs[kMaxKeys], wave_tot[kSortThreads / kWave];
    __shared__ unsigned long long block_max;
    SS_CSR_SKIP(skip);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // outputs of the finish launch of this build are cleared here
        if (hub_count) *hub_count = 0;
        if (mega_count) mega_count[0] = mega_count[1] = 0;
        for (int i = 0; i < 8; ++i) dense_count[i] = 0;  // [0] registered buckets, [1] shares, [2] the hint
    }
    if (blockIdx.x == 0 && threadIdx.x < kArriveWords) dense_count[kArriveBase + 16 * threadIdx.x] = 0;
    // the words of every dense bucket the finish launch may register (one 32-byte entry per tile-sort workgroup: there are at
    // least as many tiles as entries, the loop is for the reader)
    for (int d = blockIdx.x; d < dense_cap; d += gridDim.x)
        if (threadIdx.x < sizeof(DenseSync) / 4) reinterpret_cast<int32_t *>(dense_sync + d)[threadIdx.x] = 0;
    if (threadIdx.x < kMaxKeys) tile_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) block_max = 0;
    __syncthreads();
    constexpr int PER = kTile / kSortThreads;  // 8 edges per thread
    const int64_t t0 = (int64_t)blockIdx.x * kTile;
    const int64_t hi = t0 + kTile < E ? t0 + kTile : E;
    int64_t my_max = -1;
    bool bad = false;
    int2 ed[PER];
    int key[PER];
    uint32_t rank[PER];
    // All loads of a thread are issued before any is used, none under a per-edge branch: with `if (e < hi) { load; use; }` per edge
    // the compiler put s_waitcnt vmcnt(0) behind every pair of loads -- eight load latencies in a row per tile.  Full tiles of
    // 16-byte aligned rows read two edges per lane and load (global_load_dwordx4).
    int64_t sv[PER], dv[PER];
    bool ok[PER];
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const bool vec = src && hi - t0 == kTile && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;  // (uniform)
    if (vec) {
        const i64x2 *src2 = reinterpret_cast<const i64x2 *>(src + t0), *dst2 = reinterpret_cast<const i64x2 *>(dst + t0);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) {
            const i64x2 s2 = src2[threadIdx.x + k * kSortThreads], d2 = dst2[threadIdx.x + k * kSortThreads];
            sv[2 * k] = s2.x; sv[2 * k + 1] = s2.y;
            dv[2 * k] = d2.x; dv[2 * k + 1] = d2.y;
            ok[2 * k] = ok[2 * k + 1] = true;
        }
    } else if (src) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t e = t0 + threadIdx.x + (int64_t)k * kSortThreads;
            ok[k] = e < hi;
            sv[k] = src[ok[k] ? e : t0];  // (t0 < E: a valid entry for the lanes past the end)
            dv[k] = dst[ok[k] ? e : t0];
        }
    } else {  // a link list: the key is the pair's first node, the "source" the pair's index (see above)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t e = t0 + threadIdx.x + (int64_t)k * kSortThreads;
            ok[k] = e < hi;
            sv[k] = ok[k] ? e : t0;
            dv[k] = dst[(int64_t)key_stride * sv[k]];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t u = dv[k] < 0 ? dv[k] + N : dv[k];
            if (ok[k] && (uint64_t)u >= (uint64_t)N) bad = true;  // (reported; the entry stays, keyed to node 0)
            dv[k] = (uint64_t)u < (uint64_t)N ? u : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        key[k] = -1;
        ed[k] = make_int2(0, 0);
        if (ok[k]) {
            int64_t s = sv[k];
            const int64_t d = dv[k];
            const int64_t mx = s > d ? s : d;
            my_max = mx > my_max ? mx : my_max;
            if ((uint64_t)d >= (uint64_t)N) { bad = true; continue; }    // out of range: dropped (and reported)
            if (src && (uint64_t)s >= (uint64_t)N) { bad = true; s = 0; }  // memory safe; the host raises in strict mode
            ed[k] = make_int2((int)s, (int)d);
            key[k] = (int)(d >> shift);
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) rank[k] = key[k] >= 0 ? atomicAdd(&tile_hist[key[k]], 1u) : 0u;
    __syncthreads();
    // exclusive scan of the <= 256 key counts: thread k < 256 owns key k, the other wavefronts contribute zeros
    uint32_t tile_n = 0;
    const uint32_t ex = block_exclusive_scan_keys<kSortThreads>(threadIdx.x < kMaxKeys ? tile_hist[threadIdx.x] : 0u, wave_tot, &tile_n);
    const int lane = threadIdx.x & (kWave - 1);
    if (threadIdx.x < kMaxKeys) tile_offs[threadIdx.x] = ex;
    __syncthreads();
    uint32_t *sorted32 = reinterpret_cast<uint32_t *>(sorted);
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (key[k] >= 0) {
            const uint32_t pos = tile_offs[key[k]] + rank[k];
            if (PACKED) sorted32[pos] = (uint32_t)ed[k].x | ((uint32_t)(ed[k].y & ((1 << shift) - 1)) << src_bits);
            else sorted[pos] = ed[k];
        }
    if ((int)threadIdx.x < keys) tile_off[(int64_t)threadIdx.x * tiles + blockIdx.x] = ex;
    if (threadIdx.x == 0) tile_off[(int64_t)keys * tiles + blockIdx.x] = tile_n;
    __syncthreads();
    if (PACKED) {
        uint32_t *staged = reinterpret_cast<uint32_t *>(staged_);
        for (uint32_t q = threadIdx.x; q < tile_n; q += kSortThreads) staged[t0 + q] = sorted32[q];
    } else {
        int2 *staged = reinterpret_cast<int2 *>(staged_);
        for (uint32_t q = threadIdx.x; q < tile_n; q += kSortThreads) staged[t0 + q] = sorted[q];
    }
    unsigned long long m = my_max < 0 ? 0ULL : (unsigned long long)my_max + 1ULL;
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if (lane == 0 && m) atomicMax(&block_max, m);
    if (bad && err) *err = 1;  // (SS_CSR_ERR_BOUNDS; a later wait_until that gives up adds bit 1 to it)
    if (bad && bad_record) *bad_record = 1;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[blockIdx.x] = block_max;
}

// ---- levels >= 1 -------------------------------------------------------------------------------------------------------------------
// what a regroup workgroup needs to know about its tile, in one 32-byte load (level_fill_kernel)
struct __attribute__((aligned(32))) TileHeader {
    uint32_t group;       // G
    uint32_t count;       // edges of the tile (4096 but for the last chunk of a group)
    uint32_t start;       // first record of the tile in the level's (compact) record array
    uint32_t position;    // 4096 * chunk: position of the tile's first edge in the concatenated runs of G
    uint32_t first_tile;  // parent tile whose run holds that position
    uint32_t tile_end;    // end of the parent group's tiles
    uint32_t pad[2];
};

// descriptor arrays of one level (device)
struct LevelArrays {
    uint32_t *off;                 // [keys + 1][tmax]
    uint32_t *prefix;              // [keys][tmax]     (levels that feed a regroup level)
    uint32_t *cfirst;              // [keys][tmax]     (same levels) first tile of every chunk of every child group
    uint32_t *tstart;              // [tmax]           (levels >= 1)
    uint32_t *tb;                  // [groups + 1]     (levels >= 1)
    TileHeader *header;            // [tmax]           (levels >= 1)
    uint32_t *gcount;              // [groups]         (levels >= 1)
    uint32_t *n_tiles;             // [1]              (levels >= 1)
    unsigned long long *base;      // [groups]         (levels >= 1)
};

// the tiles a regroup / finish workgroup gathers its runs from
struct ParentLevel {
    const uint32_t *off;                // [keys + 1][tmax] exclusive key offsets inside each tile (row `keys`: the tile's edge count)
    const uint32_t *tstart;             // [tmax] first record of each tile; nullptr: tile j starts at j * kTile (level 0)
    const uint32_t *tb;                 // [groups + 1] tiles of each group of this level; nullptr: one group, tiles [0, tiles0)
    const unsigned long long *base;     // [groups] first record (= first col slot) of each group; nullptr: 0
    int tmax, tiles0, keys, log2_keys;  // log2_keys < 0: level 0, the key IS the child group
};

struct ChildGroup {
    int g, k, t_lo, t_hi;
    unsigned long long base;
};

__device__ __forceinline__ ChildGroup child_group(const ParentLevel &p, int64_t G)
{
    ChildGroup c;
    c.g = p.log2_keys < 0 ? 0 : (int)(G >> p.log2_keys);
    c.k = p.log2_keys < 0 ? (int)G : (int)(G & (p.keys - 1));
    c.t_lo = p.tb ? (int)p.tb[c.g] : 0;
    c.t_hi = p.tb ? (int)p.tb[c.g + 1] : p.tiles0;
    c.base = p.base ? p.base[c.g] : 0ULL;
    return c;
}

// one workgroup per group G of the NEXT level: exclusive prefix of its run lengths over the parent's tiles (prefix[k][tile]), its
// edge count and where it starts (base of the parent group + sum over the tiles of off[k][tile]: off[k][tile] counts the edges of the tile with a smaller key)
__global__ __launch_bounds__(1024) void level_scan_kernel(ParentLevel par, uint32_t *__restrict__ prefix, uint32_t *__restrict__ cfirst,
                                                          uint32_t *__restrict__ gcount, unsigned long long *__restrict__ base_next,
                                                          const int32_t *__restrict__ skip)
{
    __shared__ uint32_t wave_tot[1024 / kWave];
    __shared__ unsigned long long red[1024 / kWave];
    SS_CSR_SKIP(skip);
    const ChildGroup c = child_group(par, blockIdx.x);
    const uint32_t *row0 = par.off + (int64_t)c.k * par.tmax, *row1 = row0 + par.tmax;
    uint32_t *P = prefix + (int64_t)c.k * par.tmax;
    // chunk q of this group (positions [4096 q, 4096 q + 4096) of its concatenated runs) begins inside exactly one non-empty run: its
    // tile goes to C[t_lo + q] (a group of T tiles has at most T chunks) -- the regroup workgroup of that chunk starts reading there
    uint32_t *C = cfirst + (int64_t)c.k * par.tmax + c.t_lo;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    uint32_t carry = 0;
    unsigned long long bsum = 0;
    for (int t0 = c.t_lo; t0 < c.t_hi; t0 += 4 * 1024) {  // four consecutive tiles per thread
        const int t = t0 + 4 * (int)threadIdx.x;
        uint32_t len[4], run = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            len[i] = 0;
            if (t + i < c.t_hi) {
                const uint32_t o0 = row0[t + i];
                len[i] = row1[t + i] - o0;
                bsum += o0;
            }
            run += len[i];
        }
        uint32_t inc = run;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t x = __shfl_up(inc, off);
            if (lane >= off) inc += x;
        }
        if (lane == kWave - 1) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 1024 / kWave; ++w) {
            if (w < wv) pre += wave_tot[w];
            tot += wave_tot[w];
        }
        uint32_t ex = carry + pre + inc - run;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t + i < c.t_hi) {
                P[t + i] = ex;
                const uint32_t q = (ex + kTile - 1) / kTile;
                if (len[i] && q * kTile < ex + len[i]) C[q] = (uint32_t)(t + i);
                ex += len[i];
            }
        carry += tot;
        __syncthreads();
    }
    for (int off = kWave / 2; off > 0; off >>= 1) bsum += __shfl_xor(bsum, off);
    if (lane == 0) red[wv] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 1024 / kWave; ++w) bsum += red[w];
        gcount[blockIdx.x] = carry;
        base_next[blockIdx.x] = c.base + bsum;
    }
}

// single workgroup: tile ranges of the groups, tb[G] = sum over the groups before G of ceil(count / 4096)
__global__ __launch_bounds__(1024) void level_tiles_kernel(const uint32_t *__restrict__ gcount, int64_t groups, uint32_t *__restrict__ tb,
                                                           uint32_t *__restrict__ n_tiles, const int32_t *__restrict__ skip)
{
    __shared__ uint32_t wave_tot[1024 / kWave];
    SS_CSR_SKIP(skip);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    uint32_t carry = 0;
    for (int64_t g0 = 0; g0 < groups; g0 += 1024) {
        const int64_t G = g0 + threadIdx.x;
        const uint32_t x = G < groups ? (uint32_t)(((unsigned long long)gcount[G] + kTile - 1) / kTile) : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 1024 / kWave; ++w) {
            if (w < wv) pre += wave_tot[w];
            tot += wave_tot[w];
        }
        if (G < groups) tb[G] = carry + pre + inc - x;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tb[groups] = carry;
        *n_tiles = carry;
    }
}

// one wavefront per group: the headers of its tiles
__global__ __launch_bounds__(kWave) void level_fill_kernel(ParentLevel par, const uint32_t *__restrict__ cfirst, const uint32_t *__restrict__ tb,
                                                          const uint32_t *__restrict__ gcount, const unsigned long long *__restrict__ base,
                                                          TileHeader *__restrict__ header, const int32_t *__restrict__ skip)
{
    SS_CSR_SKIP(skip);
    const uint32_t a = tb[blockIdx.x], b = tb[blockIdx.x + 1];
    if (a == b) return;
    const ChildGroup c = child_group(par, blockIdx.x);
    const uint32_t *C = cfirst + (int64_t)c.k * par.tmax + c.t_lo;
    const uint32_t n = gcount[blockIdx.x];
    const unsigned long long start = base[blockIdx.x];
    for (uint32_t q = threadIdx.x; q < b - a; q += kWave) {
        TileHeader h;
        h.group = blockIdx.x;
        h.position = q * (uint32_t)kTile;
        h.count = n - h.position < (uint32_t)kTile ? n - h.position : (uint32_t)kTile;
        h.start = (uint32_t)(start + h.position);
        h.first_tile = C[q];
        h.tile_end = (uint32_t)c.t_hi;
        h.pad[0] = h.pad[1] = 0;
        header[a + q] = h;
    }
}

struct LevelOut {
    void *staged;                       // int2 records, or packed uint32 records (PACKED)
    uint32_t *off;                      // [keys + 1][tmax]
    uint32_t *tstart;                   // [tmax]
    const TileHeader *header;           // [tmax]
    const uint32_t *n_tiles;
    int tmax, keys, shift;              // key = (dst >> shift) & (keys - 1)
    int src_bits, low_mask;             // PACKED: src | (dst & low_mask) << src_bits
};

// level >= 1: tile j = chunk c of group G; its <= 4096 edges are positions [4096 c, 4096 c + n) of the concatenated runs (tile, k)
// of the parent group's tiles.  Thread i takes positions i, i + 512, ...: a wavefront's 64 positions share a 64-aligned block, whose
// first run comes from a small table; the runs are <= kRegroupRuns descriptors in LDS (a chunk that spans more -- runs shorter
// than 8 edges on average -- takes further rounds).  Then exactly tile_sort_kernel: LDS counting sort by the level's key.
// PACKED_IN (the parent is level 0 of a two-level plan whose records fit): 4-byte input records src | (dst low bits) << in_src_bits; the high
// bits of dst are the group's key at level 0
template <bool PACKED, bool PACKED_IN = false>
__global__ __launch_bounds__(kRegroupThreads) __attribute__((amdgpu_waves_per_eu(8))) void regroup_sort_kernel(ParentLevel par, const int2 *__restrict__ staged_in,
                                                                       const uint32_t *__restrict__ prefix, LevelOut out,
                                                                       const int32_t *__restrict__ skip, int in_src_bits = 0, int in_shift = 0)
{
    __shared__ int2 sorted[kTile];
    __shared__ int32_t run_start[kRegroupRuns + 3];
    __shared__ uint32_t run_delta[kRegroupRuns];  // first record of the run minus its first position
    __shared__ uint16_t first_run[kTile / kWave];
    __shared__ uint32_t tile_hist[kMaxKeys], tile_offs[kMaxKeys], wave_tot[kRegroupThreads / kWave];
    SS_CSR_SKIP(skip);
    const uint32_t j = blockIdx.x;
    if (j >= *out.n_tiles) return;
    const TileHeader hd = out.header[j];
    const uint32_t G = hd.group, lo = hd.position;
    const int cnt = (int)hd.count;
    const int k = par.log2_keys < 0 ? (int)G : (int)(G & (par.keys - 1));
    const int t_hi = (int)hd.tile_end;
    const uint32_t *P = prefix + (int64_t)k * par.tmax, *row0 = par.off + (int64_t)k * par.tmax;
    if (threadIdx.x < kMaxKeys) tile_hist[threadIdx.x] = 0;
    constexpr int PER = kTile / kRegroupThreads;  // 8 positions per thread
    int2 ed[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) ed[k] = make_int2(0, 0);
    int t_base = (int)hd.first_tile, round_lo = 0;
    for (;;) {
        const int R = t_hi - t_base < kRegroupRuns ? t_hi - t_base : kRegroupRuns;
        if ((int)threadIdx.x < R) {
            const int t = t_base + threadIdx.x;
            const int32_t st = (int32_t)(P[t] - lo);  // (negative for the run the chunk begins inside of)
            run_start[threadIdx.x] = st;
            run_delta[threadIdx.x] = (par.tstart ? par.tstart[t] : (uint32_t)t * (uint32_t)kTile) + row0[t] - (uint32_t)st;
        }
        if (threadIdx.x < 3) run_start[R + threadIdx.x] = threadIdx.x == 0 && t_base + R < t_hi ? (int32_t)(P[t_base + R] - lo) : 0x7FFFFFFF;
        __syncthreads();
        const int round_hi = run_start[R] < cnt ? run_start[R] : cnt;  // positions [round_lo, round_hi) lie in these runs
        if (threadIdx.x < kTile / kWave) {  // first run of every 64-position block: the last run that begins at or before it
            int p0 = (int)threadIdx.x * kWave;
            p0 = p0 < round_lo ? round_lo : p0;
            int a = 0, b = R;
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (run_start[mid] <= p0) a = mid; else b = mid;
            }
            first_run[threadIdx.x] = (uint16_t)a;
        }
        __syncthreads();
        // lookups, then loads, then selects -- none under a per-position branch (the compiler serialises loads behind such
        // branches with s_waitcnt vmcnt(0): eight load latencies in a row); positions outside the round read a valid record
        uint32_t rec[PER];
        int run[PER];
        bool more = false;
#pragma unroll
        for (int k = 0; k < PER; ++k) {  // (three independent boundary reads per position, no loop: the chains interleave)
            int p = (int)threadIdx.x + k * kRegroupThreads;
            p = p < round_lo ? round_lo : (p >= round_hi ? round_hi - 1 : p);
            const int r0 = first_run[p / kWave];
            const int b1 = run_start[r0 + 1], b2 = run_start[r0 + 2], b3 = run_start[r0 + 3];
            more |= b3 <= p;
            run[k] = r0 + (b1 <= p ? 1 : 0) + (b2 <= p ? 1 : 0) + (b3 <= p ? 1 : 0);
            rec[k] = (uint32_t)p;
        }
        if (__builtin_expect(more, 0)) {  // runs shorter than ~20 edges
#pragma unroll
            for (int k = 0; k < PER; ++k)
                while (run_start[run[k] + 1] <= (int)rec[k]) ++run[k];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) rec[k] += run_delta[run[k]];
        int2 got[PER];
        if (PACKED_IN) {
            uint32_t got32[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) got32[k] = reinterpret_cast<const uint32_t *>(staged_in)[rec[k]];
#pragma unroll
            for (int k = 0; k < PER; ++k)
                got[k] = make_int2((int)(got32[k] & ((1u << in_src_bits) - 1u)), (int)(((uint32_t)G << in_shift) | (got32[k] >> in_src_bits)));
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) got[k] = staged_in[rec[k]];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = (int)threadIdx.x + k * kRegroupThreads;
            if (p >= round_lo && p < round_hi) ed[k] = got[k];
        }
        if (round_hi >= cnt) break;  // (workgroup-uniform)
        round_lo = round_hi;
        t_base += R;
        __syncthreads();
    }
    // ---- LDS counting sort by this level's key (as tile_sort_kernel) ----
    int key[PER];
    uint32_t rank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int p = (int)threadIdx.x + k * kRegroupThreads;
        key[k] = p < cnt ? (ed[k].y >> out.shift) & (out.keys - 1) : -1;
        rank[k] = key[k] >= 0 ? atomicAdd(&tile_hist[key[k]], 1u) : 0u;
    }
    __syncthreads();
    const uint32_t ex = block_exclusive_scan_keys<kRegroupThreads>(threadIdx.x < kMaxKeys ? tile_hist[threadIdx.x] : 0u, wave_tot, nullptr);
    if (threadIdx.x < kMaxKeys) tile_offs[threadIdx.x] = ex;
    __syncthreads();
    uint32_t *sorted32 = reinterpret_cast<uint32_t *>(sorted);
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (key[k] >= 0) {
            const uint32_t pos = tile_offs[key[k]] + rank[k];
            if (PACKED) sorted32[pos] = (uint32_t)ed[k].x | ((uint32_t)(ed[k].y & out.low_mask) << out.src_bits);
            else sorted[pos] = ed[k];
        }
    if ((int)threadIdx.x < out.keys) out.off[(int64_t)threadIdx.x * out.tmax + j] = ex;
    const uint32_t ts = hd.start;
    if (threadIdx.x == 0) {
        out.off[(int64_t)out.keys * out.tmax + j] = (uint32_t)cnt;
        out.tstart[j] = ts;
    }
    __syncthreads();
    if (PACKED) {
        uint32_t *dst32 = reinterpret_cast<uint32_t *>(out.staged) + ts;
        for (int q = threadIdx.x; q < cnt; q += kRegroupThreads) dst32[q] = sorted32[q];
    } else {
        int2 *dst2 = reinterpret_cast<int2 *>(out.staged) + ts;
        for (int q = threadIdx.x; q < cnt; q += kRegroupThreads) dst2[q] = sorted[q];
    }
}

// ---- finish: one workgroup per fine bucket --------------------------------------------------------------------------------
struct RowOutputs {
    int64_t *rowptr;
    int hub_threshold;
    int32_t *hub_rows, *hub_count, *mega_rows, *mega_count;
    const int32_t *skip;  // see SS_CSR_SKIP
    int32_t *err;         // the build's err_flag (nullable): bit 1 (SS_CSR_ERR_PROTOCOL) reports a wait that gave up, see wait_until
    int32_t *fault_word;  // process-wide count of such waits in pinned host memory (nullable; ss_csr_protocol_faults reads it without synchronising)
    unsigned long long wait_ticks;  // bound of wait_until
    uint32_t walk_max;    // buckets of up to this many edges are finished by their own workgroup (kDenseMin: only what fits the image; kWalkMax)
};

// exclusive scan of the per-node edge counts cnt[0..nb) of the bucket that starts at node0 -> excl[0..nb]; with `publish` the
// row starts (rowptr) and the hub / mega rows of the bucket are written too.  Called by all threads of the workgroup.
__device__ __forceinline__ void scan_bucket_nodes(const uint32_t *cnt, uint32_t *excl, uint32_t *wave_tot, int nb, int64_t node0, int64_t N,
                                                  unsigned long long seg_lo, uint32_t seg_n, bool publish, const RowOutputs &o)
{
    // two counters per thread at nb = 1024
    const int per = (nb + (int)blockDim.x - 1) / (int)blockDim.x;
    const int b0 = threadIdx.x * per;
    uint32_t run = 0;
    for (int k = 0; k < per; ++k) run += (b0 + k < nb) ? cnt[b0 + k] : 0u;
    uint32_t inc = run;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t x = __shfl_up(inc, off);
        if (lane >= off) inc += x;
    }
    if (lane == kWave - 1) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int w = 0; w < wv; ++w) pre += wave_tot[w];
    uint32_t ex = pre + inc - run;
    for (int k = 0; k < per; ++k) {
        if (b0 + k >= nb) break;
        const uint32_t c = cnt[b0 + k];
        excl[b0 + k] = ex;
        if (publish && node0 + b0 + k < N) {
            o.rowptr[node0 + b0 + k] = (int64_t)(seg_lo + ex);
            // (cross-workgroup appends: agent-scope atomics on the counters, plain stores into the claimed slots; nothing in
            // THIS launch reads the lists -- the propagation launches do, and a kernel boundary orders them behind these stores)
            if (o.hub_rows && c > (uint32_t)o.hub_threshold) {
                if (o.mega_rows && c > (uint32_t)SS_MEGA_SLICE) {  // walked slice by slice by all hub workgroups
                    const int slices = (int)((c + 1 + SS_MEGA_SLICE - 1) / SS_MEGA_SLICE);  // + 1: the implicit self loop
                    const int m = atomicAdd(&o.mega_count[0], 1);
                    const int first = atomicAdd(&o.mega_count[1], slices);
                    int4 *desc = reinterpret_cast<int4 *>(o.mega_rows + (int64_t)SS_MEGA_DESC_WORDS * m);
                    desc[0] = make_int4((int)(node0 + b0 + k), first, slices, 0);  // {row, first slice, slices, ticket (MinHash side)}
                    desc[1] = make_int4(0, 0, 0, 0);                               // {ticket (HLL side), -, -, -}
                } else {
                    o.hub_rows[atomicAdd(o.hub_count, 1)] = (int32_t)(node0 + b0 + k);
                }
            }
        }
        ex += c;
    }
    if (threadIdx.x == 0) excl[nb] = seg_n;
}

struct FinishLds {
    uint32_t cnt[1024], excl[1024 + 1];
    int32_t image[kFinishCap];  // LDS image of the bucket's col segment (before: the bucket's packed records, see RunEdges::replay)
    uint32_t wave_tot[kDenseThreads / kWave];
};

// ---- dense buckets inside the finish launch: who does what, and who may wait for whom ---------------------------------------------
// A bucket that does not fit the image is REGISTERED by its workgroup (descriptor, share bounds, zeroed counters), published with a
// release, and then worked off share by share by whoever has time: every workgroup of the launch that has finished its own bucket
// (or registered it) walks the published buckets and claims shares from their counters, and `helpers` extra workgroups at the end of
// the grid do nothing else.  Two steps per share: COUNT (LDS histogram of the share, one global atomic per touched node whose return
// value is the share's offset inside the node's row) and, once every share of the bucket is counted, PLACE.
// The rule that keeps this free of deadlocks whatever shares the device (eight processes on one GPU, a CU mask, a debugger): NOBODY
// EVER WAITS FOR A WORKGROUP THAT MAY NOT BE RUNNING.  The one wait -- `counted == shares` of a bucket before placing -- is entered
// only after the bucket's claim counter has run out, i.e. every uncounted share is in the hands of a workgroup that is executing its
// count step, and a count step waits for nothing.  A registered bucket nobody else has time for is worked off by its own workgroup;
// the dedicated helpers only poll for a bounded time and are never needed for completion.  (Rounds 3-4: the helpers spun until every
// bucket workgroup of the launch had arrived, with a trap after 2^28 spins -- workgroups that are dispatched per XCD, possibly behind
// other processes' waiting helpers; an intermittent failure of the eight-process test was never explained, VERDICT r4 #1.)
// Hand-offs follow cdna_hip_programming.md Guideline 16: producer stores -> barrier -> lane-0 agent release fence -> s_waitcnt
// vmcnt(0) -> relaxed agent atomic; consumer relaxed poll -> agent acquire fence -> barrier.
constexpr int kHintWord = 2;                                   // dense_count[2]: registered buckets before it have no unclaimed share
constexpr int kSpentWord = 3;                                  // dense_count[3]: registered buckets whose last share has been taken
constexpr unsigned long long kWaitTicks = 200000000ULL;        // wall_clock64 ticks (100 MHz): 2 s -- see wait_until
constexpr unsigned long long kHelperPatienceTicks = 100000ULL;  // 1 ms: a dedicated helper that finds nothing for that long leaves

// a wait that (by the rule above) only running workgroups can end; should that ever be wrong (a process descheduled for seconds:
// a debugger, heavy oversubscription) it gives up after ~2 s instead of hanging the device or trapping the context -- and the
// build then FAILS LOUDLY: the share is not placed, so the CSR is incomplete, and the give-up is reported three ways --
// bit 1 of the build's err_flag (SS_CSR_ERR_PROTOCOL: whoever reads the flag for out-of-range ids sees it), the process-wide
// count in pinned host memory behind ss_csr_protocol_faults() (read without synchronising: the host mirror raises from its next
// call / check_errors() even for builds that were given no flag, e.g. ss_group_links_by_source) and the device counter of the
// tests (ss_debug_csr_protocol_faults)
__device__ int csr_protocol_faults;
__device__ __forceinline__ bool wait_until(int32_t *word, int target, int *flag, int32_t *err, int32_t *fault_word, unsigned long long ticks)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 >= ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) {
            atomicAdd(&csr_protocol_faults, 1);
            // plain system-scope loads / stores, no read-modify-write on host memory (an atomic there needs PCIe AtomicOps end to end;
            // racing reporters only have to leave SOMETHING non-zero behind)
            if (err) __hip_atomic_store(err, __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) | SS_CSR_ERR_PROTOCOL, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_SYSTEM);
            if (fault_word)  // a stamp that differs from the one before (the clock moves on), never 0
                __hip_atomic_store(fault_word, (int32_t)((wall_clock64() >> 4) & 0x7FFFFFFF) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

// ---- finish over runs ------------------------------------------------------------------------------------------------------------
// A bucket of up to kWalkMax edges -- above the image, but not by much -- is still finished by its OWN workgroup, image by image (node
// ranges whose rows fit the image, every pass re-reading the bucket's records): under id-correlated skew most buckets that do not
// fit the image are of this kind (ogbl-ppa size, rank^-0.5 endpoints: ~135 of ~180), and two or three shares each through the
// count / place protocol cost them several times what one more sweep costs (round 6).  Only what is larger, or holds a single row
// above the image, is registered and worked off in shares.
constexpr int kWalkMax = 3 * kFinishCap;  // (what the tables are sized for; the launch picks 1 .. 3 images, see finish_walk_max)
constexpr int kRunBlocks = kWalkMax / kWave;  // 64-position blocks of the largest bucket / share a workgroup walks
static_assert(kWalkMax >= kDenseMin + kTile, "a share (at most kDensePart edges + one run) is walkable");
struct RunLds {
    uint32_t delta[kRunCap];           // first record of the run of each listed tile MINUS the run's first position: record = delta + position
    uint16_t start[kRunCap + 4];       // exclusive prefix of the run lengths: position of each run's first edge; [n] = total, then 0xFFFF
    uint16_t first_run[kRunBlocks + 2];  // the run that holds the first position of each 64-position block
    uint32_t wave_tot[kDenseThreads / kWave];
};
static_assert(kWalkMax < 65536 && kDensePart <= kDenseMin, "16-bit positions");

// The runs (tile, k) of the tiles [t_lo, t_hi) as an edge source for the finish step and the dense steps, walked by POSITION: the edges
// of the listed runs are numbered 0 .. total in tile order and thread i takes positions i, i + THREADS, ...: every lane has an edge
// whatever the run lengths are (16 lanes per 18-edge run left a quarter of them idle and a second dependent load for every run
// above the lane group: 271 us for the ppa-size finish), a wavefront's 64 positions are consecutive records of one or two runs, and
// a thread's loads are independent (eight in flight).  The run of a position: table lookup per 64-position block + up to three
// boundary compares.  Up to kRunCap descriptors are resident in LDS; a longer range (a bucket under a heavily skewed parent group)
// is walked batch by batch, reloading the descriptors in every for_each.  All threads call prepare / for_each / replay (barriers).
// The kernel is bound by its VALU instruction count (PMC: 116 instructions per edge in the first version, VALUBusy 55 %), hence
// the batches without bounds checks (FULL), the exact trip counts and the loop-free lookup.
template <bool PACKED, int THREADS>
struct RunEdges {
    const void *staged;
    const uint32_t *row0, *row1, *tstart;  // descriptor rows k and k + 1 (indexed by tile), tile starts (nullptr: t * kTile)
    int t_lo, t_hi;
    RunLds *lds;
    int src_bits, node0;
    static constexpr int kPer = kRunCap / THREADS;  // descriptors per thread in prepare

    __device__ __forceinline__ bool resident() const { return t_hi - t_lo <= kRunCap; }

    // descriptors of the tiles [b0, b0 + n), n <= kRunCap -> LDS; returns the number of edges in them
    // (o0_sum: the thread's run offsets inside their tiles are added -- summed over the tiles of a group that is the bucket's start)
    // prepare = scan + tables.  Split for the finish launch, which announces its decision (ordinary bucket / dense bucket) between the
    // two -- as soon as the edge count is known, ~2 us before the tables stand -- and registers a dense bucket's shares from the
    // prefix the scan left in its registers (no second pass over the descriptors)
    struct Scan {
        uint32_t total, ex;            // edges of the listed runs; position of this thread's first run (descriptor threadIdx.x * kPer)
        uint32_t len[kPer], addr[kPer];  // this thread's run lengths (0 past the end) and first records
    };
    __device__ __forceinline__ Scan scan(int b0, int n, unsigned long long *o0_sum = nullptr) const
    {
        const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
        // (every descriptor word is loaded UNCONDITIONALLY from a clamped index and only its use is predicated: loads under
        // `if (i < n)` are awaited branch by branch)
        Scan sc;
        uint32_t o0[kPer], o1[kPer], ts[kPer], run = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x * kPer + j;
            const int t = i < n ? b0 + i : 0;  // (word 0 of a descriptor row always exists)
            o0[j] = row0[t];
            o1[j] = row1[t];
            ts[j] = tstart ? tstart[t] : (uint32_t)t * (uint32_t)kTile;
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x * kPer + j;
            const bool in = i < n;
            sc.len[j] = in ? o1[j] - o0[j] : 0u;
            sc.addr[j] = in ? ts[j] + o0[j] : 0u;
            if (o0_sum && in) *o0_sum += o0[j];
            run += sc.len[j];
        }
        uint32_t inc = run;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t x = __shfl_up(inc, off);
            if (lane >= off) inc += x;
        }
        if (lane == kWave - 1) lds->wave_tot[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, total = 0;
#pragma unroll
        for (int w = 0; w < THREADS / kWave; ++w) {
            if (w < wv) pre += lds->wave_tot[w];
            total += lds->wave_tot[w];
        }
        sc.total = total;
        sc.ex = pre + inc - run;
        return sc;
    }
    __device__ __forceinline__ void tables(int n, const Scan &sc) const
    {
        uint32_t ex = sc.ex;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x * kPer + j;
            if (i < n) {
                lds->start[i] = (uint16_t)ex;
                lds->delta[i] = sc.addr[j] - ex;
            }
            ex += sc.len[j];
        }
        const uint32_t total = sc.total;
        if (threadIdx.x < 4) lds->start[n + threadIdx.x] = threadIdx.x == 0 ? (uint16_t)total : (uint16_t)0xFFFF;
        __syncthreads();
        // (a bucket above kDenseMin edges is not walked by its workgroup -- it goes to the dense steps, whose shares are walkable:
        // no table for it, its 16-bit starts are not used)
        const uint32_t walkable = total <= (uint32_t)(kRunBlocks * kWave) ? total : 0u;
        for (uint32_t q = threadIdx.x; q * kWave < walkable; q += THREADS) {  // last run that begins at or before position 64 q
            const uint32_t p0 = q * kWave;
            int a = 0, b = n;
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (lds->start[mid] <= p0) a = mid; else b = mid;
            }
            lds->first_run[q] = (uint16_t)a;
        }
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t prepare(int b0, int n, unsigned long long *o0_sum = nullptr) const
    {
        const Scan sc = scan(b0, n, o0_sum);
        tables(n, sc);
        return sc.total;
    }
    // the run of position p: the first run of p's 64-position block from the table (uniform over a wavefront), plus one for each of the
    // next three run starts at or before p -- three independent reads, no loop: the positions' chains interleave (twelve while-loops of
    // dependent LDS reads in a row were most of the 8.6 us a ppa-size bucket spent in a walk).  `more`: a fourth boundary may follow
    // (runs shorter than ~20 edges), the caller finishes with advance()
    __device__ __forceinline__ int run_of(uint32_t p, bool &more) const
    {
        const int r0 = lds->first_run[p / kWave];
        const uint32_t b1 = lds->start[r0 + 1], b2 = lds->start[r0 + 2], b3 = lds->start[r0 + 3];
        more |= b3 <= p;
        return r0 + (b1 <= p ? 1 : 0) + (b2 <= p ? 1 : 0) + (b3 <= p ? 1 : 0);
    }
    __device__ __forceinline__ int advance(int r, uint32_t p) const
    {
        while (lds->start[r + 1] <= p) ++r;
        return r;
    }
    // positions p0 + u THREADS + thread, u < U.  Three stages, none of them under a per-position branch: with `if (p < total)
    // { lookup; load; }` per position the compiler put s_waitcnt vmcnt(0) in front of every lookup -- U loads in a row, each waiting
    // for the one before.  FULL: every position of the batch exists (no bounds checks at all).
    // STASH (packed records, total <= kFinishCap): the records are also left in stash[position] -- see replay
    template <int U, bool FULL, bool STASH, typename F>
    __device__ __forceinline__ void batch(uint32_t p0, uint32_t total, uint32_t *stash, F &&f) const
    {
        uint32_t a[U], v32[U];
        int2 v64[U];
        int r[U];
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = p0 + u * THREADS + threadIdx.x;
            a[u] = FULL || p < total ? p : total - 1;  // (a valid position for the lanes past the end; its record is not used)
            r[u] = run_of(a[u], more);
        }
        if (__builtin_expect(more, 0)) {
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = advance(r[u], a[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] += lds->delta[r[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (PACKED) v32[u] = reinterpret_cast<const uint32_t *>(staged)[a[u]];
            else v64[u] = reinterpret_cast<const int2 *>(staged)[a[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = p0 + u * THREADS + threadIdx.x;
            if (FULL || p < total) {
                if (PACKED) {
                    if (STASH) stash[p] = v32[u];
                    f((int)(v32[u] & ((1u << src_bits) - 1u)), (int)(v32[u] >> src_bits));
                } else {
                    f(v64[u].x, v64[u].y - node0);
                }
            }
        }
    }
    template <bool STASH, typename F>
    __device__ __forceinline__ void walk(uint32_t total, uint32_t *stash, F &&f) const
    {
        constexpr int U = PACKED ? 8 : 4;  // loads in flight per thread
        uint32_t p0 = 0;
        for (; p0 + U * THREADS <= total; p0 += U * THREADS) batch<U, true, STASH>(p0, total, stash, f);
        for (; p0 + 2 * THREADS <= total; p0 += 2 * THREADS) batch<2, true, STASH>(p0, total, stash, f);
        if (p0 < total) batch<2, false, STASH>(p0, total, stash, f);
    }
    template <typename F>
    __device__ __forceinline__ void for_each_stash(uint32_t *stash, F &&f) const { walk<true>(lds->start[t_hi - t_lo], stash, f); }
    // the stashed records again; f may overwrite the stash array (every thread holds its records in registers behind a barrier)
    template <typename F>
    __device__ __forceinline__ void replay(uint32_t *stash, uint32_t total, F &&f) const
    {
        constexpr int U = kFinishCap / THREADS, C = 8;  // chunks of C positions: the ones past the bucket's end are skipped as a whole
        uint32_t v[U];
#pragma unroll
        for (int c = 0; c < U; c += C)
            if ((uint32_t)(c * THREADS) < total) {  // (uniform)
#pragma unroll
                for (int u = c; u < c + C; ++u) {
                    const uint32_t p = u * THREADS + threadIdx.x;
                    v[u] = stash[p < total ? p : 0];
                }
            }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < U; c += C)
            if ((uint32_t)(c * THREADS) < total) {
#pragma unroll
                for (int u = c; u < c + C; ++u) {
                    const uint32_t p = u * THREADS + threadIdx.x;
                    if (p < total) f((int)(v[u] & ((1u << src_bits) - 1u)), (int)(v[u] >> src_bits));
                }
            }
    }
    template <typename F>
    __device__ __forceinline__ void for_each(F &&f) const
    {
        if (resident()) {  // prepared by the caller
            walk<false>(lds->start[t_hi - t_lo], nullptr, f);
            return;
        }
        for (int b0 = t_lo; b0 < t_hi; b0 += kRunCap) {
            const int n = t_hi - b0 < kRunCap ? t_hi - b0 : kRunCap;
            __syncthreads();  // the walk of the batch before is done with the descriptors
            const uint32_t total = prepare(b0, n);
            walk<false>(total, nullptr, f);
        }
    }
};

// dense buckets of a level plan: shares are TILE ranges
struct DenseRunBucket {
    int32_t bucket, first_share, shares, t_hi;  // tiles of share s: [share_lo[first_share + s], s + 1 < shares ? share_lo[first_share + s + 1] : t_hi)
    uint32_t n;
    unsigned long long base;
};

struct DenseRunArgs {
    static constexpr int kMin = kDenseMin;
    int32_t *count;           // [0] registered buckets, [1] shares, [kHintWord] see above, [kArriveBase ...] arrivals
    DenseRunBucket *list;
    DenseSync *sync;          // [registered bucket]
    uint32_t *node_cnt;       // [dense bucket][1024]
    uint32_t *share_off;      // [share][1024]
    uint32_t *share_lo;       // [share] first tile of each share
    int32_t *claim;           // [share] 0; 2: counted, its records no longer held by anybody (an orphan); 1: an orphan somebody is placing
    const uint32_t *row0, *row1;  // descriptor rows of the registering bucket (set per workgroup)
    int t_lo, t_hi;

    // the registration is complete: publish it, then count this bucket's workgroup as arrived (G16 producer form)
    __device__ __forceinline__ void publish(int d, int shares) const
    {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&sync[d].pub, shares, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&count[kArriveBase + 16 * (blockIdx.x % kArriveWords)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the bucket's runs in tile order: a share ends behind the tile in which the running edge count crosses a multiple of kDensePart
    __device__ __forceinline__ int register_bucket(FinishLds &lds, unsigned long long seg_lo, uint32_t seg_n, int nb) const
    {
        const int shares = (int)((seg_n + kDensePart - 1) / kDensePart);
        if (threadIdx.x == 0) {
            const int d = atomicAdd(&count[0], 1);
            const int first = atomicAdd(&count[1], shares);
            list[d] = DenseRunBucket{(int32_t)blockIdx.x, first, shares, t_hi, seg_n, seg_lo};
            share_lo[first] = (uint32_t)t_lo;
            lds.excl[0] = (uint32_t)d;
            lds.excl[1] = (uint32_t)first;
        }
        __syncthreads();
        const int d = (int)lds.excl[0], first = (int)lds.excl[1];
        uint32_t *mine = node_cnt + (size_t)d * 1024;
        for (int i = threadIdx.x; i < nb; i += (int)blockDim.x) mine[i] = 0;
        for (int i = threadIdx.x; i < shares; i += (int)blockDim.x) claim[first + i] = 0;
        const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
        uint32_t carry = 0;
        for (int t0 = t_lo; t0 < t_hi; t0 += (int)blockDim.x) {
            const int t = t0 + threadIdx.x;
            const int tc = t < t_hi ? t : 0;  // (unconditional loads from a clamped index, see RunEdges::prepare)
            const uint32_t d0 = row0[tc], d1 = row1[tc];
            const uint32_t len = t < t_hi ? d1 - d0 : 0u;
            uint32_t inc = len;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t x = __shfl_up(inc, off);
                if (lane >= off) inc += x;
            }
            __syncthreads();  // (wave_tot of the round before has been read)
            if (lane == kWave - 1) lds.wave_tot[wv] = inc;
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int w = 0; w < (int)blockDim.x / kWave; ++w) {
                if (w < wv) pre += lds.wave_tot[w];
                tot += lds.wave_tot[w];
            }
            const uint32_t after = carry + pre + inc, before = after - len;
            const uint32_t s_hi = after / (uint32_t)kDensePart, s_lo = before / (uint32_t)kDensePart;
            if (s_hi != s_lo && s_hi < (uint32_t)shares) share_lo[first + s_hi] = (uint32_t)(t + 1);  // (a run is at most a tile: one crossing)
            carry += tot;
        }
        publish(d, shares);
        return d;
    }
    // the same registration from what RunEdges::prepare left in the registers of the bucket's workgroup (thread i: the position `ex`
    // of the run of tile t_lo + i * KPER and the lengths of its KPER runs): no second pass over the descriptors, no second scan
    // (rank^-0.5 endpoints at collab size: the dense buckets are registered 7.5 us into the launch instead of 12.5)
    template <int KPER>
    __device__ __forceinline__ int register_from_prefix(FinishLds &lds, unsigned long long seg_lo, uint32_t seg_n, int nb, uint32_t ex,
                                                         const uint32_t (&len)[KPER]) const
    {
        const int shares = (int)((seg_n + kDensePart - 1) / kDensePart);
        if (threadIdx.x == 0) {
            const int d = atomicAdd(&count[0], 1);
            const int first = atomicAdd(&count[1], shares);
            list[d] = DenseRunBucket{(int32_t)blockIdx.x, first, shares, t_hi, seg_n, seg_lo};
            share_lo[first] = (uint32_t)t_lo;
            lds.excl[0] = (uint32_t)d;
            lds.excl[1] = (uint32_t)first;
        }
        __syncthreads();
        const int d = (int)lds.excl[0], first = (int)lds.excl[1];
        uint32_t *mine = node_cnt + (size_t)d * 1024;
        for (int i = threadIdx.x; i < nb; i += (int)blockDim.x) mine[i] = 0;
        for (int i = threadIdx.x; i < shares; i += (int)blockDim.x) claim[first + i] = 0;
#pragma unroll
        for (int j = 0; j < KPER; ++j) {
            const int t = t_lo + (int)threadIdx.x * KPER + j;
            const uint32_t after = ex + len[j];  // (len is 0 for the descriptors past t_hi)
            const uint32_t s_hi = after / (uint32_t)kDensePart, s_lo = ex / (uint32_t)kDensePart;
            if (s_hi != s_lo && s_hi < (uint32_t)shares && t < t_hi) share_lo[first + s_hi] = (uint32_t)(t + 1);  // (a run is at most a tile: one crossing)
            ex = after;
        }
        publish(d, shares);
        return d;
    }
    // a bucket that is finished by its own workgroup has decided too (the dedicated helpers leave once every bucket has)
    __device__ __forceinline__ void arrive() const
    {
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(&count[kArriveBase + 16 * (blockIdx.x % kArriveWords)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// LDS of a workgroup while it works on other buckets' shares (aliases the finish step's col image)
struct DenseRunLds {
    uint32_t cnt[1024], excl[1024 + 1];
    uint32_t wave_tot[kDenseThreads / kWave];
    RunLds runs;
    int desc, claim, pick, n_dense, unpublished, flag;
};
static_assert(sizeof(DenseRunLds) <= sizeof(int32_t) * kFinishCap, "the helpers' LDS aliases the finish step's col image");

struct DenseRunWork {  // what the dense steps read
    ParentLevel par;
    const void *staged;
    int node_shift, src_bits;
    int64_t N;
    const DenseRunBucket *list;
    const uint32_t *share_lo;
    uint32_t *node_cnt, *share_off;
    int32_t *col;
};

// (all threads; barriers) the tile range of share `item` of the registered bucket b; the descriptors of the range (or of its first batch) -> LDS
template <bool PACKED, int THREADS>
__device__ __forceinline__ RunEdges<PACKED, THREADS> locate_run_share(DenseRunLds &lds, int item, const DenseRunBucket &b, const DenseRunWork &w)
{
    __syncthreads();  // the previous share's readers of lds are done
    const int s = item - b.first_share;
    const int t_lo = (int)w.share_lo[item], t_hi = s + 1 < b.shares ? (int)w.share_lo[item + 1] : b.t_hi;
    const ChildGroup c = child_group(w.par, b.bucket);
    const uint32_t *row0 = w.par.off + (int64_t)c.k * w.par.tmax, *row1 = row0 + w.par.tmax;
    RunEdges<PACKED, THREADS> e{w.staged, row0, row1, w.par.tstart, t_lo, t_hi, &lds.runs, w.src_bits, (int)((int64_t)b.bucket << w.node_shift)};
    if (e.resident()) e.prepare(t_lo, t_hi - t_lo);
    return e;
}

// a counted share whose records nobody holds any more: row starts from the summed counters (the first share of a bucket also
// publishes rowptr and the hub lists), then every edge of the share goes to start + share offset + LDS cursor.  (The counters and
// share offsets were written by other workgroups of THIS launch: they are read past the L1 with agent-scope loads.)
template <bool PACKED, int THREADS>
__device__ __forceinline__ void dense_place_run_share(DenseRunLds &lds, int item, int d, const DenseRunBucket &b, const DenseRunWork &w, const RowOutputs &o)
{
    const int nb = 1 << w.node_shift;
    const RunEdges<PACKED, THREADS> edges = locate_run_share<PACKED, THREADS>(lds, item, b, w);
    const uint32_t *total = w.node_cnt + (size_t)d * 1024;
    for (int i = threadIdx.x; i < nb; i += THREADS)
        lds.cnt[i] = __hip_atomic_load(const_cast<uint32_t *>(total) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    scan_bucket_nodes(lds.cnt, lds.excl, lds.wave_tot, nb, (int64_t)b.bucket << w.node_shift, w.N, b.base, b.n, item == b.first_share, o);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += THREADS)
        lds.cnt[i] = lds.excl[i] + __hip_atomic_load(w.share_off + (size_t)item * 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long cbase = b.base;
    int32_t *col = w.col;
    edges.for_each([&](int x, int y) { col[cbase + atomicAdd(&lds.cnt[y], 1u)] = x; });
}

// ---- working off the registered buckets (every workgroup of the finish launch, see "who may wait for whom" above) -------------------
// A workgroup keeps the records of the LAST share it counted in LDS behind its tables and places that one from there -- no second
// descriptor table, no second gather; with one share per workgroup, the usual case, that is every share (placing step 13.4 -> 7.5 us
// at rank^-0.5, collab size).  A share it counted BEFORE that one is marked an orphan (claim word 2) and placed by whoever draws its
// ticket after the bucket is counted -- from global memory, as dense_place_run_share does.
constexpr int kHelperStashWord = (int)((sizeof(DenseRunLds) + 255) / 256 * 256 / 4);
constexpr int kHelperStashCap = kFinishCap - kHelperStashWord;  // (a share is at most kDensePart edges + one run: 12 288 records)
static_assert(kHelperStashCap >= kDensePart + kTile, "a share fits behind the tables");

// all shares of registered bucket d that nobody has claimed yet, then the placing of what this workgroup counted; returns whether
// it got a share at all.  On return the bucket has no unclaimed share left.
template <bool PACKED>
__device__ __forceinline__ bool dense_help_bucket(DenseRunLds &lds, uint32_t *stash, int d, const DenseRunWork &w, const RowOutputs &o,
                                                  const DenseRunArgs &dense)
{
    const DenseRunBucket b = w.list[d];  // (published: behind the acquire of the scan that picked d)
    DenseSync *sy = dense.sync + d;
    const int nb = 1 << w.node_shift;
    constexpr int kPerThread = 1024 / kRunThreads;  // node counters per thread
    auto take = [&](int32_t *word) {  // (all threads) the next ticket of an agent-scope counter
        __syncthreads();
        if (threadIdx.x == 0) lds.claim = __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return lds.claim;
    };
    // ---- count
    int mine = -1;  // the share whose records (and run tables) this workgroup still holds
    RunEdges<PACKED, kRunThreads> edges = {};
    uint32_t total = 0, off[kPerThread] = {};
    bool stashed = false;
    lds.desc = d;
    for (;;) {
        const int s = take(&sy->next);
        if (s >= b.shares) break;  // workgroup-uniform
        if (s == b.shares - 1 && threadIdx.x == 0)  // the bucket's last share: nobody needs to look at it again
            __hip_atomic_fetch_add(&dense.count[kSpentWord], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine >= 0 && threadIdx.x == 0) {  // the records of the share before are about to be overwritten: somebody else places it
            __hip_atomic_store(&dense.claim[mine], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&sy->orphans, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (visible before this share's `counted`: release below)
        }
        const int item = b.first_share + s;
        edges = locate_run_share<PACKED, kRunThreads>(lds, item, b, w);
        total = edges.resident() ? lds.runs.start[edges.t_hi - edges.t_lo] : 0u;
        stashed = PACKED && edges.resident() && total <= (uint32_t)kHelperStashCap;
        for (int i = threadIdx.x; i < nb; i += kRunThreads) lds.cnt[i] = 0;
        __syncthreads();
        if (stashed) edges.for_each_stash(stash, [&](int, int y) { atomicAdd(&lds.cnt[y], 1u); });
        else edges.for_each([&](int, int y) { atomicAdd(&lds.cnt[y], 1u); });
        __syncthreads();
        uint32_t *sum = w.node_cnt + (size_t)d * 1024;
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = threadIdx.x + k * kRunThreads;
            const uint32_t c = i < nb ? lds.cnt[i] : 0u;
            off[k] = c ? atomicAdd(&sum[i], c) : 0u;  // (where this share's edges of node i start inside the node's row)
            if (i < nb) w.share_off[(size_t)item * 1024 + i] = off[k];  // (for whoever places the share, should it not be this workgroup)
        }
        mine = item;
        __syncthreads();
        if (threadIdx.x == 0) {  // this share is counted
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&sy->counted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (mine < 0) return false;
    SS_MARK(11);
    // every share of the bucket is claimed (the counter ran out above), each by a workgroup that is inside its count step: safe to wait
    if (!wait_until(&sy->counted, b.shares, &lds.flag, o.err, o.fault_word, o.wait_ticks)) return true;
    SS_MARK(12);
    // ---- place: the share whose records are still here ...
    {
        const uint32_t *sum = w.node_cnt + (size_t)d * 1024;
        for (int i = threadIdx.x; i < nb; i += kRunThreads)
            lds.cnt[i] = __hip_atomic_load(const_cast<uint32_t *>(sum) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) lds.claim = __hip_atomic_load(&sy->orphans, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        scan_bucket_nodes(lds.cnt, lds.excl, lds.wave_tot, nb, (int64_t)b.bucket << w.node_shift, w.N, b.base, b.n, mine == b.first_share, o);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = threadIdx.x + k * kRunThreads;
            if (i < nb) lds.cnt[i] = lds.excl[i] + off[k];
        }
        const int orphans = lds.claim;
        __syncthreads();
        const unsigned long long cbase = b.base;
        int32_t *col = w.col;
        auto place = [&](int x, int y) { col[cbase + atomicAdd(&lds.cnt[y], 1u)] = x; };
        if (stashed) edges.replay(stash, total, place);
        else edges.for_each(place);
        if (orphans == 0) return true;  // (workgroup-uniform) the usual case: every share is placed by the workgroup that counted it
    }
    // ---- ... then the orphans, by ticket
    for (;;) {
        const int s = take(&sy->pnext);
        if (s >= b.shares) break;
        __syncthreads();
        if (threadIdx.x == 0) {
            int expect = 2;
            lds.claim = __hip_atomic_compare_exchange_strong(&dense.claim[b.first_share + s], &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        }
        __syncthreads();
        if (lds.claim) dense_place_run_share<PACKED, kRunThreads>(lds, b.first_share + s, d, b, w, o);  // (workgroup-uniform)
    }
    return true;
}

// Looks over the registered buckets, 64 at a time (lane l of wave 0 takes bucket d0 + l), until a whole round over the windows finds
// nothing to claim.  WHICH bucket of a window a workgroup goes for is drawn at random, weighted by the shares each bucket still
// has to give: the first version sent everybody to the FIRST open bucket -- with 22 dense buckets (collab size, rank^-0.5) 128 helpers
// queued through them one failed claim (~4 us: look, ticket, barriers) after the other, 84 us instead of 39 for the launch; with the
// ~500 dense buckets of a ppa-size graph 1.3 ms instead of 0.21 -- and the window a round starts with is drawn too.
// count[kSpentWord] counts the buckets whose last share has been taken: a workgroup that finds it equal to the number of registered
// buckets has nothing to look for (one round trip at the end of every bucket workgroup of a skewed graph instead of a scan).
// dedicated: a helper workgroup -- it keeps polling until every bucket workgroup has arrived (or its patience ends: it is an
// accelerator, not a participant anybody depends on)
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

template <bool PACKED>
__device__ __forceinline__ void dense_help(DenseRunLds &lds, uint32_t *stash, bool dedicated, int n_buckets, int own, const DenseRunWork &w,
                                           const RowOutputs &o, const DenseRunArgs &dense)
{
    const unsigned long long t_start = dedicated ? wall_clock64() : 0ULL;
    uint32_t draws = blockIdx.x * 0x9E3779B9u;
    int forced = own;  // own >= 0: this workgroup has just registered bucket `own` -- its first pick, without a look (ONE call site of dense_help_bucket: a second inlined copy costs 56 bytes of scratch)
    for (;;) {  // rounds
        bool progress = false;
        __syncthreads();
        if (threadIdx.x == 0) {
            lds.unpublished = 0;
            lds.pick = __hip_atomic_load(&dense.count[kHintWord], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds.n_dense = __hip_atomic_load(&dense.count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds.claim = __hip_atomic_load(&dense.count[kSpentWord], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int hint = lds.pick, registered = lds.n_dense, spent = lds.claim;
        if (spent < registered) {  // (workgroup-uniform) something may be left to claim
            const int windows = (registered - hint + kWave - 1) / kWave;
            const int w0 = windows > 1 ? (int)(mix32(draws += 0x632BE5ABu) % (uint32_t)windows) : 0;
            int spent_end = hint;  // (thread 0's copy counts) every registered bucket before it has no unclaimed share left
            int idle = 0, win = w0;  // windows in a row without anything to claim; the window being looked at
            for (int guard = 0; idle < windows && guard < 64 * windows + 64; ++guard) {  // (the second bound: a round of lost draws ends, the next begins)
                const int d0 = hint + kWave * win;
                const uint32_t draw = mix32(draws += 0x632BE5ABu);
                __syncthreads();
                if (forced >= 0) {  // (workgroup-uniform)
                    if (threadIdx.x == 0) lds.pick = forced;
                    forced = -1;
                } else if (threadIdx.x < kWave) {
                    const int lane = threadIdx.x;
                    const int n_dense = __hip_atomic_load(&dense.count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int d = d0 + lane;
                    // two independent loads, one round trip: the published word IS the bucket's share count, and its claim counter has
                    // been zero since the tile sort of this build
                    const int shares = d < n_dense ? __hip_atomic_load(&dense.sync[d].pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    const int nxt = d < n_dense ? __hip_atomic_load(&dense.sync[d].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (before anybody reads the descriptor of a bucket picked here)
                    const bool pub = shares != 0;
                    const int left = pub && nxt < shares ? shares - nxt : 0;  // shares this bucket still has to give
                    int inc = left;
#pragma unroll
                    for (int off = 1; off < kWave; off <<= 1) {
                        const int x = __shfl_up(inc, off);
                        if (lane >= off) inc += x;
                    }
                    const int total = __shfl(inc, kWave - 1);
                    // the bucket whose stretch of the running sum holds the draw: a workgroup lands on a bucket with probability
                    // proportional to what is left there
                    const int x = total > 0 ? (int)(draw % (uint32_t)total) : 0;
                    const unsigned long long here = __ballot(left > 0 && inc > x);
                    const unsigned long long unpub = __ballot(d < n_dense && !pub);
                    const unsigned long long done = __ballot(pub && nxt >= shares);
                    if (lane == 0) {
                        lds.pick = total > 0 ? d0 + __builtin_ctzll(here) : -1;
                        if (unpub) lds.unpublished = 1;
                        // the leading buckets of this window that have no unclaimed share left, if everything before the window is in
                        // the same state: later rounds (anybody's) start behind them
                        const int lead = ~done ? __builtin_ctzll(~done) : kWave;
                        if (d0 == spent_end && lead > 0) {
                            spent_end = d0 + lead;
                            atomicMax(&dense.count[kHintWord], spent_end);
                        }
                    }
                }
                __syncthreads();
                const int pick = lds.pick;
                if (pick < 0) {
                    ++idle;
                    win = win + 1 < windows ? win + 1 : 0;
                    continue;
                }
                if (dense_help_bucket<PACKED>(lds, stash, pick, w, o, dense)) progress = true;
                idle = 0;  // (the same window again while it has something to give)
            }
        }
        if (!dedicated) {
            if (!progress) return;  // a bucket registered later is worked off by its own workgroup and whoever finishes after it
            continue;
        }
        // a dedicated helper: done once every bucket workgroup has decided and a round found nothing to claim
        __syncthreads();
        if (threadIdx.x < kWave) {
            int v = __hip_atomic_load(&dense.count[kArriveBase + 16 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (threadIdx.x == 0) {
                lds.claim = v;
                lds.flag = wall_clock64() - t_start > kHelperPatienceTicks ? 1 : 0;
            }
        }
        __syncthreads();
        const bool all_arrived = lds.claim >= n_buckets, impatient = lds.flag != 0;  // (one thread looked: workgroup-uniform)
        if (all_arrived && !progress && !lds.unpublished) return;
        if (!progress) {
            if (impatient) return;
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

template <bool PACKED>
__global__ __launch_bounds__(kRunThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void finish_runs_kernel(ParentLevel par, const void *__restrict__ staged,
                                                                     const unsigned long long *__restrict__ tile_max, int tiles0, int node_shift,
                                                                     int src_bits, int64_t N, int32_t *__restrict__ col,
                                                                     unsigned long long *__restrict__ n_self, RowOutputs o, DenseRunArgs dense,
                                                                     int64_t fine_buckets)
{
    __shared__ FinishLds lds;
    __shared__ RunLds runs;
    __shared__ unsigned long long red_base[kRunThreads / kWave], red_max[kRunThreads / kWave];
    __shared__ uint32_t red_n[kRunThreads / kWave];
    SS_CSR_SKIP(o.skip);
    SS_MARK_START();
    const DenseRunWork work = {par, staged, node_shift, src_bits, N, dense.list, dense.share_lo, dense.node_cnt, dense.share_off, col};
    DenseRunLds &help_lds = *reinterpret_cast<DenseRunLds *>(lds.image);
    uint32_t *help_stash = reinterpret_cast<uint32_t *>(lds.image) + kHelperStashWord;
    const bool dedicated = (int64_t)blockIdx.x >= fine_buckets;  // a helper workgroup: no bucket of its own
    int own = -1;                                                // the registered bucket of this workgroup, if its bucket is dense
    // ONE call site of dense_help at the end of the kernel (three inlined copies: 166 VGPRs, one workgroup per CU instead of two)
    if (!dedicated) {
    SS_TICK_START();
    const ChildGroup c = child_group(par, blockIdx.x);
    const uint32_t *row0 = par.off + (int64_t)c.k * par.tmax, *row1 = row0 + par.tmax;
    const int node0 = (int)((int64_t)blockIdx.x << node_shift);
    const RunEdges<PACKED, kRunThreads> edges{staged, row0, row1, par.tstart, c.t_lo, c.t_hi, &runs, src_bits, node0};
    unsigned long long base = 0, mx = 0;
    uint32_t n = 0;
    const bool resident = edges.resident();  // (workgroup-uniform) one descriptor per thread: loaded once, for the sums and the walks
    typename RunEdges<PACKED, kRunThreads>::Scan sc = {};
    bool announced = false;
    dense.t_lo = c.t_lo;
    dense.t_hi = c.t_hi;
    if (resident) {
        // the decision (ordinary / dense) is announced as soon as the edge count is known, before the run tables are built: the
        // dedicated helpers leave when every bucket has decided and nothing is registered
        sc = edges.scan(c.t_lo, c.t_hi - c.t_lo, &base);
        announced = true;
        if (sc.total <= o.walk_max) {  // (workgroup-uniform) finished here: in one image, or image by image
            dense.arrive();
            edges.tables(c.t_hi - c.t_lo, sc);
        }
        const uint32_t total = sc.total;
        n = threadIdx.x == 0 ? total : 0u;
    } else {
        for (int t = c.t_lo + threadIdx.x; t < c.t_hi; t += kRunThreads) {
            const uint32_t o0 = row0[t];
            base += o0;
            n += row1[t] - o0;
        }
    }
    // the LAST bucket's workgroup also reduces max(edge_index) + 1 (the self-loop count, hashing.py:148) -- not the first: under
    // id-correlated skew bucket 0 is the dense one everybody else helps with
    const bool reducer = (int64_t)blockIdx.x == fine_buckets - 1;
    if (reducer)
        for (int t = threadIdx.x; t < tiles0; t += kRunThreads) {
            const unsigned long long v = tile_max[t];
            mx = v > mx ? v : mx;
        }
    for (int off = kWave / 2; off > 0; off >>= 1) {
        base += __shfl_xor(base, off);
        n += __shfl_xor(n, off);
        const unsigned long long x = __shfl_xor(mx, off);
        mx = x > mx ? x : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red_base[threadIdx.x / kWave] = base;
        red_n[threadIdx.x / kWave] = n;
        red_max[threadIdx.x / kWave] = mx;
    }
    __syncthreads();
    base = c.base, n = 0, mx = 0;
    for (int w = 0; w < kRunThreads / kWave; ++w) {
        base += red_base[w];
        n += red_n[w];
        mx = red_max[w] > mx ? red_max[w] : mx;
    }
    if (threadIdx.x == 0 && reducer) {
        *n_self = mx;
        o.rowptr[N] = (int64_t)(base + n);  // the last bucket ends the edge list
    }
    dense.row0 = row0;
    dense.row1 = row1;
    dense.t_lo = c.t_lo;
    dense.t_hi = c.t_hi;
    SS_TICK(0);
    const int nb = 1 << node_shift;  // <= 1024 nodes
    uint32_t *cnt = lds.cnt;
    // above the image but walkable (resident descriptors, <= walk_max edges): counted here first -- it stays with this workgroup unless
    // ONE row alone exceeds the image (then it is registered like the larger ones; nothing has been published yet)
    bool by_images = false;  // (workgroup-uniform)
    if (n > (uint32_t)kDenseMin && n <= o.walk_max && resident) {
        for (int i = threadIdx.x; i < nb; i += kRunThreads) cnt[i] = 0;
        __syncthreads();
        edges.for_each([&](int, int y) { atomicAdd(&cnt[y], 1u); });
        __syncthreads();
        uint32_t big = 0;
        for (int i = threadIdx.x; i < nb; i += kRunThreads) big = cnt[i] > big ? cnt[i] : big;
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const uint32_t x = __shfl_xor(big, off);
            big = x > big ? x : big;
        }
        if ((threadIdx.x & (kWave - 1)) == 0) red_n[threadIdx.x / kWave] = big;
        __syncthreads();
        big = 0;
        for (int w = 0; w < kRunThreads / kWave; ++w) big = red_n[w] > big ? red_n[w] : big;
        by_images = big <= (uint32_t)kFinishCap;
        __syncthreads();
    }
    if (n > (uint32_t)kDenseMin && !by_images) {  // (workgroup-uniform) registered, then worked off share by share by everybody
        own = announced ? dense.register_from_prefix(lds, base, n, nb, sc.ex, sc.len) : dense.register_bucket(lds, base, n, nb);
        SS_MARK(14);
    } else {
    if (!announced) dense.arrive();
    SS_MARK(15);
    if (!by_images) {
    for (int i = threadIdx.x; i < nb; i += kRunThreads) cnt[i] = 0;
    __syncthreads();
    }
    // (workgroup-uniform) packed records: the counting sweep leaves them in the image array and the placing sweep takes them from
    // there -- no second gather, no second run lookup (ppa-size finish: 9.3 us of 26.7 per workgroup)
    const bool stashed = PACKED && edges.resident() && !by_images;
    uint32_t *stash = reinterpret_cast<uint32_t *>(lds.image);
    if (by_images) {}  // (counted above)
    else if (stashed) edges.for_each_stash(stash, [&](int, int y) { atomicAdd(&cnt[y], 1u); });
    else edges.for_each([&](int, int y) { atomicAdd(&cnt[y], 1u); });
    __syncthreads();
    SS_TICK(1);
    scan_bucket_nodes(cnt, lds.excl, lds.wave_tot, nb, (int64_t)node0, N, base, n, true, o);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += kRunThreads) cnt[i] = lds.excl[i];  // cursors
    __syncthreads();
    SS_TICK(2);
    // anything registered by now?  Loaded here, used behind the bucket's last stores (the load's latency hides under the placing
    // sweep); a bucket registered later than this look is worked off by its own workgroup and by whoever finishes later
    const int registered = __hip_atomic_load(&dense.count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                           __hip_atomic_load(&dense.count[kSpentWord], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (registered, not yet given away)
    if (by_images) {
        // node ranges [lo, hi) whose rows together fit the image, one after the other: every pass walks the bucket's records again
        // (they were read a moment ago: L2) and keeps the edges of its nodes -- two passes for a bucket of up to two images, a
        // third when the cut falls badly
        for (int lo = 0; lo < nb;) {  // (workgroup-uniform)
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t lim = lds.excl[lo] + (uint32_t)kFinishCap;  // (every row fits: checked above, so hi > lo)
                int a = lo + 1, b = nb;
                while (a < b) {
                    const int mid = (a + b + 1) >> 1;
                    if (lds.excl[mid] <= lim) a = mid; else b = mid - 1;
                }
                red_n[1] = (uint32_t)a;
            }
            __syncthreads();
            const int hi = (int)red_n[1];
            const uint32_t e0 = lds.excl[lo], e1 = lds.excl[hi];
            for (int i = lo + (int)threadIdx.x; i < hi; i += kRunThreads) cnt[i] = lds.excl[i] - e0;  // cursors inside the image
            __syncthreads();
            edges.for_each([&](int x, int y) {
                if (y >= lo && y < hi) lds.image[atomicAdd(&cnt[y], 1u)] = x;
            });
            __syncthreads();
            for (uint32_t q = threadIdx.x; q < e1 - e0; q += kRunThreads) col[base + e0 + q] = lds.image[q];
            lo = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) red_n[0] = (uint32_t)registered;
    } else {
    auto place = [&](int x, int y) { lds.image[atomicAdd(&cnt[y], 1u)] = x; };
    if (stashed) edges.replay(stash, n, place);
    else edges.for_each(place);
    __syncthreads();
    SS_TICK(3);
    if (threadIdx.x == 0) red_n[0] = (uint32_t)registered;  // (one thread's look decides for the workgroup)
    for (uint32_t q = threadIdx.x; q < n; q += kRunThreads) col[base + q] = lds.image[q];
    }
    SS_TICK(4);
    SS_MARK(9);
    __syncthreads();  // the image is free (it becomes the helper's LDS), the look is visible
    if ((int)red_n[0] <= 0) return;  // every unskewed graph; a skewed one whose dense buckets have all been taken
    }
    }
    dense_help<PACKED>(help_lds, help_stash, dedicated, (int)fine_buckets, own, work, o, dense);
    SS_MARK(13);
}

struct Workspace {
    LevelArrays lv[kMaxLevels];
    int2 *staged_a;                 // level 0 (tile j at j * kTile) and level 2
    void *staged_b;                 // level 1
    unsigned long long *tile_max;   // [tiles0]
    unsigned long long *scratch;    // [1] n_self when the caller does not want it
    int32_t *dense_count;           // [kDenseSyncInts]
    DenseRunBucket *dense_list;
    uint32_t *dense_node_cnt, *dense_share_off, *dense_share_lo;
    int32_t *dense_claim;           // [shares] (who places a share)
    DenseSync *dense_sync;          // [dense buckets]
    size_t bytes;
};

inline Workspace carve(const LevelPlan &p, int64_t E, void *base)
{
    Workspace w;
    char *c = reinterpret_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t n) { char *r = c ? c + off : nullptr; off += align256(n); return r; };
    for (int l = 0; l < p.levels; ++l) {
        LevelArrays &a = w.lv[l];
        a.off = reinterpret_cast<uint32_t *>(take((size_t)(p.keys[l] + 1) * p.tmax[l] * 4));
        a.prefix = reinterpret_cast<uint32_t *>(take(l + 1 < p.levels ? (size_t)p.keys[l] * p.tmax[l] * 4 : 0));
        a.cfirst = reinterpret_cast<uint32_t *>(take(l + 1 < p.levels ? (size_t)p.keys[l] * p.tmax[l] * 4 : 0));
        const bool sub = l >= 1;
        a.tstart = reinterpret_cast<uint32_t *>(take(sub ? (size_t)p.tmax[l] * 4 : 0));
        a.tb = reinterpret_cast<uint32_t *>(take(sub ? (size_t)(p.groups[l] + 1) * 4 : 0));
        a.header = reinterpret_cast<TileHeader *>(take(sub ? (size_t)p.tmax[l] * sizeof(TileHeader) : 0));
        a.gcount = reinterpret_cast<uint32_t *>(take(sub ? (size_t)p.groups[l] * 4 : 0));
        a.n_tiles = reinterpret_cast<uint32_t *>(take(sub ? 4 : 0));
        a.base = reinterpret_cast<unsigned long long *>(take(sub ? (size_t)p.groups[l] * 8 : 0));
    }
    w.staged_a = reinterpret_cast<int2 *>(take((size_t)p.tmax[0] * kTile * 8));
    w.staged_b = take(p.levels >= 2 ? (size_t)(E > 0 ? E : 1) * 8 : 0);
    w.tile_max = reinterpret_cast<unsigned long long *>(take((size_t)p.tmax[0] * 8));
    w.scratch = reinterpret_cast<unsigned long long *>(take(8));
    w.dense_count = reinterpret_cast<int32_t *>(take(4 * kDenseSyncInts));
    const int64_t db = max_dense_buckets(E), ds = max_dense_shares(E);
    w.dense_list = reinterpret_cast<DenseRunBucket *>(take((size_t)db * sizeof(DenseRunBucket)));
    w.dense_node_cnt = reinterpret_cast<uint32_t *>(take((size_t)db * 1024 * 4));
    w.dense_share_off = reinterpret_cast<uint32_t *>(take((size_t)ds * 1024 * 4));
    w.dense_share_lo = reinterpret_cast<uint32_t *>(take((size_t)(ds + 1) * 4));
    w.dense_claim = reinterpret_cast<int32_t *>(take((size_t)(ds + 1) * 4));
    w.dense_sync = reinterpret_cast<DenseSync *>(take((size_t)db * sizeof(DenseSync)));
    w.bytes = off;
    return w;
}

// ---- content fingerprint of an edge list (ss_csr_build_cached) ---------------------------------------------------------------
// ELPH.forward concatenates a fresh self-looped edge_index every training step (reference models/elph.py:186) -- the same
// edges in a new tensor -- and the CSR of it was rebuilt every step (43 us of a 0.30 ms step).  The cached form streams the
// edge list once (two independent 64-bit sums of a 64-bit mix of every (src, dst): order-independent, like the CSR's meaning),
// compares them ON THE DEVICE with the sums stored beside the CSR by the build that made it, and lets every kernel of the build
// exit when they agree.  No host read; an edited or recycled edge_index has other sums and is rebuilt.
constexpr int kFpBlocks = 512;
struct FingerprintWords {  // layout of the caller's device buffer (SS_CSR_FINGERPRINT_BYTES)
    unsigned long long stored[2];
    int32_t valid, skip;
    int32_t bad, pad;  // the build that produced the cached CSR met ids out of range (re-reported when a later call is skipped)
    unsigned long long partial[kFpBlocks][2];
};
static_assert(sizeof(FingerprintWords) <= SS_CSR_FINGERPRINT_BYTES, "SS_CSR_FINGERPRINT_BYTES");

__global__ __launch_bounds__(256) void fingerprint_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t E,
                                                          FingerprintWords *__restrict__ fp)
{
    __shared__ unsigned long long red[2][256 / kWave];
    unsigned long long a = 0, b = 0;
    // per edge: one 64-bit mix m of (src, dst) -- injective in (s, d) for ids below 2^32 before the finaliser -- summed as is and
    // summed once more through a second multiply-xorshift (two sums of different functions of m: an edit has to preserve both).
    // Two edges per lane and load (16-byte loads of both rows), two such loads in flight: 11.3 -> 7 us for 2.6 M edges
    auto add = [&](uint64_t sv, uint64_t dv) {
        const uint64_t m = hash_u64(sv * 0x9E3779B97F4A7C15ULL + dv + 0x632BE59BD9B4E019ULL);
        a += m;
        uint64_t m2 = (m ^ (m >> 29)) * 0xC2B2AE3D27D4EB4FULL;
        b += m2 ^ (m2 >> 32);
    };
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int64_t pairs = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) ? 0 : E / 2;  // 16-byte aligned rows only
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u64x2 *src2 = reinterpret_cast<const u64x2 *>(src), *dst2 = reinterpret_cast<const u64x2 *>(dst);
    for (; q + stride < pairs; q += 2 * stride) {
        const u64x2 s0 = src2[q], d0 = dst2[q], s1 = src2[q + stride], d1 = dst2[q + stride];
        add(s0.x, d0.x);
        add(s0.y, d0.y);
        add(s1.x, d1.x);
        add(s1.y, d1.y);
    }
    for (; q < pairs; q += stride) {
        const u64x2 s0 = src2[q], d0 = dst2[q];
        add(s0.x, d0.x);
        add(s0.y, d0.y);
    }
    for (int64_t e = 2 * pairs + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) add((uint64_t)src[e], (uint64_t)dst[e]);
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[0][threadIdx.x / kWave] = a;
        red[1][threadIdx.x / kWave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        fp->partial[blockIdx.x][0] = a;
        fp->partial[blockIdx.x][1] = b;
    }
}

// one workgroup: sums of the partials (+ the shape, so that another N / E / hub threshold never matches) against the stored
// ones -> skip word; the new sums are stored: after this build the outputs hold the CSR of THIS edge list either way
__global__ __launch_bounds__(kFpBlocks) void fingerprint_decide_kernel(FingerprintWords *__restrict__ fp, int64_t E, int64_t N, int hub_threshold,
                                                                       int32_t *__restrict__ err)
{
    __shared__ unsigned long long red[2][kFpBlocks / kWave];
    unsigned long long a = fp->partial[threadIdx.x][0], b = fp->partial[threadIdx.x][1];
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[0][threadIdx.x / kWave] = a;
        red[1][threadIdx.x / kWave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kFpBlocks / kWave; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        a ^= hash_u64((uint64_t)E * 0x9E3779B97F4A7C15ULL + (uint64_t)N);
        b ^= hash_u64((uint64_t)N * 0xC2B2AE3D27D4EB4FULL + (uint64_t)(uint32_t)hub_threshold + ((uint64_t)E << 20));
        const int skip = (fp->valid == 1 && fp->stored[0] == a && fp->stored[1] == b) ? 1 : 0;
        fp->skip = skip;
        // the same edge list again: what its build reported is reported again (the reference raises on every call with bad ids);
        // a new edge list: its build records afresh
        if (skip && fp->bad && err) *err = 1;
        if (!skip) fp->bad = 0;
        fp->stored[0] = a;
        fp->stored[1] = b;
        fp->valid = 1;
    }
}

// dedicated helper workgroups appended to a finish launch: at most a quarter of the workgroups of that kernel the device can hold
// (more only poll beside each other), at most kDenseHelpers, SS_CSR_HELPERS=n for fewer (0: none -- every registered bucket is then
// worked off by the bucket workgroups alone: the helpers are an accelerator, nothing waits for them)
constexpr int kDenseHelpers = 128;  // (32 .. 256 helpers finish a rank^-0.9 collab-size graph in the same time: shares outnumber none of them)
template <typename Kernel>
int helper_budget(Kernel kernel)
{
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kRunThreads, 0) != hipSuccess) return 0;
    const int64_t slots = (int64_t)per_cu * prop.multiProcessorCount;
    const int64_t h = slots / 4 < kDenseHelpers ? slots / 4 : kDenseHelpers;
    return (int)(h > 0 ? h : 0);
}

inline int run_helpers(bool packed)
{
    static const int cap_env = getenv("SS_CSR_HELPERS") ? atoi(getenv("SS_CSR_HELPERS")) : -1;  // tuning / bisecting hook
    // per device: a process may drive GPUs of different sizes.  -1 = not computed yet; relaxed atomics: callers on several host
    // threads may race to fill an entry, with the same value
    static std::atomic<int> cache[2][64];
    static std::atomic<bool> ready{false};
    if (!ready.load(std::memory_order_acquire)) {
        for (auto &row : cache)
            for (auto &e : row) e.store(-1, std::memory_order_relaxed);
        ready.store(true, std::memory_order_release);
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int h = cache[packed][dev].load(std::memory_order_relaxed);
    if (h < 0) {
        h = packed ? helper_budget(finish_runs_kernel<true>) : helper_budget(finish_runs_kernel<false>);
        cache[packed][dev].store(h, std::memory_order_relaxed);
    }
    return cap_env >= 0 && cap_env < h ? cap_env : h;
}

}  // namespace ss

// (subgraph_sketch_debug.h) dedicated helper workgroups a build appends to its finish launch (whatever the stream: nothing waits for
// them, a CU mask only makes them fewer useful), and the number of cross-workgroup waits of any build of this process that gave up
extern "C" int ss_debug_csr_helpers(void *stream)
{
    (void)stream;
    return ss::run_helpers(true);
}
extern "C" int ss_debug_csr_protocol_faults(void)
{
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(ss::csr_protocol_faults), sizeof(v)) != hipSuccess) return -1;
    return v;
}


#ifdef SS_CSR_TIMING
extern "C" int ss_csr_timing_read(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ss::csr_phase_ticks), 16 * 8) != hipSuccess) return SS_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ss::csr_phase_ticks), z, 16 * 8) != hipSuccess) return SS_ERR_LAUNCH;
    }
    return SS_OK;
}
extern "C" int ss_csr_timing_write(const unsigned long long *in16)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ss::csr_phase_ticks), in16, 16 * 8) == hipSuccess ? SS_OK : SS_ERR_LAUNCH;
}
#endif

extern "C" size_t ss_csr_workspace_bytes(int64_t N, int64_t E)
{
    ss::LevelPlan lp;
    if (!ss::make_plan(N, E, N, lp)) return 0;
    return ss::carve(lp, E, nullptr).bytes;
}

// every check that can fail without a launch: arguments, plan, workspace (shared by the plain and the cached entry point -- the
// cached one must not let its fingerprint kernels declare the outputs valid for a call that then builds nothing)
static int csr_check(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, const int64_t *rowptr, const int32_t *col,
                     const int32_t *hub_rows, const int32_t *hub_count, const int32_t *mega_rows, const int32_t *mega_count,
                     const void *workspace, size_t workspace_bytes, ss::LevelPlan &lp)
{
    if (N < 0 || E < 0 || N >= ((int64_t)1 << 31) || !rowptr) return SS_ERR_INVALID_ARG;
    if (E > 0 && (!dst || !col)) return SS_ERR_INVALID_ARG;
    if ((hub_rows == nullptr) != (hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    if ((mega_rows == nullptr) != (mega_count == nullptr) || (mega_rows && !hub_rows)) return SS_ERR_INVALID_ARG;
    if (!ss::make_plan(N, E, src ? N : E, lp)) return SS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ss_csr_workspace_bytes(N, E)) return SS_ERR_WORKSPACE;
    return SS_OK;
}

// the process-wide fault count of wait_until: pinned + portable (every device of the process reaches it) + coherent host memory,
// allocated on first use and never freed; nullptr if the allocation fails (then only err_flag and the debug counter report)
static int32_t *protocol_fault_word()
{
    static int32_t *word = [] {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            return (int32_t *)nullptr;
        }
        memset(p, 0, 64);
        return (int32_t *)p;
    }();
    return word;
}
// How many images' worth of edges a bucket may hold and still be finished by its own workgroup (image by image).  More than one only
// when the launch has several workgroups per slot (512 resident: two per CU) -- a one-level plan (ogbl-collab size: 231 buckets)
// is ONE wave of workgroups, a bucket that sweeps twice is the launch's critical path while other CUs idle, and the share protocol,
// which spreads a bucket over those CUs, wins (measured: 58.8 -> 69.6 us at rank^-0.5); with thousands of buckets the sweeps of one
// overlap the others' (ogbl-ppa size, rank^-0.5: 602.5 / 495.0 / 491.6 us with 1 / 2 / 3 images, ogbl-citation2 size 959.8 / 735.2 / 726.6;
// rank^-0.9: 716 / 684 / 643 and 1 106 / 982 / 955; profiles/round6_csr_walk_images.txt).  SS_CSR_WALK_IMAGES = 1 .. 3 forces a
// value (1: every bucket above the image goes through the share protocol, as in round 5).
static uint32_t finish_walk_max(int64_t fine_buckets)
{
    static const int env = getenv("SS_CSR_WALK_IMAGES") ? atoi(getenv("SS_CSR_WALK_IMAGES")) : 0;
    const int images = (env >= 1 && env <= 3) ? env : (fine_buckets > 2048 ? 3 : 1);
    return (uint32_t)(images * ss::kFinishCap);
}
// SS_CSR_WAIT_TICKS (100 MHz ticks; test hook: 0 makes every wait that is not already satisfied give up): the bound of wait_until
static unsigned long long protocol_wait_ticks()
{
    static const unsigned long long ticks = getenv("SS_CSR_WAIT_TICKS") ? strtoull(getenv("SS_CSR_WAIT_TICKS"), nullptr, 10) : ss::kWaitTicks;
    return ticks;
}

// (subgraph_sketch.h) what the product reads: one int32 in pinned, portable host memory that every finish launch of this process
// stamps when a wait gives up -- read here with a plain load, no synchronisation, whatever the device
extern "C" int ss_csr_protocol_faults(void)
{
    const int32_t *w = protocol_fault_word();
    return w ? __atomic_load_n(w, __ATOMIC_RELAXED) : -1;
}

// the launches of one build (arguments checked by csr_check)
static int csr_build_launch(const ss::LevelPlan &lp, const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                            int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count, int32_t *mega_rows,
                            int32_t *mega_count, int32_t *err_flag, void *workspace, hipStream_t stream, const int32_t *skip = nullptr,
                            int32_t *bad_record = nullptr, int key_stride = 2)
{
    using namespace ss;
    if (N == 0 || E == 0) {
        if (n_self_loops_out && hipMemsetAsync(n_self_loops_out, 0, 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        if (hub_count && hipMemsetAsync(hub_count, 0, 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
        if (mega_count && hipMemsetAsync(mega_count, 0, 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        if (hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        return SS_OK;
    }
    ProfileSpan span(stream, SS_PROF_CSR);  // all launches of this build
    const RowOutputs rows_out = {rowptr, (int)hub_threshold, hub_rows, hub_count, mega_rows, mega_count, skip, err_flag, protocol_fault_word(), protocol_wait_ticks(), finish_walk_max(lp.groups[lp.levels])};
    const Workspace w = carve(lp, E, workspace);
    unsigned long long *n_self = n_self_loops_out ? reinterpret_cast<unsigned long long *>(n_self_loops_out) : w.scratch;
    const int tiles0 = (int)lp.tmax[0];
    const bool packed = lp.packed;  // records of the last level (read by the finish step) are 4 bytes
    if (lp.packed0)  // (two levels: 4-byte level-0 records, see make_plan)
        hipLaunchKernelGGL(tile_sort_kernel<true>, dim3(tiles0), dim3(kSortThreads), 0, stream, src, dst, E, N, key_stride, lp.shift[0], lp.src_bits0, lp.keys[0],
                           tiles0, (void *)w.staged_a, w.lv[0].off, w.tile_max, err_flag, hub_count, mega_count, w.dense_count, w.dense_sync, (int)max_dense_buckets(E), skip, bad_record);
    else if (packed && lp.levels == 1)
        hipLaunchKernelGGL(tile_sort_kernel<true>, dim3(tiles0), dim3(kSortThreads), 0, stream, src, dst, E, N, key_stride, lp.shift[0], lp.src_bits, lp.keys[0],
                           tiles0, (void *)w.staged_a, w.lv[0].off, w.tile_max, err_flag, hub_count, mega_count, w.dense_count, w.dense_sync, (int)max_dense_buckets(E), skip, bad_record);
    else
        hipLaunchKernelGGL(tile_sort_kernel<false>, dim3(tiles0), dim3(kSortThreads), 0, stream, src, dst, E, N, key_stride, lp.shift[0], lp.src_bits, lp.keys[0],
                           tiles0, (void *)w.staged_a, w.lv[0].off, w.tile_max, err_flag, hub_count, mega_count, w.dense_count, w.dense_sync, (int)max_dense_buckets(E), skip, bad_record);
    SS_LAUNCH_CHECK();
    ParentLevel par = {w.lv[0].off, nullptr, nullptr, nullptr, tiles0, tiles0, lp.keys[0], -1};
    const void *in = w.staged_a;
    for (int l = 1; l < lp.levels; ++l) {
        const LevelArrays &a = w.lv[l];
        hipLaunchKernelGGL(level_scan_kernel, dim3((unsigned)lp.groups[l]), dim3(1024), 0, stream, par, w.lv[l - 1].prefix, w.lv[l - 1].cfirst,
                           a.gcount, a.base, skip);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(level_tiles_kernel, dim3(1), dim3(1024), 0, stream, a.gcount, lp.groups[l], a.tb, a.n_tiles, skip);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(level_fill_kernel, dim3((unsigned)lp.groups[l]), dim3(kWave), 0, stream, par, w.lv[l - 1].cfirst, a.tb, a.gcount, a.base,
                           a.header, skip);
        SS_LAUNCH_CHECK();
        void *out_buf = (l & 1) ? w.staged_b : (void *)w.staged_a;
        const bool last = l == lp.levels - 1;
        const LevelOut out = {out_buf, a.off, a.tstart, a.header, a.n_tiles, (int)lp.tmax[l], lp.keys[l], lp.shift[l], lp.src_bits,
                              (1 << lp.node_shift) - 1};
        if (l == 1 && lp.packed0)
            hipLaunchKernelGGL((regroup_sort_kernel<true, true>), dim3((unsigned)lp.tmax[l]), dim3(kRegroupThreads), 0, stream, par,
                               (const int2 *)in, w.lv[l - 1].prefix, out, skip, lp.src_bits0, lp.shift[0]);
        else if (last && packed)
            hipLaunchKernelGGL(regroup_sort_kernel<true>, dim3((unsigned)lp.tmax[l]), dim3(kRegroupThreads), 0, stream, par,
                               (const int2 *)in, w.lv[l - 1].prefix, out, skip);
        else
            hipLaunchKernelGGL(regroup_sort_kernel<false>, dim3((unsigned)lp.tmax[l]), dim3(kRegroupThreads), 0, stream, par,
                               (const int2 *)in, w.lv[l - 1].prefix, out, skip);
        SS_LAUNCH_CHECK();
        par = ParentLevel{a.off, a.tstart, a.tb, a.base, (int)lp.tmax[l], tiles0, lp.keys[l], lp.log2_keys[l]};
        in = out_buf;
    }
    const int64_t fine = lp.groups[lp.levels];
    // buckets that do not fit the LDS image are registered and worked off inside the finish launch, by every workgroup that has time
    // and by `helpers` extra ones (see "who may wait for whom"): no launches of their own, nothing to skip on an unskewed graph
    const int64_t share_cap = max_dense_shares(E);
    int helpers = run_helpers(packed);
    if ((int64_t)helpers > share_cap) helpers = (int)share_cap;
    const DenseRunArgs dense = {w.dense_count, w.dense_list, w.dense_sync, w.dense_node_cnt, w.dense_share_off, w.dense_share_lo, w.dense_claim, nullptr, nullptr, 0, 0};
    if (packed)
        hipLaunchKernelGGL(finish_runs_kernel<true>, dim3((unsigned)(fine + helpers)), dim3(kRunThreads), 0, stream, par, in, w.tile_max, tiles0,
                           lp.node_shift, lp.src_bits, N, col, n_self, rows_out, dense, fine);
    else
        hipLaunchKernelGGL(finish_runs_kernel<false>, dim3((unsigned)(fine + helpers)), dim3(kRunThreads), 0, stream, par, in, w.tile_max, tiles0,
                           lp.node_shift, lp.src_bits, N, col, n_self, rows_out, dense, fine);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_csr_build(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                            int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count,
                            int32_t *mega_rows, int32_t *mega_count,
                            int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (E > 0 && !src) return SS_ERR_INVALID_ARG;
    ss::LevelPlan lp;
    const int rc = csr_check(src, dst, E, N, rowptr, col, hub_rows, hub_count, mega_rows, mega_count, workspace, workspace_bytes, lp);
    if (rc != SS_OK) return rc;
    return csr_build_launch(lp, src, dst, E, N, rowptr, col, n_self_loops_out, hub_threshold, hub_rows, hub_count, mega_rows, mega_count, err_flag,
                            workspace, (hipStream_t)stream_);
}

// ss_csr_build that first compares a content fingerprint of (src, dst) with the one the previous call left in `fingerprint`
// (device buffer of SS_CSR_FINGERPRINT_BYTES, zeroed by the caller before its first use, tied to THESE output buffers): equal ->
// the outputs already hold this CSR and every kernel of the build exits at once; different (or first use) -> an ordinary build,
// after which `fingerprint` describes the new contents.  No host synchronisation either way.  A skipped call re-reports (err_flag)
// the out-of-range ids the build of the cached CSR met.  (reference models/elph.py:186 + runners/train.py:188-198: the same edges in a fresh tensor every step)
// Every check that can fail without a launch runs BEFORE the fingerprint kernels (they declare the outputs valid for this edge
// list); if a launch of the build itself fails after them, the fingerprint is invalidated on the stream before the error is
// returned -- a later call with the same edges must not skip over a CSR that was never completed.
extern "C" int ss_csr_build_cached(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                                   int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count,
                                   int32_t *mega_rows, int32_t *mega_count, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                                   void *fingerprint, void *stream_)
{
    using namespace ss;
    if (!fingerprint || E <= 0 || N <= 0 || !src || !dst) return SS_ERR_INVALID_ARG;
    LevelPlan lp;
    int rc = csr_check(src, dst, E, N, rowptr, col, hub_rows, hub_count, mega_rows, mega_count, workspace, workspace_bytes, lp);
    if (rc != SS_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    FingerprintWords *fp = reinterpret_cast<FingerprintWords *>(fingerprint);
    hipLaunchKernelGGL(fingerprint_kernel, dim3(kFpBlocks), dim3(256), 0, stream, src, dst, E, fp);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(fingerprint_decide_kernel, dim3(1), dim3(kFpBlocks), 0, stream, fp, E, N, (int)hub_threshold, err_flag);
    rc = hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_LAUNCH;
    if (rc == SS_OK)
        rc = csr_build_launch(lp, src, dst, E, N, rowptr, col, n_self_loops_out, hub_threshold, hub_rows, hub_count, mega_rows, mega_count, err_flag,
                              workspace, stream, &fp->skip, &fp->bad);
    if (rc != SS_OK) (void)hipMemsetAsync(&fp->valid, 0, sizeof(fp->valid), stream);  // the outputs may be half-written
    return rc;
}

// The pairs of a query grouped by their first node (reference hashing.py:270-274 reads cards[u] / the rows of u once per PAIR;
// BUDDY's link sets repeat every source many times -- ogbl-citation2's evaluation set lists 1 000 negatives per source): the
// CSR builder run on (destination = first node of pair q, source = q).  order[0 .. B) is a permutation of the pair indices in
// which all pairs of one first node are consecutive (rowptr [N + 1] says where each node's group starts).  Nothing is dropped:
// ids out of range are keyed to node 0.  ss_pair_features_grouped walks `order` and reloads a first node's rows only when it
// changes.  Workspace: ss_csr_workspace_bytes(N, B).
extern "C" int ss_group_links_by_source(const int64_t *links, int64_t B, int64_t N, int32_t *order, int64_t *rowptr, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (B < 0 || B >= ((int64_t)1 << 31) || N <= 0 || (B > 0 && (!links || !order))) return SS_ERR_INVALID_ARG;
    ss::LevelPlan lp;
    const int rc = csr_check(nullptr, links, B, N, rowptr, order, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, lp);
    if (rc != SS_OK) return rc;
    return csr_build_launch(lp, nullptr, links, B, N, rowptr, order, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                            (hipStream_t)stream);
}

// The entries of an id array grouped by id: order[0 .. E) is a permutation of the entry indices in which all entries with the
// same id are consecutive (rowptr [N + 1] says where each id's group starts), in unspecified order inside a group -- follow with
// ss_csr_sort_rows for ascending entry indices, i.e. a STABLE grouping (reference datasets/elph.py:87-110: gcn_norm's degree sums
// and torch_sparse.spmm's scatter-add both accumulate in edge order; sign.py groups the edge list by column and by row).  Ids out
// of [0, N) (after torch-style negative wrapping) are reported through err_flag and keyed to node 0.  Workspace: ss_csr_workspace_bytes(N, E).
extern "C" int ss_csr_group_ids(const int64_t *ids, int64_t E, int64_t N, int32_t *order, int64_t *rowptr, int32_t *err_flag, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    if (E < 0 || E >= ((int64_t)1 << 31) || N <= 0 || (E > 0 && (!ids || !order))) return SS_ERR_INVALID_ARG;
    ss::LevelPlan lp;
    const int rc = csr_check(nullptr, ids, E, N, rowptr, order, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, lp);
    if (rc != SS_OK) return rc;
    return csr_build_launch(lp, nullptr, ids, E, N, rowptr, order, nullptr, 0, nullptr, nullptr, nullptr, nullptr, err_flag, workspace,
                            (hipStream_t)stream, nullptr, nullptr, 1);
}

namespace ss {

// ---- rows of a CSR sorted ascending ----------------------------------------------------------------------------------------------
// compare-exchange network over the low `G` lanes of a lane group (G = 16 or 64, a power of two): ascending
template <int G>
__device__ __forceinline__ int bitonic_lanes(int v, int lane)
{
#pragma unroll
    for (int k = 2; k <= G; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j);
            const bool up = (lane & k) == 0, low = (lane & j) == 0;
            v = (low == up) ? (v < o ? v : o) : (v > o ? v : o);
        }
    return v;
}

// rows of at most 64 entries: four rows per wavefront (16 lanes each) where all four have at most 16 entries, else one after the
// other over the whole wavefront; longer rows are listed for sort_rows_long_kernel
__global__ __launch_bounds__(256) void sort_rows_short_kernel(const int64_t *__restrict__ rowptr, int32_t *__restrict__ col, int64_t N,
                                                              int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long,
                                                              const int32_t *__restrict__ keep_word)
{
    if (keep_word && *keep_word == 0) return;  // (see ss_csr_sort_rows)
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t r0 = wave * 4;
    if (r0 >= N) return;
    // lanes 0..4 read the five row bounds of this wavefront's four rows
    const int64_t my = lane <= 4 ? rowptr[r0 + lane < N ? r0 + lane : N] : 0;
    int64_t b[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) b[k] = __shfl(my, k);
    const int64_t max_len = [&] { int64_t m = 0; for (int k = 0; k < 4; ++k) m = b[k + 1] - b[k] > m ? b[k + 1] - b[k] : m; return m; }();
    if (max_len <= 16) {  // (wave-uniform)
        const int g = lane >> 4, l = lane & 15;
        const int64_t a = b[0] + 0 * g;  // (keep b[] in SGPRs: select by comparisons)
        const int64_t lo = g == 0 ? b[0] : g == 1 ? b[1] : g == 2 ? b[2] : b[3];
        const int64_t hi = g == 0 ? b[1] : g == 1 ? b[2] : g == 2 ? b[3] : b[4];
        (void)a;
        const int len = (int)(hi - lo);
        int v = l < len ? col[lo + l] : 0x7FFFFFFF;
        if (len > 1) v = bitonic_lanes<16>(v, l);  // (the whole 16-lane group takes the same branch)
        if (l < len) col[lo + l] = v;
        return;
    }
    for (int k = 0; k < 4; ++k) {
        const int64_t lo = b[k], len = b[k + 1] - b[k];
        if (len > kWave) {
            if (lane == 0) long_rows[atomicAdd(n_long, 1)] = (int32_t)(r0 + k);
            continue;
        }
        if (len < 2) continue;
        int v = lane < len ? col[lo + lane] : 0x7FFFFFFF;
        v = bitonic_lanes<kWave>(v, lane);
        if (lane < len) col[lo + lane] = v;
    }
}

// rows above 64 entries, one workgroup each (grid-stride over the list): bitonic sort in LDS up to 16 384 entries; longer rows (hub
// rows of power-law graphs) are sorted in LDS chunk by chunk and the chunks merged pairwise (merge path: every thread finds where
// its stretch of the output begins in both runs by a binary search along a diagonal, then merges sequentially), ping-ponging
// between col and scratch (the row's own stretch of an E-entry scratch array)
constexpr int kSortRowLds = 16384;

// ascending bitonic sort of buf[0 .. n2), n2 a power of two <= kSortRowLds (all 1024 threads)
__device__ __forceinline__ void bitonic_lds(int32_t *buf, int n2)
{
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const int32_t a = buf[i], c = buf[p];
                    if ((a > c) == ((i & k) == 0)) { buf[i] = c; buf[p] = a; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(1024) void sort_rows_long_kernel(const int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                              int32_t *__restrict__ scratch, const int32_t *__restrict__ long_rows,
                                                              const int32_t *__restrict__ n_long)
{
    __shared__ int32_t buf[kSortRowLds];
    const int n_rows = *n_long;
    for (int q = blockIdx.x; q < n_rows; q += gridDim.x) {
        const int64_t lo = rowptr[long_rows[q]], len = rowptr[long_rows[q] + 1] - lo;
        int32_t *base = col + lo;
        for (int64_t c0 = 0; c0 < len; c0 += kSortRowLds) {  // sorted chunks of up to 16 384 entries
            const int n = (int)(len - c0 < kSortRowLds ? len - c0 : kSortRowLds);
            int n2 = 1;
            while (n2 < n) n2 <<= 1;
            for (int i = threadIdx.x; i < n2; i += 1024) buf[i] = i < n ? base[c0 + i] : 0x7FFFFFFF;
            __syncthreads();
            bitonic_lds(buf, n2);
            for (int i = threadIdx.x; i < n; i += 1024) base[c0 + i] = buf[i];
            __syncthreads();
        }
        int32_t *src = base, *dst = scratch + lo;
        for (int64_t width = kSortRowLds; width < len; width <<= 1) {  // pairwise merges of runs of `width`
            for (int64_t s0 = 0; s0 < len; s0 += 2 * width) {
                const int64_t na = len - s0 < width ? len - s0 : width, nb = len - s0 - na < width ? len - s0 - na : width;
                const int32_t *A = src + s0, *Bp = A + na;
                int32_t *O = dst + s0;
                const int64_t m = na + nb, per = (m + 1023) / 1024;
                const int64_t d0 = per * threadIdx.x < m ? per * threadIdx.x : m, d1 = d0 + per < m ? d0 + per : m;
                // merge path: the number i of A's entries among the first d0 outputs (ties: A first -- the sort need not be
                // stable, the entries of a row are distinct positions / may repeat harmlessly)
                int64_t a_lo = d0 > nb ? d0 - nb : 0, a_hi = d0 < na ? d0 : na;
                while (a_lo < a_hi) {
                    const int64_t i = (a_lo + a_hi) >> 1;  // try i entries of A, d0 - i of B
                    if (A[i] <= Bp[d0 - i - 1]) a_lo = i + 1; else a_hi = i;
                }
                int64_t i = a_lo, j = d0 - a_lo;
                for (int64_t o = d0; o < d1; ++o) {
                    const bool take_a = j >= nb || (i < na && A[i] <= Bp[j]);
                    O[o] = take_a ? A[i++] : Bp[j++];
                }
            }
            __syncthreads();
            int32_t *t = src; src = dst; dst = t;
        }
        if (src != base) {
            for (int64_t i = threadIdx.x; i < len; i += 1024) base[i] = src[i];
        }
        __syncthreads();
    }
}

}  // namespace ss

// Every row of a CSR sorted ascending in place (col of rows [0, N)).  With the entry indices as payload (ss_csr_group_ids,
// ss_group_links_by_source) that makes the grouping STABLE -- the reference's edge order inside every group; with source ids it
// gives the sorted adjacency scipy / torch_sparse hold.  Workspace: ss_csr_sort_workspace_bytes(E) device bytes.
extern "C" size_t ss_csr_sort_workspace_bytes(int64_t E) { return E < 0 ? 0 : (size_t)(E / ss::kWave + 2) * 4 + 256 + (size_t)(E > 0 ? E : 1) * 4; }

// only_if (nullable): device word; the rows are sorted only if it is NON-zero when the launches run (sign.py: the grouping by column
// needs its order only when some edge weight differs from 1 -- ss_gcn_scan_edges leaves that in the first word of its buffer)
extern "C" int ss_csr_sort_rows(const int64_t *rowptr, int32_t *col, int64_t N, int64_t E, const int32_t *only_if, void *workspace,
                                size_t workspace_bytes, void *stream_)
{
    using namespace ss;
    if (N < 0 || E < 0 || N >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31)) return SS_ERR_INVALID_ARG;
    if (N == 0 || E == 0) return SS_OK;
    if (!rowptr || !col) return SS_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < ss_csr_sort_workspace_bytes(E)) return SS_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    int32_t *n_long = reinterpret_cast<int32_t *>(workspace);
    int32_t *long_rows = n_long + 64;  // (its own cache lines)
    int32_t *scratch = long_rows + (E / kWave + 2);
    if (hipMemsetAsync(n_long, 0, 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
    const int64_t waves = (N + 3) / 4;
    hipLaunchKernelGGL(sort_rows_short_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, rowptr, col, N, long_rows, n_long, only_if);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sort_rows_long_kernel, dim3(512), dim3(1024), 0, stream, rowptr, col, scratch, long_rows, n_long);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
