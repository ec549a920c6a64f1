// ss_csr.hip -- CSR-by-destination construction on the device.
//
// The reference materialises one message x[src] per edge and scatter-maxes it (PyG propagate,
// hashing.py:34,44).  The MI355X engine instead pulls: edges are grouped by destination once and every
// hop streams whole neighbour rows.  The order of sources inside a row is unspecified (min / max do not
// care), which lets the build be a two-level counting sort with NO per-edge global atomics -- on gfx950
// device-scope atomics from 8 non-coherent XCD L2s are served at the memory side and cost ~45 ns per
// 1000 edges each way (first version: 0.23 ms for 2.4 M edges, profiles/round1_v1_kernel_stats.csv).
//
//   A1 bucket_count   : each block takes a contiguous slice of the edge list, histograms dst >> shift in
//                       LDS (a bucket = NB = 1024 consecutive nodes; 256..8192 were measured, 1024 is best on
//                       every shape), writes its column of the bucket-major [buckets x blocks] count matrix.
//   A2 bucket_offsets : per bucket, exclusive scan over blocks (column of the matrix) + bucket totals.
//   A3 bucket_bases   : single block, exclusive scan of bucket totals -> bucket base offsets.
//   A4 bucket_scatter : same slices as A1; LDS cursors seeded with base[bucket] + offset[bucket][block];
//                       edges are written as (src, dst) int32 pairs grouped by bucket; also validates src and
//                       reduces max(id)+1 (= self-loop count of add_self_loops, hashing.py:148).
//   B  bucket_finish  : one block per bucket: LDS histogram over its NB nodes, LDS scan -> rowptr,
//                       LDS cursors -> col.
#include "ss_common.hpp"

namespace ss {

constexpr int kCsrThreads = 256;
constexpr int kEdgesPerBlockMin = 4096;   // slice size lower bound (A1 / A4)
constexpr int kMaxSliceBlocks = 4096;
constexpr int kMaxBuckets = 4096;         // LDS histogram size of A1 / A4
constexpr int kMinNodesPerBucket = 1024;
constexpr int kMaxNodesPerBucket = 16384; // LDS of B: 2 * NB * 4 bytes
constexpr int kFinishThreads = 1024;

struct CsrPlan {
    int shift;         // bucket = dst >> shift
    int nodes_per_bucket;
    int buckets;
    int slice_blocks;
    int64_t slice_edges;
};

inline bool make_plan(int64_t N, int64_t E, CsrPlan &p)
{
    int nb = kMinNodesPerBucket, shift = 10;
    while ((N + nb - 1) / nb > kMaxBuckets && nb < kMaxNodesPerBucket) { nb <<= 1; ++shift; }
    if ((N + nb - 1) / nb > kMaxBuckets) return false;
    p.shift = shift;
    p.nodes_per_bucket = nb;
    p.buckets = (int)((N + nb - 1) / nb);
    if (p.buckets < 1) p.buckets = 1;
    int64_t blocks = (E + kEdgesPerBlockMin - 1) / kEdgesPerBlockMin;
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxSliceBlocks) blocks = kMaxSliceBlocks;
    p.slice_blocks = (int)blocks;
    p.slice_edges = (E + blocks - 1) / blocks;
    return true;
}

// A1: histogram of dst >> shift over this block's edge slice.  Reads dst only (8 B/edge); edges whose dst is out of
// range are dropped here and in A4 alike.  counts is bucket-major: counts[bucket * slice_blocks + block].
__global__ __launch_bounds__(kCsrThreads) void bucket_count_kernel(const int64_t *__restrict__ dst, int64_t E, int64_t N, int shift,
                                                                   int buckets, int64_t slice_edges, int slice_blocks,
                                                                   uint32_t *__restrict__ counts, int32_t *__restrict__ err,
                                                                   unsigned long long *__restrict__ n_self, int32_t *__restrict__ hub_count)
{
    __shared__ uint32_t hist[kMaxBuckets];
    // outputs of the LATER kernels of this build (A4: n_self, B: hub_count) are cleared here: saves two memset launches
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *n_self = 0ULL;
        if (hub_count) *hub_count = 0;
    }
    for (int b = threadIdx.x; b < buckets; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice_edges;
    const int64_t hi = lo + slice_edges < E ? lo + slice_edges : E;
    bool bad = false;
    int64_t e = lo + threadIdx.x;
    for (; e + 3 * (int64_t)kCsrThreads < hi; e += 4 * (int64_t)kCsrThreads) {
        int64_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = dst[e + k * (int64_t)kCsrThreads];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((uint64_t)d[k] < (uint64_t)N) atomicAdd(&hist[d[k] >> shift], 1u);
            else bad = true;
        }
    }
    for (; e < hi; e += kCsrThreads) {
        const int64_t d = dst[e];
        if ((uint64_t)d < (uint64_t)N) atomicAdd(&hist[d >> shift], 1u);
        else bad = true;
    }
    if (bad && err) *err = 1;
    __syncthreads();
    for (int b = threadIdx.x; b < buckets; b += blockDim.x) counts[(int64_t)b * slice_blocks + blockIdx.x] = hist[b];
}

// one wave per bucket: exclusive scan of the bucket's column over slice blocks (in place), bucket total out
__global__ __launch_bounds__(kCsrThreads) void bucket_offsets_kernel(uint32_t *__restrict__ counts, int slice_blocks, int buckets,
                                                                     unsigned long long *__restrict__ bucket_total)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= buckets) return;
    uint32_t carry = 0;
    for (int g0 = 0; g0 < slice_blocks; g0 += kWave) {
        const int g = g0 + lane;
        const uint32_t x = g < slice_blocks ? counts[(int64_t)b * slice_blocks + g] : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (g < slice_blocks) counts[(int64_t)b * slice_blocks + g] = carry + inc - x;
        carry += __shfl(inc, kWave - 1);
    }
    if (lane == 0) bucket_total[b] = carry;
}

// single block: exclusive scan of bucket totals in place (-> bucket bases); grand total to rowptr[N]
__global__ __launch_bounds__(kCsrThreads) void bucket_bases_kernel(unsigned long long *__restrict__ bucket_total, int buckets,
                                                                   int64_t *__restrict__ rowptr, int64_t N)
{
    __shared__ unsigned long long wave_tot[kCsrThreads / kWave];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    for (int start = 0; start < buckets; start += kCsrThreads) {
        const int i = start + threadIdx.x;
        const unsigned long long x = i < buckets ? bucket_total[i] : 0ULL;
        unsigned long long inc = x;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned long long o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wv] = inc;
        __syncthreads();
        unsigned long long pre = carry_s;
        for (int w = 0; w < wv; ++w) pre += wave_tot[w];
        if (i < buckets) bucket_total[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == kCsrThreads - 1) carry_s = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowptr[N] = (int64_t)carry_s;
}

// A4: scatter (src, dst) pairs into their bucket's segment; also reduces max(id) + 1 (the self-loop count of
// add_self_loops, hashing.py:148).  An out-of-range src sets the error flag and is clamped to 0 (memory safe; the
// host raises IndexError in strict mode), an out-of-range dst drops the edge exactly as A1 did.
__global__ __launch_bounds__(kCsrThreads) void bucket_scatter_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                                     int64_t E, int64_t N, int shift, int buckets, int64_t slice_edges,
                                                                     int slice_blocks, const uint32_t *__restrict__ offsets,
                                                                     const unsigned long long *__restrict__ bucket_base,
                                                                     int2 *__restrict__ staged /*[E] (src, dst)*/,
                                                                     unsigned long long *__restrict__ n_self, int32_t *__restrict__ err)
{
    __shared__ unsigned long long cursor[kMaxBuckets];
    __shared__ unsigned long long block_max;
    for (int b = threadIdx.x; b < buckets; b += blockDim.x)
        cursor[b] = bucket_base[b] + offsets[(int64_t)b * slice_blocks + blockIdx.x];
    if (threadIdx.x == 0) block_max = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice_edges;
    const int64_t hi = lo + slice_edges < E ? lo + slice_edges : E;
    int64_t my_max = -1;
    bool bad = false;
    auto place = [&](int64_t s, int64_t d) {
        const int64_t mx = s > d ? s : d;
        my_max = mx > my_max ? mx : my_max;
        if ((uint64_t)d >= (uint64_t)N) return;
        if ((uint64_t)s >= (uint64_t)N) { bad = true; s = 0; }
        const unsigned long long pos = atomicAdd(&cursor[d >> shift], 1ULL);
        staged[pos] = make_int2((int)s, (int)d);
    };
    int64_t e = lo + threadIdx.x;
    for (; e + 3 * (int64_t)kCsrThreads < hi; e += 4 * (int64_t)kCsrThreads) {
        int64_t sv[4], dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sv[k] = src[e + k * (int64_t)kCsrThreads];
            dv[k] = dst[e + k * (int64_t)kCsrThreads];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) place(sv[k], dv[k]);
    }
    for (; e < hi; e += kCsrThreads) place(src[e], dst[e]);
    unsigned long long m = my_max < 0 ? 0ULL : (unsigned long long)my_max + 1ULL;
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicMax(&block_max, m);
    if (bad && err) *err = 1;
    __syncthreads();
    if (threadIdx.x == 0 && block_max) atomicMax(n_self, block_max);
}

// one block per bucket
__global__ __launch_bounds__(kFinishThreads) void bucket_finish_kernel(const int2 *__restrict__ staged,
                                                                       const unsigned long long *__restrict__ bucket_base,
                                                                       int buckets, int nodes_per_bucket, int64_t N,
                                                                       int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                                       int hub_threshold, int32_t *__restrict__ hub_rows,
                                                                       int32_t *__restrict__ hub_count)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *cnt = smem;                       // [NB] counts, then cursors
    uint32_t *excl = smem + nodes_per_bucket;   // [NB] exclusive offsets
    __shared__ uint32_t wave_tot[kFinishThreads / kWave];
    const int b = blockIdx.x;
    const int64_t node0 = (int64_t)b * nodes_per_bucket;
    const unsigned long long seg_lo = bucket_base[b];
    const unsigned long long seg_hi = (b + 1 < buckets) ? bucket_base[b + 1] : (unsigned long long)rowptr[N];
    for (int i = threadIdx.x; i < nodes_per_bucket; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    {
        unsigned long long e = seg_lo + threadIdx.x;
        for (; e + 3ULL * blockDim.x < seg_hi; e += 4ULL * blockDim.x) {
            int y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = staged[e + (unsigned long long)k * blockDim.x].y;
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(&cnt[y[k] - (int)node0], 1u);
        }
        for (; e < seg_hi; e += blockDim.x) atomicAdd(&cnt[staged[e].y - (int)node0], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..NB): each thread owns a contiguous run
    const int per = nodes_per_bucket / (int)blockDim.x;  // NB is a power of two >= blockDim.x
    const int base = threadIdx.x * per;
    uint32_t run = 0;
    for (int k = 0; k < per; ++k) run += cnt[base + k];
    uint32_t inc = run;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int w = 0; w < wv; ++w) pre += wave_tot[w];
    uint32_t ex = pre + inc - run;
    for (int k = 0; k < per; ++k) {
        const uint32_t c = cnt[base + k];
        excl[base + k] = ex;
        if (node0 + base + k < N) rowptr[node0 + base + k] = (int64_t)(seg_lo + ex);
        if (hub_rows && c > (uint32_t)hub_threshold) hub_rows[atomicAdd(hub_count, 1)] = (int32_t)(node0 + base + k);
        ex += c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nodes_per_bucket; i += blockDim.x) cnt[i] = excl[i];
    __syncthreads();
    {
        unsigned long long e = seg_lo + threadIdx.x;
        for (; e + 3ULL * blockDim.x < seg_hi; e += 4ULL * blockDim.x) {
            int2 sd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[k] = staged[e + (unsigned long long)k * blockDim.x];
#pragma unroll
            for (int k = 0; k < 4; ++k) col[seg_lo + atomicAdd(&cnt[sd[k].y - (int)node0], 1u)] = sd[k].x;
        }
        for (; e < seg_hi; e += blockDim.x) {
            const int2 sd = staged[e];
            col[seg_lo + atomicAdd(&cnt[sd.y - (int)node0], 1u)] = sd.x;
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ss

// workspace layout: [count matrix: slice_blocks*buckets u32][bucket totals/bases: buckets+1 u64][staged edges: E int2]
extern "C" size_t ss_csr_workspace_bytes(int64_t N, int64_t E)
{
    ss::CsrPlan p;
    if (N < 0 || E < 0 || !ss::make_plan(N, E, p)) return 0;
    return ss::align256((size_t)p.slice_blocks * p.buckets * 4) + ss::align256((size_t)(p.buckets + 1) * 8) +
           ss::align256((size_t)(E > 0 ? E : 1) * 8);
}

extern "C" int ss_csr_build(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                            int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count,
                            int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace ss;
    if (N < 0 || E < 0 || N >= ((int64_t)1 << 31) || !rowptr) return SS_ERR_INVALID_ARG;
    if (E > 0 && (!src || !dst || !col)) return SS_ERR_INVALID_ARG;
    CsrPlan p;
    if (!make_plan(N, E, p)) return SS_ERR_UNSUPPORTED;  // N > 64 M nodes
    if (!workspace || workspace_bytes < ss_csr_workspace_bytes(N, E)) return SS_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    char *ws = reinterpret_cast<char *>(workspace);
    auto *counts = reinterpret_cast<uint32_t *>(ws);
    ws += align256((size_t)p.slice_blocks * p.buckets * 4);
    auto *bucket_total = reinterpret_cast<unsigned long long *>(ws);
    ws += align256((size_t)(p.buckets + 1) * 8);
    auto *staged = reinterpret_cast<int2 *>(ws);

    if ((hub_rows == nullptr) != (hub_count == nullptr)) return SS_ERR_INVALID_ARG;
    if (N == 0 || E == 0) {
        if (n_self_loops_out && hipMemsetAsync(n_self_loops_out, 0, 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        if (hub_count && hipMemsetAsync(hub_count, 0, 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
        if (hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        return SS_OK;
    }
    unsigned long long *n_self = reinterpret_cast<unsigned long long *>(n_self_loops_out);
    if (!n_self) n_self = bucket_total + p.buckets;  // spare workspace slot when the caller does not want the value
    hipLaunchKernelGGL(bucket_count_kernel, dim3(p.slice_blocks), dim3(kCsrThreads), 0, stream, dst, E, N, p.shift, p.buckets,
                       p.slice_edges, p.slice_blocks, counts, err_flag, n_self, hub_count);
    SS_LAUNCH_CHECK();
    const int waves_per_block = kCsrThreads / kWave;
    hipLaunchKernelGGL(bucket_offsets_kernel, dim3((p.buckets + waves_per_block - 1) / waves_per_block), dim3(kCsrThreads), 0, stream,
                       counts, p.slice_blocks, p.buckets, bucket_total);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_bases_kernel, dim3(1), dim3(kCsrThreads), 0, stream, bucket_total, p.buckets, rowptr, N);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(p.slice_blocks), dim3(kCsrThreads), 0, stream, src, dst, E, N, p.shift, p.buckets,
                       p.slice_edges, p.slice_blocks, counts, bucket_total, staged, n_self, err_flag);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_finish_kernel, dim3(p.buckets), dim3(p.nodes_per_bucket < kFinishThreads ? p.nodes_per_bucket : kFinishThreads), (size_t)p.nodes_per_bucket * 8, stream, staged,
                       bucket_total, p.buckets, p.nodes_per_bucket, N, rowptr, col, (int)hub_threshold, hub_rows, hub_count);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
