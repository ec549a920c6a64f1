// ss_csr.hip -- CSR-by-destination construction on the device.
//
// The reference materialises one message x[src] per edge and scatter-maxes it (PyG propagate,
// hashing.py:34,44).  The MI355X engine instead pulls: rows are grouped by destination once and every
// hop streams whole neighbour rows.  Construction = degree histogram (atomics) -> exclusive scan ->
// cursor fill (atomics).  The order of sources inside a row is unspecified; min/max do not care.
#include "ss_common.hpp"

namespace ss {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;                        // items per thread
constexpr int kScanTile = kScanBlock * kScanItems;   // 2048 counters per block

__global__ __launch_bounds__(256) void degree_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                     int64_t E, int64_t N, unsigned long long *__restrict__ deg,
                                                     int32_t *__restrict__ err)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = src[e], d = dst[e];
        if ((uint64_t)s >= (uint64_t)N || (uint64_t)d >= (uint64_t)N) {
            if (err) *err = 1;
            continue;
        }
        atomicAdd(&deg[d], 1ULL);
    }
}

// block-local exclusive scan of a 2048-counter tile; writes tile totals
__global__ __launch_bounds__(kScanBlock) void scan_tiles_kernel(const unsigned long long *__restrict__ deg, int64_t N,
                                                                int64_t *__restrict__ rowptr,
                                                                unsigned long long *__restrict__ tile_sum)
{
    __shared__ unsigned long long wave_tot[kScanBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    unsigned long long v[kScanItems], run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < N) ? deg[base + k] : 0ULL;
        run += v[k];
    }
    // inclusive scan of per-thread totals across the wave
    unsigned long long inc = run;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) wave_tot[wv] = inc;
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < wv; ++w) pre += wave_tot[w];
    unsigned long long ex = pre + inc - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < N) rowptr[base + k] = (int64_t)ex;
        ex += v[k];
    }
    if (threadIdx.x == kScanBlock - 1) tile_sum[blockIdx.x] = pre + inc;
}

// single block: exclusive scan of the tile totals in place; writes the grand total to rowptr[N]
__global__ __launch_bounds__(kScanBlock) void scan_totals_kernel(unsigned long long *__restrict__ tile_sum, int64_t tiles,
                                                                 int64_t *__restrict__ rowptr, int64_t N)
{
    __shared__ unsigned long long wave_tot[kScanBlock / kWave];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    for (int64_t start = 0; start < tiles; start += kScanBlock) {
        const int64_t i = start + threadIdx.x;
        const unsigned long long x = i < tiles ? tile_sum[i] : 0ULL;
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
        if (i < tiles) tile_sum[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry_s = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowptr[N] = (int64_t)carry_s;
}

__global__ __launch_bounds__(256) void add_tile_offsets_kernel(int64_t *__restrict__ rowptr, int64_t N,
                                                               const unsigned long long *__restrict__ tile_sum)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        rowptr[i] += (int64_t)tile_sum[i / kScanTile];
}

__global__ __launch_bounds__(256) void fill_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                   int64_t E, int64_t N, const int64_t *__restrict__ rowptr,
                                                   unsigned long long *__restrict__ cursor, int32_t *__restrict__ col)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = src[e], d = dst[e];
        if ((uint64_t)s >= (uint64_t)N || (uint64_t)d >= (uint64_t)N) continue;
        const unsigned long long pos = atomicAdd(&cursor[d], 1ULL);
        col[rowptr[d] + (int64_t)pos] = (int32_t)s;
    }
}

inline int64_t csr_tiles(int64_t N) { return (N + kScanTile - 1) / kScanTile; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int csr_grid(int64_t items)
{
    int64_t g = (items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

}  // namespace ss

// workspace layout: [deg/cursor: N u64][tile sums: tiles u64]
extern "C" size_t ss_csr_workspace_bytes(int64_t N, int64_t E)
{
    (void)E;
    if (N < 0) return 0;
    return ss::align256((size_t)(N + 1) * 8) + ss::align256((size_t)(ss::csr_tiles(N) + 1) * 8);
}

extern "C" int ss_csr_build(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                            int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace ss;
    if (N < 0 || E < 0 || N >= ((int64_t)1 << 31) || !rowptr) return SS_ERR_INVALID_ARG;
    if (E > 0 && (!src || !dst || !col)) return SS_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < ss_csr_workspace_bytes(N, E)) return SS_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    auto *deg = reinterpret_cast<unsigned long long *>(workspace);
    auto *tile_sum = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + align256((size_t)(N + 1) * 8));
    const int64_t tiles = csr_tiles(N);
    if (hipMemsetAsync(deg, 0, (size_t)(N + 1) * 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
    if (N == 0) {
        if (hipMemsetAsync(rowptr, 0, 8, stream) != hipSuccess) return SS_ERR_LAUNCH;
        return SS_OK;
    }
    if (E > 0) {
        hipLaunchKernelGGL(degree_kernel, dim3(csr_grid(E)), dim3(256), 0, stream, src, dst, E, N, deg, err_flag);
        SS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)tiles), dim3(kScanBlock), 0, stream, deg, N, rowptr, tile_sum);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, stream, tile_sum, tiles, rowptr, N);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(add_tile_offsets_kernel, dim3(csr_grid(N)), dim3(256), 0, stream, rowptr, N, tile_sum);
    SS_LAUNCH_CHECK();
    if (E > 0) {
        if (hipMemsetAsync(deg, 0, (size_t)N * 8, stream) != hipSuccess) return SS_ERR_LAUNCH;  // reuse as cursors
        hipLaunchKernelGGL(fill_kernel, dim3(csr_grid(E)), dim3(256), 0, stream, src, dst, E, N, rowptr, deg, col);
        SS_LAUNCH_CHECK();
    }
    return SS_OK;
}
