// micro-benchmark: is the h = 3 query on HBM-resident tables limited by the NUMBER OF SEGMENTS a pair touches?
// A pair reads the rows of two nodes in 2h tables: at h = 3, P = 128, p = 8 that is 12 separate segments (6 x 512 B MinHash rows,
// 6 x 256 B HLL rows) in 6 tables spread over N * 2 304 B (6.7 GB at ogbl-citation2 size).  A node-major layout -- one 2 304-byte
// record per node -- would make it 2 contiguous records.  Same bytes, same number of 16-byte loads per lane (16 lanes per pair as
// in ss::pair_features_kernel, 18 loads per lane and pair), same grid; only the addresses differ:
//   tables   : six arrays, row k of node n at base_k + n * row_bytes_k                     (what the engine has today)
//   records  : one array, node n at base + n * 2304, the six rows behind each other       (node-major)
//   records4k: the same with records padded to 4 096 B (a record never straddles a 4 KiB page; 1.78x the footprint)
// The kernel xors what it loads (no estimator, no features): what is measured is the memory system's answer to the address pattern.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/record_gather tools/micro/record_gather.hip && /tmp/record_gather
#define HIP_DISABLE_WARN_UNUSED_RESULT 1
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRow = 16;  // lanes per pair

struct Tables {
    const u32x4 *mh[3];   // 32 chunks of 16 B per row
    const u32x4 *hll[3];  // 16 chunks of 16 B per row
};

// MODE 0: six tables; 1: records of `stride16` chunks of 16 B (144 = packed, 256 = padded to 4 KiB)
template <int MODE>
__global__ __launch_bounds__(256) void gather_pairs(Tables t, const u32x4 *__restrict__ rec, int64_t stride16, const int32_t *__restrict__ ids,
                                                    int64_t pairs, u32x4 *__restrict__ out)
{
    const int l = threadIdx.x & (kRow - 1);
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kRow, stride = ((int64_t)gridDim.x * blockDim.x) / kRow;
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t q = g0; q < pairs; q += stride) {
        const int64_t u = ids[2 * q], v = ids[2 * q + 1];
        u32x4 x[18];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t n = s ? v : u;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (MODE == 0) {
                    x[9 * s + 3 * k + 0] = t.mh[k][n * 32 + l];
                    x[9 * s + 3 * k + 1] = t.mh[k][n * 32 + 16 + l];
                    x[9 * s + 3 * k + 2] = t.hll[k][n * 16 + l];
                } else {  // record: [mh1 | mh2 | mh3 | hll1 | hll2 | hll3]
                    x[9 * s + 3 * k + 0] = rec[n * stride16 + 32 * k + l];
                    x[9 * s + 3 * k + 1] = rec[n * stride16 + 32 * k + 16 + l];
                    x[9 * s + 3 * k + 2] = rec[n * stride16 + 96 + 16 * k + l];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) acc ^= x[i];
    }
    if (acc.x == 0x12345u) out[threadIdx.x] = acc;  // (never true: keeps the loads alive)
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    u32x4 *out;
    hipMalloc(&out, 4096);
    printf("%-10s %10s %10s %9s %8s %10s %12s\n", "layout", "nodes", "footprint", "pairs", "wg/CU", "us", "TB/s (rows)");
    for (int64_t nodes : {2927963ll, 576289ll, 235868ll}) {
        const int64_t pairs = 4 << 20;
        std::vector<int32_t> h(2 * pairs);
        std::mt19937_64 rng(11);
        for (auto &x : h) x = (int32_t)(rng() % nodes);
        int32_t *ids;
        hipMalloc(&ids, h.size() * 4);
        hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int layout = 0; layout < 3; ++layout) {
            Tables t = {};
            u32x4 *rec = nullptr;
            const int64_t stride16 = layout == 2 ? 256 : 144;
            size_t bytes = 0;
            void *bufs[6] = {};
            if (layout == 0) {
                for (int k = 0; k < 3; ++k) {
                    hipMalloc(&bufs[k], nodes * 512);
                    hipMemset(bufs[k], 1 + k, nodes * 512);
                    hipMalloc(&bufs[3 + k], nodes * 256);
                    hipMemset(bufs[3 + k], 4 + k, nodes * 256);
                    t.mh[k] = (const u32x4 *)bufs[k];
                    t.hll[k] = (const u32x4 *)bufs[3 + k];
                }
                bytes = nodes * 2304;
            } else {
                bytes = nodes * stride16 * 16;
                if (hipMalloc(&bufs[0], bytes) != hipSuccess) { printf("(records of %lld B: allocation failed)\n", (long long)stride16 * 16); continue; }
                hipMemset(bufs[0], 3, bytes);
                rec = (u32x4 *)bufs[0];
            }
            for (int per_cu : {8, 16, 32}) {
                const int grid = 256 * per_cu;
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    hipEventRecord(e0);
                    if (layout == 0) gather_pairs<0><<<grid, 256>>>(t, rec, stride16, ids, pairs, out);
                    else gather_pairs<1><<<grid, 256>>>(t, rec, stride16, ids, pairs, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                printf("%-10s %10lld %8.2f GB %9lld %8d %10.1f %12.2f\n", layout == 0 ? "tables" : layout == 1 ? "records" : "records4k", (long long)nodes,
                       bytes / 1e9, (long long)pairs, per_cu, best * 1e3, pairs * 2.0 * 2304 / best / 1e9);
            }
            for (void *b : bufs)
                if (b) hipFree(b);
        }
        hipFree(ids);
    }
    // where hipMalloc puts a large allocation: base alignment of a 1 GiB block (the fragment size the driver maps it with is not exposed;
    // a base aligned to 2 MiB or more is what allows 2 MiB fragments)
    void *big;
    hipMalloc(&big, 1ull << 30);
    printf("hipMalloc(1 GiB) base %p: aligned to %llu KiB\n", big, (unsigned long long)(((uintptr_t)big & -(uintptr_t)big) >> 10));
    return 0;
}
