// micro-benchmark: what the memory system delivers for the access pattern of the MinHash table hop -- whole 512-byte rows of a
// table gathered in random order, 16 bytes per lane (one row per 32-lane half wavefront), 2 - 24 rows in flight per lane (few in flight: eight wavefronts per SIMD; many: four), reduced
// with v_min_u32 -- against the table size (Infinity-Cache resident vs HBM resident).  No CSR, no output rows: a ceiling.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_ceiling tools/micro/gather_ceiling.hip && /tmp/gather_ceiling
#define HIP_DISABLE_WARN_UNUSED_RESULT 1
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// rows_per_item rows gathered per 32-lane group, `items` groups in all (grid-stride); INFLIGHT row loads per lane before the first min
template <bool WRITE, int INFLIGHT>
__global__ __launch_bounds__(256) void gather_rows(const u32x4 *__restrict__ table, const int32_t *__restrict__ ids, int64_t items, int per_item,
                                                   u32x4 *__restrict__ out)
{
    const int lane = threadIdx.x & 31;
    const int64_t group0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5, stride = ((int64_t)gridDim.x * blockDim.x) >> 5;
    u32x4 acc = {~0u, ~0u, ~0u, ~0u};
    for (int64_t it = group0; it < items; it += stride) {
        const int32_t *my = ids + it * per_item;
        if (WRITE) acc = u32x4{~0u, ~0u, ~0u, ~0u};
        for (int k = 0; k < per_item; k += INFLIGHT) {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) v[u] = table[(int64_t)my[k + u] * 32 + lane];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) {
                acc.x = min(acc.x, v[u].x); acc.y = min(acc.y, v[u].y); acc.z = min(acc.z, v[u].z); acc.w = min(acc.w, v[u].w);
            }
        }
        if (WRITE) out[it * 32 + lane] = acc;  // the item's own 512-byte output row
    }
    if (!WRITE && acc.x == 12345u) out[threadIdx.x] = acc;  // (never true: keeps the loads alive)
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int64_t n_items_max = 256ll << 10;
    u32x4 *out; hipMalloc(&out, n_items_max * 512);
    printf("%10s %9s %6s %9s %5s %14s\n", "table MB", "rows/item", "write", "inflight", "wg/CU", "TB/s (gathers + output rows)");
    for (int per_item : {12, 24, 72})
    for (int write = 0; write < 2; ++write)
    for (int inflight : {2, 4, 12, 24})
    for (int per_cu : {8, 16, 32})
    for (double mb : {30.0, 60.0, 120.0, 200.0, 295.0, 600.0, 1500.0}) {
        const int64_t rows = (int64_t)(mb * 1e6 / 512);
        const int64_t n_items = n_items_max;             // 256 Ki x 12 x 512 B = 1.6 GB per launch (the collab-size hop moves 1.46 GB)
        u32x4 *table; int32_t *ids;
        hipMalloc(&table, rows * 512); hipMemset(table, 1, rows * 512);
        std::vector<int32_t> h(n_items * per_item);
        std::mt19937_64 rng(7);
        for (auto &x : h) x = (int32_t)(rng() % rows);
        hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        if (inflight > per_item) { hipFree(table); hipFree(ids); continue; }
        const int grid = 256 * per_cu;
        float best_g = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
#define LAUNCH(W, F) gather_rows<W, F><<<grid, 256>>>(table, ids, n_items, per_item, out)
            if (write) { if (inflight == 2) LAUNCH(true, 2); else if (inflight == 4) LAUNCH(true, 4); else if (inflight == 12) LAUNCH(true, 12); else LAUNCH(true, 24); }
            else { if (inflight == 2) LAUNCH(false, 2); else if (inflight == 4) LAUNCH(false, 4); else if (inflight == 12) LAUNCH(false, 12); else LAUNCH(false, 24); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best_g) best_g = ms;
        }
        printf("%10.0f %9d %6d %9d %5d %14.2f\n", mb, per_item, write, inflight, per_cu, n_items * (per_item + write) * 512.0 / best_g / 1e9);
        hipFree(table); hipFree(ids);
    }
    return 0;
}
