// micro-benchmark: how does FETCH_SIZE tally streams of different request widths?  The roofline tables turn the counter into bytes as
// 2 x FETCH_SIZE (MI355X_MICROARCH.md: the 128-byte requests of dwordx4 streams are tallied at 64 B).  hll_first_hop_kernel reads its
// neighbour ids as DWORD loads of 16-lane groups (64-byte requests at 4-byte-aligned offsets); at ogbl-ppa size the ids are half of
// its bytes and its "PMC bytes / algorithmic" came out at 1.19 - 1.23.  Is that traffic, or the correction factor
// applied to a stream it does not hold for?  Three kernels stream the same 170 MB once:
//   stream_x4        lane i of a wavefront reads 16 bytes (dwordx4): 1 KiB per wavefront-instruction
//   stream_dword     lane i reads 4 bytes: 256 B per wavefront-instruction
//   stream_dword16   as hll_first_hop_kernel: a 16-lane group reads a "row" of 74 dwords in steps of 16 from a 4-byte-aligned start
// run: hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_granularity tools/micro/fetch_granularity.hip
//      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fg -- /tmp/fetch_granularity   (then divide per kernel)
#define HIP_DISABLE_WARN_UNUSED_RESULT 1
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_x4(const u32x4 *__restrict__ p, int64_t n16, uint32_t *out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

__global__ __launch_bounds__(256) void stream_dword(const uint32_t *__restrict__ p, int64_t n4, uint32_t *out)
{
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345u) out[0] = 1;
}

__global__ __launch_bounds__(256) void stream_dword16(const uint32_t *__restrict__ p, int64_t rows, int deg, uint32_t *out)
{
    const int l = threadIdx.x & 15;
    uint32_t acc = 0;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < rows; r += ((int64_t)gridDim.x * blockDim.x) >> 4) {
        const uint32_t *row = p + r * deg;
        for (int t = l; t < deg; t += 16) acc ^= row[t];
    }
    if (acc == 0x12345u) out[0] = 1;
}

int main()
{
    const int deg = 74;
    const int64_t rows = 576289, n4 = rows * deg, n16 = n4 / 4;
    uint32_t *p, *out;
    hipMalloc(&p, n4 * 4 + 64);
    hipMemset(p, 1, n4 * 4 + 64);
    hipMalloc(&out, 64);
    for (int rep = 0; rep < 3; ++rep) {
        stream_x4<<<4096, 256>>>((const u32x4 *)p, n16, out);
        stream_dword<<<4096, 256>>>(p, n4, out);
        stream_dword16<<<4096, 256>>>(p, rows, deg, out);
    }
    hipDeviceSynchronize();
    printf("streamed %.1f MB per launch\n", n4 * 4 / 1e6);
    return 0;
}
