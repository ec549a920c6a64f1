// ss_digest.hip -- content digest of a sketch table (one streaming pass), for the multi-GPU builds.
//
// The reference holds ONE table (hashing.py:139-165).  The multi-GPU builds of this engine leave a replica on every rank -- replicated
// (every rank computes everything), row-sharded with an exchange, or peer-write: every rank's kernels store their rows into all
// ranks' IPC-mapped tables, so most of a rank's table arrives through OTHER GPUs' stores.  Whether those stores are all visible to
// the rank's next kernel is what dist.verify_replicas checks after the first such build: every rank digests each of its 2h + 1
// tables, the digests are all-gathered, and a rank whose replica differs makes every rank raise (and fall back to the exchange
// form) instead of querying a silently wrong table.
#include "ss_common.hpp"

namespace ss {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ULL;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ULL;
    x ^= x >> 32;
    return x;
}

// 4 chunks (64 B) per lane and iteration in flight; sum / xor are order independent, so lanes, waves and workgroups combine in any
// order: DPP-free wave reduction through __shfl_xor (64-bit), one pair of global atomics per wavefront
__global__ __launch_bounds__(256) void table_digest_kernel(const u32x4 *__restrict__ data, int64_t chunks, unsigned long long *__restrict__ out)
{
    uint64_t sum = 0, x = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < chunks; i += 4 * stride) {
        u32x4 c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_nontemporal_load(data + i + k * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t lo = ((uint64_t)c[k].y << 32) | c[k].x, hi = ((uint64_t)c[k].w << 32) | c[k].z;
            const uint64_t h = mix64(lo ^ ((uint64_t)(i + k * stride) * 0x9E3779B97F4A7C15ULL)) + mix64(hi + (uint64_t)(i + k * stride));
            sum += h;
            x ^= h;
        }
    }
    for (; i < chunks; i += stride) {
        const u32x4 c = data[i];
        const uint64_t lo = ((uint64_t)c.y << 32) | c.x, hi = ((uint64_t)c.w << 32) | c.z;
        const uint64_t h = mix64(lo ^ ((uint64_t)i * 0x9E3779B97F4A7C15ULL)) + mix64(hi + (uint64_t)i);
        sum += h;
        x ^= h;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += ((uint64_t)__shfl_xor((int)(sum >> 32), d) << 32) | (uint32_t)__shfl_xor((int)sum, d);
        x ^= ((uint64_t)__shfl_xor((int)(x >> 32), d) << 32) | (uint32_t)__shfl_xor((int)x, d);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(out, (unsigned long long)sum);
        atomicXor(out + 1, (unsigned long long)x);
    }
}

}  // namespace ss

extern "C" int ss_table_digest(const void *data, int64_t bytes, uint64_t *out, void *stream)
{
    if (bytes < 0 || !out || (bytes > 0 && !data)) return SS_ERR_INVALID_ARG;
    if ((bytes & 15) || (reinterpret_cast<uintptr_t>(data) & 15)) return SS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 16, s) != hipSuccess) return SS_ERR_LAUNCH;
    if (bytes == 0) return SS_OK;
    const int64_t chunks = bytes >> 4;
    int64_t blocks = (chunks + 4 * 256 - 1) / (4 * 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ss::table_digest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const ss::u32x4 *>(data), chunks,
                       reinterpret_cast<unsigned long long *>(out));
    SS_LAUNCH_CHECK();
    return SS_OK;
}
