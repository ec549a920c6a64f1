// ss_init.hip -- hop-0 sketches and int64<->u32 MinHash repacking.
// Store-bound streaming kernels: every lane writes one 16-byte chunk.
#include "ss_common.hpp"

namespace ss {

// out[i, 4c..4c+3] = ((a_j * hv_i + b_j) mod 2^64 mod (2^61-1)) & 0xFFFFFFFF   (hashing.py:118-124)
__global__ __launch_bounds__(256) void minhash_init_kernel(uint32_t *__restrict__ out, int64_t first_node, int64_t n,
                                                           const uint64_t *__restrict__ a,
                                                           const uint64_t *__restrict__ b, int P)
{
    const int chunks = P >> 2;
    const int64_t total = n * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / chunks;
        const int c = (int)(t - i * chunks);
        const uint64_t hv = hash_u64((uint64_t)(first_node + i + 1));
        u32x4 v;
        v.x = (uint32_t)mod_mersenne61(a[4 * c + 0] * hv + b[4 * c + 0]);
        v.y = (uint32_t)mod_mersenne61(a[4 * c + 1] * hv + b[4 * c + 1]);
        v.z = (uint32_t)mod_mersenne61(a[4 * c + 2] * hv + b[4 * c + 2]);
        v.w = (uint32_t)mod_mersenne61(a[4 * c + 3] * hv + b[4 * c + 3]);
        *reinterpret_cast<u32x4 *>(out + i * P + 4 * c) = v;
    }
}

// one non-zero register per row: reg[hv & (m-1)] = (64-p) - bit_length(hv >> p) + 1  (hashing.py:126-137)
__global__ __launch_bounds__(256) void hll_init_kernel(uint8_t *__restrict__ out, int64_t first_node, int64_t n, int p)
{
    const int64_t chunks = (int64_t)1 << (p - 4);  // 16-byte chunks per row
    const int64_t total = n * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t >> (p - 4);
        const int64_t c = t & (chunks - 1);
        const uint64_t hv = hash_u64((uint64_t)(first_node + i + 1));
        const uint64_t idx = hv & (((uint64_t)1 << p) - 1);
        const uint64_t bits = hv >> p;
        const int bl = bits ? 64 - __builtin_clzll(bits) : 0;
        const uint32_t rank = (uint32_t)((64 - p) - bl + 1);  // always in [1, 64-p+1]
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((int64_t)(idx >> 4) == c) {
            const uint32_t byte = (uint32_t)(idx & 15u);
            const uint32_t w = rank << (8u * (byte & 3u));
            switch (byte >> 2) {
                case 0: v.x = w; break;
                case 1: v.y = w; break;
                case 2: v.z = w; break;
                default: v.w = w; break;
            }
        }
        *reinterpret_cast<u32x4 *>(out + (i << p) + (c << 4)) = v;
    }
}

__global__ __launch_bounds__(256) void pack_kernel(const int64_t *__restrict__ in, uint32_t *__restrict__ out, int64_t count)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; 4 * t < count; t += (int64_t)gridDim.x * blockDim.x) {
        if (4 * t + 3 < count) {
            u32x4 v;
            v.x = (uint32_t)in[4 * t + 0];
            v.y = (uint32_t)in[4 * t + 1];
            v.z = (uint32_t)in[4 * t + 2];
            v.w = (uint32_t)in[4 * t + 3];
            *reinterpret_cast<u32x4 *>(out + 4 * t) = v;
        } else {
            for (int64_t k = 4 * t; k < count; ++k) out[k] = (uint32_t)in[k];
        }
    }
}

__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t *__restrict__ in, int64_t *__restrict__ out, int64_t count)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; 4 * t < count; t += (int64_t)gridDim.x * blockDim.x) {
        if (4 * t + 3 < count) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(in + 4 * t);
            out[4 * t + 0] = (int64_t)v.x;
            out[4 * t + 1] = (int64_t)v.y;
            out[4 * t + 2] = (int64_t)v.z;
            out[4 * t + 3] = (int64_t)v.w;
        } else {
            for (int64_t k = 4 * t; k < count; ++k) out[k] = (int64_t)in[k];
        }
    }
}

inline int grid_for(int64_t work_items)
{
    int64_t g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;  // 16 blocks per CU, grid-stride beyond
    return (int)g;
}

}  // namespace ss

extern "C" int ss_minhash_init(uint32_t *out, int64_t first_node, int64_t n, const uint64_t *a, const uint64_t *b,
                               int32_t P, void *stream)
{
    if (n < 0 || P <= 0 || (P & 3) || first_node < 0) return SS_ERR_INVALID_ARG;
    if (n == 0) return SS_OK;
    if (!out || !a || !b) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ss::minhash_init_kernel, dim3(ss::grid_for(n * (P >> 2))), dim3(256), 0, (hipStream_t)stream, out,
                       first_node, n, a, b, (int)P);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_hll_init(uint8_t *out, int64_t first_node, int64_t n, int32_t p, void *stream)
{
    if (n < 0 || p < 4 || p > 16 || first_node < 0) return SS_ERR_INVALID_ARG;
    if (n == 0) return SS_OK;
    if (!out) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ss::hll_init_kernel, dim3(ss::grid_for(n << (p - 4))), dim3(256), 0, (hipStream_t)stream, out,
                       first_node, n, (int)p);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_pack_minhash(const int64_t *in, uint32_t *out, int64_t count, void *stream)
{
    if (count < 0) return SS_ERR_INVALID_ARG;
    if (count == 0) return SS_OK;
    if (!in || !out) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ss::pack_kernel, dim3(ss::grid_for((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, count);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" int ss_unpack_minhash(const uint32_t *in, int64_t *out, int64_t count, void *stream)
{
    if (count < 0) return SS_ERR_INVALID_ARG;
    if (count == 0) return SS_OK;
    if (!in || !out) return SS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ss::unpack_kernel, dim3(ss::grid_for((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, count);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
