// any_order.hip -- does hipExtAnyOrderLaunch let a kernel start beside its predecessor on the same stream (gfx950)?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for the module-launch API; this measures what happens.
// Two spin kernels of ~200 us that each fill a quarter of the chip: back to back they take ~400 us, side by side ~200 us.
// A third case checks the ordering argument the engine would rely on: K0 (normal) -> H (normal) -> R (any order): R must still
// see everything K0 wrote, because R is launched after H and H waited for K0.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin(long long cycles, int *out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1);
}
__global__ void fill(int *p, int n, int v)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
__global__ void check(const int *p, int n, int v, int *bad)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (p[i] != v) atomicAdd(bad, 1);
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    int *d;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long cyc = 20000;  // wall_clock64 ticks at 100 MHz: 200 us
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, cyc, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, cyc, d);
            else hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%s: %.1f us for two 200 us kernels\n", mode ? "second launch with hipExtAnyOrderLaunch" : "two normal launches", best * 1e3f);
    }
    // ordering: K0 fills, H spins (normal: waits for K0), R checks with the any-order flag
    const int n = 64 << 20;
    int *buf, *bad;
    hipMalloc(&buf, (size_t)n * 4);
    hipMalloc(&bad, 4);
    int total_bad = 0;
    for (int rep = 0; rep < 50; ++rep) {
        hipMemsetAsync(bad, 0, 4, s);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s, buf, n, rep + 1);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, 2000LL, d);
        hipExtLaunchKernelGGL(check, dim3(4096), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const int *)buf, n, rep + 1, bad);
        int h = 0;
        hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        total_bad += h;
    }
    printf("ordering K0 -> H -> R(any order): %d stale words seen by R over 50 rounds of 256 MB\n", total_bad);
    return 0;
}
