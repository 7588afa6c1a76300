// gups.hip -- the random-access ceiling of one MI355X for the access pattern of the chain kernel: independent 16 / 32 / 64-byte loads at
// uniformly random, naturally aligned addresses of a large table, every lane its own address, enough waves to fill the chip.
// Prints requests/s and the bytes they carry.   hipcc --offload-arch=gfx950 -O3 tools/micro/gups.hip -o tools/micro/gups && tools/micro/gups
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// NV uint4 per request (16 * NV bytes), R requests per thread, UN independent requests in flight per lane
template <int NV, int UN> __global__ __launch_bounds__(256) void k_gups(const uint4 *tab, uint64_t nunits, int R, uint32_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < R; r += UN) {
        uint4 v[UN][NV];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const uint64_t i = __umul64hi(mix(tid * 0x9E3779B97F4A7C15ULL + (uint64_t)(r + u)), nunits) * NV;
#pragma unroll
            for (int q = 0; q < NV; q++) v[u][q] = tab[i + q];
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int q = 0; q < NV; q++) acc ^= v[u][q].x ^ v[u][q].w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int NV, int UN> static void run(const uint4 *tab, uint64_t bytes, int blocks, int R, uint32_t *out)
{
    const uint64_t nunits = bytes / (16ull * NV);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gups<NV, UN>), dim3(blocks), dim3(256), 0, 0, tab, nunits, R, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gups<NV, UN>), dim3(blocks), dim3(256), 0, 0, tab, nunits, R, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double req = (double)blocks * 256.0 * R;
    printf("footprint %6.1f GiB  %2d B/request  %d in flight/lane  blocks %6d : %7.2f G requests/s  %8.1f GB/s\n", bytes / 1073741824.0, 16 * NV, UN, blocks,
           req / (ms * 1e-3) / 1e9, req * 16.0 * NV / (ms * 1e-3) / 1e9);
}
int main(int argc, char **argv)
{
    const double gib[] = { 0.25, 2, 16, 64 };
    uint32_t *out; CK(hipMalloc(&out, 64));
    for (double g : gib) {
        const uint64_t bytes = (uint64_t)(g * 1073741824.0);
        uint4 *tab; if (hipMalloc(&tab, bytes) != hipSuccess) { printf("cannot allocate %.1f GiB\n", g); continue; }
        CK(hipMemset(tab, 1, bytes));
        for (int blocks : { 4096, 16384 }) {
            run<1, 1>(tab, bytes, blocks, 64, out);
            run<1, 4>(tab, bytes, blocks, 64, out);
            run<2, 1>(tab, bytes, blocks, 64, out);
            run<2, 4>(tab, bytes, blocks, 64, out);
            run<4, 2>(tab, bytes, blocks, 64, out);
        }
        CK(hipFree(tab));
    }
    return 0;
}
