// gups.hip -- the random-access ceiling of one MI355X for the access pattern of the chain kernel: independent 16 / 32 / 64-byte loads at
// uniformly random, naturally aligned addresses of a large table, every lane its own address, enough waves to fill the chip.
// Prints requests/s and the bytes they carry.   hipcc --offload-arch=gfx950 -O3 tools/micro/gups.hip -o tools/micro/gups && tools/micro/gups
// `gups sector [GiB]` (round 5): the shape of k_steps' L2 misses -- ONE 4- or 8-byte load per request at a random 64-byte granule (what a
// bitmap word, a claim word or a table slot costs: one sector), eight waves per SIMD resident -- the same shape tools/micro/lat2 measures in
// its independent mode, so that the two tools can be held against each other (profiles/r05/random_access_ceiling.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
// one LB-byte load (LB = 4 or 8) at a random 64-byte granule per request
template <int LB, int UN> __global__ __launch_bounds__(256) void k_sector(const uint32_t *tab, uint64_t ngran, int R, uint32_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < R; r += UN) {
        uint32_t v[UN][2];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const uint64_t i = __umul64hi(mix(tid * 0x9E3779B97F4A7C15ULL + (uint64_t)(r + u)), ngran) * 16;
            if (LB == 8) { const uint2 x = *reinterpret_cast<const uint2 *>(tab + i); v[u][0] = x.x; v[u][1] = x.y; }
            else { v[u][0] = tab[i]; v[u][1] = 0; }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) acc ^= v[u][0] ^ v[u][1];
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int LB, int UN> static void run_sector(const uint32_t *tab, uint64_t bytes, int blocks, int R, uint32_t *out)
{
    const uint64_t ngran = bytes / 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_sector<LB, UN>), dim3(blocks), dim3(256), 0, 0, tab, ngran, R, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_sector<LB, UN>), dim3(blocks), dim3(256), 0, 0, tab, ngran, R, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double req = (double)blocks * 256.0 * R;
    printf("gups sector: footprint %6.1f GiB  %d-byte load per 64-byte granule  %d in flight/lane  blocks %6d : %7.2f G requests/s\n", bytes / 1073741824.0, LB, UN, blocks, req / (ms * 1e-3) / 1e9);
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
    if (argc > 1 && !strcmp(argv[1], "sector")) {
        const double g = argc > 2 ? atof(argv[2]) : 64.0;
        const uint64_t bytes = (uint64_t)(g * 1073741824.0);
        uint32_t *tab; if (hipMalloc(&tab, bytes) != hipSuccess) { printf("cannot allocate %.1f GiB\n", g); return 1; }
        CK(hipMemset(tab, 1, bytes));
        for (int blocks : { 2048, 16384 }) {      // 2048 x 4 waves = one round of 8 waves per SIMD; 16384: eight rounds
            run_sector<4, 1>(tab, bytes, blocks, 256, out); run_sector<8, 1>(tab, bytes, blocks, 256, out);
            run_sector<8, 2>(tab, bytes, blocks, 256, out); run_sector<8, 4>(tab, bytes, blocks, 256, out);
        }
        CK(hipFree(tab));
        return 0;
    }
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
