// Dependent random-load latency on one MI355X under LOAD, at the footprints k_steps works on: every wave chases its own pseudo-random
// cycle p -> (a p + c) mod n over a table of n 64-byte granules (n a power of two: full period), `lanes` lanes of the wave active, `waves`
// waves resident.  Prints ns per dependent hop and the aggregate request rate.  The chain kernel's step is ~4 such hops (bitmap word, table
// slot, claim word, read), so (hops per step) x (ns per hop at 8 waves per SIMD) is its floor whatever the instruction count.
// `lat2 indep [GiB]` (round 5): the same table and launch shapes with INDEPENDENT requests -- the next granule comes from a hash of the lane's
// counter, not from the loaded value, so a lane has `un` loads in flight -- which is the ceiling of tools/micro/gups' sector mode measured by a second program.
// Build: hipcc --offload-arch=gfx950 -O3 lat2.hip -o lat2 ; run: ./lat2 [GiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
__global__ void fill(uint64_t *buf, uint64_t n, uint64_t a, uint64_t c)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i * 8] = (a * i + c) & (n - 1);
}
__global__ __launch_bounds__(256) void chase(const uint64_t *buf, uint64_t n, int hops, int lanes, uint64_t *out, long long *cycles)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint64_t p = (w * 0x9E3779B97F4A7C15ULL + (uint64_t)lane * 0xD1B54A32D192ED03ULL) & (n - 1);
    const long long t0 = wall_clock64();
    if (lane < lanes) for (int i = 0; i < hops; i++) p = buf[p * 8];
    const long long t1 = wall_clock64();
    if (p == 0xFFFFFFFFFFFFFFFFULL) out[0] = p;
    if (lane == 0) cycles[w] = t1 - t0;
}
__device__ __forceinline__ uint64_t mixl(uint64_t x) { x ^= x >> 31; x *= 0x7fb5d329728ea185ULL; x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL; x ^= x >> 33; return x; }
template <int UN> __global__ __launch_bounds__(256) void indep(const uint64_t *buf, uint64_t n, int hops, int lanes, uint64_t *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t seed = w * 64 + (uint64_t)lane;
    uint64_t acc = 0;
    if (lane < lanes) for (int i = 0; i < hops; i += UN) {
        uint64_t v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) v[u] = buf[(mixl(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(i + u)) & (n - 1)) * 8];
#pragma unroll
        for (int u = 0; u < UN; u++) acc ^= v[u];
    }
    if (acc == 0xFFFFFFFFFFFFFFFFULL) out[0] = acc;
}
template <int UN> static void run_indep(const uint64_t *d, uint64_t n, int waves, int lanes, uint64_t *out)
{
    const int hops = 256;
    hipLaunchKernelGGL((indep<UN>), dim3((waves + 3) / 4), dim3(256), 0, 0, d, n, hops, lanes, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((indep<UN>), dim3((waves + 3) / 4), dim3(256), 0, 0, d, n, hops, lanes, out);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("lat2 indep: footprint %6.2f GiB  8-byte load per 64-byte granule  %d in flight/lane  lanes %2d  waves %6d : %7.2f G requests/s\n",
           (double)n * 64 / 1073741824.0, UN, lanes, waves, (double)waves * lanes * hops / (ms * 1e-3) / 1e9);
}
int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "indep")) {
        const double gib = argc > 2 ? atof(argv[2]) : 64.0;
        uint64_t n = 1; while ((double)(n * 2) * 64.0 <= gib * 1073741824.0) n *= 2;
        uint64_t *d, *out;
        if (hipMalloc(&d, n * 64) != hipSuccess) { printf("footprint %.2f GiB: allocation failed\n", gib); return 1; }
        hipMalloc(&out, 64);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, n, 0x5851F42D4C957F2DULL | 1ULL, 0x14057B7EF767814FULL | 1ULL);
        hipDeviceSynchronize();
        for (int waves : { 8192, 65536 }) for (int lanes : { 8, 64 }) { run_indep<1>(d, n, waves, lanes, out); run_indep<2>(d, n, waves, lanes, out); run_indep<4>(d, n, waves, lanes, out); }
        hipFree(d); hipFree(out);
        return 0;
    }
    double gibs[8] = { 0.25, 2, 16, 64 }; int ng = 4;
    if (argc > 1) { ng = 0; for (int i = 1; i < argc && ng < 8; i++) gibs[ng++] = atof(argv[i]); }
    for (int g = 0; g < ng; g++) {
        uint64_t n = 1; while ((double)(n * 2) * 64.0 <= gibs[g] * 1073741824.0) n *= 2;
        uint64_t *d, *out; long long *cyc;
        if (hipMalloc(&d, n * 64) != hipSuccess) { printf("footprint %.2f GiB: allocation failed\n", gibs[g]); continue; }
        hipMalloc(&out, 64); hipMalloc(&cyc, 65536 * 8);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, n, 0x5851F42D4C957F2DULL | 1ULL, 0x14057B7EF767814FULL | 1ULL);
        hipDeviceSynchronize();
        const int wl[] = { 1, 1024, 4096, 8192 };
        for (int lanes : { 1, 8, 64 }) for (int waves : wl) {
            const int hops = waves >= 4096 ? 400 : 1000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chase, dim3((waves + 3) / 4), dim3(waves >= 4 ? 256 : 64), 0, 0, d, n, hops, lanes, out, cyc);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c[64]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
            double mean = 0; const int nw = waves < 64 ? waves : 64; for (int i = 0; i < nw; i++) mean += (double)c[i]; mean /= nw;
            printf("footprint %6.2f GiB  lanes %2d  waves %5d : %7.0f ns per dependent hop (wall clock of a wave), kernel %8.3f ms, %7.2f G requests/s\n",
                   (double)n * 64 / 1073741824.0, lanes, waves, mean * 10.0 / hops, ms, (double)waves * lanes * hops / (ms * 1e-3) / 1e9);
        }
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
