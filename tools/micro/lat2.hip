// Dependent random-load latency on one MI355X under LOAD, at the footprints k_steps works on: every wave chases its own pseudo-random
// cycle p -> (a p + c) mod n over a table of n 64-byte granules (n a power of two: full period), `lanes` lanes of the wave active, `waves`
// waves resident.  Prints ns per dependent hop and the aggregate request rate.  The chain kernel's step is ~4 such hops (bitmap word, table
// slot, claim word, read), so (hops per step) x (ns per hop at 8 waves per SIMD) is its floor whatever the instruction count.
// Build: hipcc --offload-arch=gfx950 -O3 lat2.hip -o lat2 ; run: ./lat2 [GiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
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
int main(int argc, char **argv)
{
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
