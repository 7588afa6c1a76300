// Dependent random-load latency on one MI355X: a single lane (and a full wave of independent lanes) chases a random cycle over a
// footprint of F bytes, 64-B granules.  Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat ; run: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
__global__ void chase(const uint64_t *buf, uint64_t start_stride, uint64_t n, int hops, int lanes, uint64_t *out, long long *cycles)
{
    const int lane = threadIdx.x;
    uint64_t p = ((uint64_t)lane * start_stride + (uint64_t)blockIdx.x * 7919u) % n;
    const long long t0 = wall_clock64();
    if (lane < lanes) for (int i = 0; i < hops; i++) p = buf[p * 8];       // granule p holds the index of the next granule
    const long long t1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = p;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}
int main()
{
    const size_t sizes[] = { (size_t)16 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30 };
    for (size_t F : sizes) {
        const size_t n = F / 64;
        std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937_64 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint64_t> h(n * 8, 0);
        for (size_t i = 0; i < n; i++) h[(size_t)perm[i] * 8] = perm[(i + 1) % n];          // one big cycle
        uint64_t *d, *out; long long *cyc;
        hipMalloc(&d, F); hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&cyc, 4096 * 8);
        hipMemcpy(d, h.data(), F, hipMemcpyHostToDevice);
        for (int lanes : { 1, 64 }) for (int blocks : { 1, 1024 }) {
            const int hops = 2000;
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, (uint64_t)(n / 64 / 2 + 1), (uint64_t)n, hops, lanes, out, cyc);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("footprint %6zu MiB  lanes %2d  waves %4d : %.0f ns per dependent hop\n", F >> 20, lanes, blocks, (double)c * 10.0 / hops);
        }
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
