// What a launch of 2^32 and more work-items does on this platform (round 5: stage II's window passes over 402 M events were cut short silently).
// hipcc --offload-arch=gfx950 -O2 tools/micro/grid_limit.hip -o tools/micro/grid_limit && tools/micro/grid_limit
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_count(unsigned long long *cnt, unsigned int *maxb) { if (threadIdx.x == 0) { atomicAdd(cnt, 1ULL); atomicMax(maxb, blockIdx.x); } }
int main()
{
    unsigned long long *d; unsigned int *m; hipMalloc(&d, 8); hipMalloc(&m, 4);
    const unsigned long long blocks[] = { 1ull << 23, (1ull << 24) - 1, 1ull << 24, (1ull << 24) + 1, 1ull << 25, 39000000ull, 1ull << 28 };
    for (unsigned long long nb : blocks) for (int bs : { 64, 256 }) {
        hipMemset(d, 0, 8); hipMemset(m, 0, 4);
        hipLaunchKernelGGL(k_count, dim3((unsigned)nb), dim3(bs), 0, 0, d, m);
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = hipDeviceSynchronize();
        unsigned long long h = 0; unsigned int hm = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hm, m, 4, hipMemcpyDeviceToHost);
        printf("blocks %10llu x %3d threads = %12llu work-items: launch -> %s, sync -> %s, blocks that ran %llu, highest blockIdx %u\n", nb, bs, nb * bs, hipGetErrorName(e1), hipGetErrorName(e2), h, hm);
    }
    return 0;
}
