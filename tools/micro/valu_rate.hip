// valu_rate.hip -- issue cost of the integer instructions the chain kernel is made of, on one MI355X: dependent chains of one opcode,
// 8 waves per SIMD (enough to hide the dependency), cycles per wave-instruction per SIMD from the wall clock of the launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int OP> __global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters, uint32_t c0)
{
    uint32_t a = threadIdx.x * 2654435761u + 1u, b = blockIdx.x + 7u, c = c0 | 1u, d = a ^ 0x5bd1e995u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { a = a + c; b = b + a; d = d + b; c = c + d; }                                  // v_add_u32
            if (OP == 1) { a = a * c; b = b * a; d = d * b; c = c * d; }                                  // v_mul_lo_u32
            if (OP == 2) { a = __umulhi(a, c); b = __umulhi(b, a | 1u); d = __umulhi(d, b | 1u); c = __umulhi(c, d | 1u) | 1u; }   // v_mul_hi_u32
            if (OP == 3) { a = __umul24(a, c); b = __umul24(b, a); d = __umul24(d, b); c = __umul24(c, d); }          // v_mul_u32_u24
            if (OP == 4) { a = __builtin_amdgcn_alignbit(a, c, b); b = __builtin_amdgcn_alignbit(b, a, d); d = __builtin_amdgcn_alignbit(d, b, a); c = __builtin_amdgcn_alignbit(c, d, b); }   // v_alignbit_b32
            if (OP == 5) { a = a < c ? a : c + 1u; b = b < a ? b : a + 3u; d = d < b ? d : b + 5u; c = c < d ? c : d + 7u; }   // v_min_u32 + v_add
            if (OP == 6) { a = __popc(a ^ c) + b; b = __popc(b ^ a) + d; d = __popc(d ^ b) + c; c = __popc(c ^ d) + a; }      // v_xor + v_bcnt (bcnt adds)
            if (OP == 7) { a = a ^ (c >> 15) ^ b; b = b ^ (a >> 13) ^ d; d = d ^ (b >> 16) ^ c; c = c ^ (d >> 11) ^ a; }      // v_lshrrev + v_xor3
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[0] = a;
}
template <int OP> static void run(const char *name, int per_iter, uint32_t *out)
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int blocks = pr.multiProcessorCount * 8, iters = 2000;           // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = 8.0 * iters * 16.0 * per_iter;          // wave-instructions one SIMD issues
    const double clk = pr.clockRate * 1e3;                                  // Hz (peak)
    printf("%-28s %8.3f ms   %.2f cycles per wave-instruction per SIMD at %.2f GHz peak clock\n", name, ms, ms * 1e-3 * clk / insts_per_simd, clk / 1e9);
}
int main()
{
    uint32_t *out; CK(hipMalloc(&out, 64));
    run<0>("v_add_u32", 4, out);
    run<1>("v_mul_lo_u32", 4, out);
    run<2>("v_mul_hi_u32 (+or)", 8, out);
    run<3>("v_mul_u32_u24", 4, out);
    run<4>("v_alignbit_b32", 4, out);
    run<5>("v_min_u32 + v_add_u32", 8, out);
    run<6>("v_xor + v_bcnt_u32_b32", 8, out);
    run<7>("v_lshrrev + v_xor3", 8, out);
    return 0;
}
