// prims.hip -- device-wide sort / scan building blocks (rocPRIM).  Everything algorithm-specific is hand-written in
// stage1.hip / stage2.hip; these are the "plain library" pieces (the role std::sort plays at reorder.cpp:305).
#include "devutil.h"
#include <rocprim/rocprim.hpp>

int harc_tmp_reserve(harc_amd_ctx *c, size_t bytes)
{
    if (bytes <= c->tmp_bytes) return HARC_AMD_OK;
    if (c->d_tmp) harc_raw_free(c, c->d_tmp);
    c->d_tmp = nullptr; c->tmp_bytes = 0;
    size_t want = bytes + (bytes >> 3) + 4096;
    RC_TRY(harc_raw_alloc(c, &c->d_tmp, want));
    c->tmp_bytes = want;
    return HARC_AMD_OK;
}

#define PRIM_CALL(call_with_tmp)                                \
    do {                                                        \
        size_t bytes = 0; void *tmp = nullptr;                  \
        HIP_TRY(call_with_tmp);                                 \
        RC_TRY(harc_tmp_reserve(c, bytes));                     \
        tmp = c->d_tmp; bytes = c->tmp_bytes;                   \
        HIP_TRY(call_with_tmp);                                 \
    } while (0)

int prim_sort_pairs_u64_u32(harc_amd_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit)
{
    if (n == 0) return HARC_AMD_OK;
    if (end_bit > 64) end_bit = 64;
    PRIM_CALL(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0u, end_bit, c->stream));
    return HARC_AMD_OK;
}

int prim_sort_keys_u64(harc_amd_ctx *c, const uint64_t *kin, uint64_t *kout, size_t n, unsigned end_bit)
{
    if (n == 0) return HARC_AMD_OK;
    if (end_bit > 64) end_bit = 64;
    if (end_bit < 1) end_bit = 1;
    PRIM_CALL(rocprim::radix_sort_keys(tmp, bytes, kin, kout, n, 0u, end_bit, c->stream));
    return HARC_AMD_OK;
}

int prim_excl_scan_u32(harc_amd_ctx *c, const uint32_t *in, uint32_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    PRIM_CALL(rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), c->stream));
    return HARC_AMD_OK;
}

struct u32_to_u64 { __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; } };
struct u8_to_u64 { __host__ __device__ uint64_t operator()(uint8_t x) const { return (uint64_t)x; } };

int prim_excl_scan_u32_to_u64(harc_amd_ctx *c, const uint32_t *in, uint64_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    auto it = rocprim::make_transform_iterator(in, u32_to_u64());
    PRIM_CALL(rocprim::exclusive_scan(tmp, bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
    return HARC_AMD_OK;
}

int prim_excl_scan_u8_to_u64(harc_amd_ctx *c, const uint8_t *in, uint64_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    auto it = rocprim::make_transform_iterator(in, u8_to_u64());
    PRIM_CALL(rocprim::exclusive_scan(tmp, bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
    return HARC_AMD_OK;
}

int prim_incl_scan_u64(harc_amd_ctx *c, const uint64_t *in, uint64_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    PRIM_CALL(rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<uint64_t>(), c->stream));
    return HARC_AMD_OK;
}

int prim_incl_max_u32(harc_amd_ctx *c, const uint32_t *in, uint32_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    PRIM_CALL(rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<uint32_t>(), c->stream));
    return HARC_AMD_OK;
}

int prim_incl_max_u64(harc_amd_ctx *c, const uint64_t *in, uint64_t *out, size_t n)
{
    if (n == 0) return HARC_AMD_OK;
    PRIM_CALL(rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<uint64_t>(), c->stream));
    return HARC_AMD_OK;
}

// ---- self-test of the launch geometry (devutil.h: harc_grid256 / harc_gid): n items, a thread each; returns how many were visited and the sum of their
// indices -- n and n (n - 1) / 2 (mod 2^64) when every item was visited exactly once.  tests/test_gpu_parity.py runs it beyond 2^32 items, where a plain
// one-dimensional grid is cut short without an error on this platform.
__global__ void k_selftest_grid(uint64_t n, unsigned long long *out)
{
    const uint64_t i = harc_gid();
    unsigned long long one = i < n ? 1ULL : 0ULL, idx = i < n ? (unsigned long long)i : 0ULL;
    for (int o = 32; o > 0; o >>= 1) { one += __shfl_xor(one, o, 64); idx += __shfl_xor(idx, o, 64); }
    if ((threadIdx.x & 63) == 0 && one) { atomicAdd(&out[0], one); atomicAdd(&out[1], idx); }
}
// ... the same through harc_gid32() (kernels whose item count is a 32-bit number; n < 2^32 only) ...
__global__ void k_selftest_grid32(uint32_t n, unsigned long long *out)
{
    const uint32_t i = harc_gid32();
    unsigned long long one = i < n ? 1ULL : 0ULL, idx = i < n ? (unsigned long long)i : 0ULL;
    for (int o = 32; o > 0; o >>= 1) { one += __shfl_xor(one, o, 64); idx += __shfl_xor(idx, o, 64); }
    if ((threadIdx.x & 63) == 0 && one) { atomicAdd(&out[0], one); atomicAdd(&out[1], idx); }
}
// ... and with FOUR LANES PER ITEM in folded workgroups (harc_fold256 / harc_bid: the geometry of k_orient and k_succ -- 4 n work-items pass 2^32 at a
// quarter of the items): lane 0 of an item's four counts it
__global__ void k_selftest_fold(uint64_t n, unsigned long long *out)
{
    const uint64_t wave = harc_bid() * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t i = wave * 16 + (uint64_t)(lane >> 2);
    const bool mine = i < n && (lane & 3) == 0;
    unsigned long long one = mine ? 1ULL : 0ULL, idx = mine ? (unsigned long long)i : 0ULL;
    for (int o = 32; o > 0; o >>= 1) { one += __shfl_xor(one, o, 64); idx += __shfl_xor(idx, o, 64); }
    if (lane == 0 && one) { atomicAdd(&out[0], one); atomicAdd(&out[1], idx); }
}
extern "C" int harc_amd_selftest_launch(harc_amd_ctx *c, uint64_t n, uint64_t *visited, uint64_t *index_sum)
{
    if (!c || !visited || !index_sum) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    unsigned long long *d = nullptr, h[6] = { 0, 0, 0, 0, 0, 0 };
    HIP_TRY(hipMalloc((void **)&d, sizeof h));
    const bool n32 = n <= 0xFFFFFFFFull;
    hipError_t e = hipMemsetAsync(d, 0, sizeof h, c->stream);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_selftest_grid, harc_grid256(n), dim3(256), 0, c->stream, n, d); e = hipGetLastError(); }
    if (e == hipSuccess && n32) { hipLaunchKernelGGL(k_selftest_grid32, harc_grid256(n), dim3(256), 0, c->stream, (uint32_t)n, d + 2); e = hipGetLastError(); }
    if (e == hipSuccess) { hipLaunchKernelGGL(k_selftest_fold, harc_fold256((n + 63) / 64), dim3(256), 0, c->stream, n, d + 4); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { harc_set_error("harc_amd_selftest_launch: %s", hipGetErrorString(e)); return HARC_AMD_ENODEVICE; }
    // the three geometries must agree; the first one that does not is what the caller gets to see
    int k = 0;
    if (n32 && (h[2] != h[0] || h[3] != h[1])) k = 2;
    else if (h[4] != h[0] || h[5] != h[1]) k = 4;
    if (k) harc_set_error("harc_amd_selftest_launch: %s visited %llu items (index sum %llx), the thread-per-item grid %llu (%llx)", k == 2 ? "harc_gid32" : "harc_fold256 / harc_bid", h[k], h[k + 1], h[0], h[1]);
    *visited = h[k]; *index_sum = h[k + 1];
    return k ? HARC_AMD_EINTERNAL : HARC_AMD_OK;
}
