// devutil.h -- device-side helpers shared by stage1.hip and stage2.hip (wave = 64 lanes on gfx950)
#pragma once
#include "internal.h"

// ------------------------------------------------------------------------------------------------ launch geometry
// A grid is dispatched with its size in WORK-ITEMS per dimension as a 32-bit number: blocks x threads of 2^32 and more is taken modulo 2^32 WITHOUT an error
// (tools/micro/grid_limit.hip: 39 000 000 x 256 runs 5 445 568 workgroups and reports hipSuccess; an exact multiple of 2^32 is hipErrorInvalidConfiguration,
// which the next successful call erases).  Found in round 5 at 402 M probes into stage II's large bins; the decoders had it from 42 M reads per shard on.
// A thread per item: harc_grid256(n) folds the workgroups into rows of at most HARC_GRID_ROW (x 256 threads < 2^32) and EVERY such kernel takes its index
// from harc_gid() -- or from harc_gid32() when its item count is a 32-bit number (a count in (2^32 - 256, 2^32) already asks for 2^24 workgroups = a
// second row): no kernel of the library computes blockIdx.x * blockDim.x + threadIdx.x by itself outside a grid-stride loop over a bounded grid.
#define HARC_GRID_ROW 16777215u
static inline dim3 harc_grid256(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    if (b == 0) b = 1;
    if (b <= HARC_GRID_ROW) return dim3((unsigned)b);
    return dim3(HARC_GRID_ROW, (unsigned)((b + HARC_GRID_ROW - 1) / HARC_GRID_ROW));
}
// workgroups of 256 threads that are not one thread per item (a wave per item, W lanes per read): nblocks of them, folded the same way; the kernel takes
// its workgroup's number from harc_bid()
static inline dim3 harc_fold256(uint64_t nblocks)
{
    if (nblocks == 0) nblocks = 1;
    if (nblocks <= HARC_GRID_ROW) return dim3((unsigned)nblocks);
    return dim3(HARC_GRID_ROW, (unsigned)((nblocks + HARC_GRID_ROW - 1) / HARC_GRID_ROW));
}
// a workgroup of 64 threads per item (blockIdx.y * gridDim.x + blockIdx.x in the kernel): rows of 2^25 items x 64 threads = 2^31 work-items
static inline dim3 wave_grid(uint32_t n) { if (n <= (1u << 25)) return dim3(n ? n : 1u); return dim3(1u << 25, (n + (1u << 25) - 1) >> 25); }
__device__ __forceinline__ uint64_t harc_bid() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }
__device__ __forceinline__ uint64_t harc_gid() { return ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; }
// ... for an item count that is a 32-bit number: 0xFFFFFFFF -- beyond every such count's last item -- for the threads of a second row past it
__device__ __forceinline__ uint32_t harc_gid32()
{
    // one row -- every launch of fewer than 2^32 - 256 items --: at most HARC_GRID_ROW x 256 + 255 < 2^32, a wave-uniform branch.  (Without it the 64-bit index and its
    // saturation cost the streaming kernels of the index build 2.7 ms per configs[2] step: k_bin_starts 916 -> 1061 us, k_s1_bloom_keys 1205 -> 1571 us.)
    if (gridDim.y == 1) return blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g = harc_gid(); return g > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)g;
}

// ------------------------------------------------------------------------------------------------ device helpers
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// The dictionaries store and compare SCRAMBLED keys: a bijection of the 64-bit key space (three Feistel rounds of 32-bit
// multiply-xorshift -- the chain kernel is bound by instruction issue, 64 x 64 -> 128 multiplies are a tenth of its instructions) whose
// high word picks the bucket monotonically.  Sorting the reads by scrambled key groups equal k-mers just as sorting by key does, and
// leaves the bins in BUCKET order: the table is then written front to back without a single atomic (harc_dict_build).
__device__ __forceinline__ uint64_t key_scramble(uint64_t key)
{
    uint32_t a = (uint32_t)key, b = (uint32_t)(key >> 32), y;
    y = a * 0x9E3779B1u; b ^= y ^ (y >> 15);
    y = b * 0x85EBCA77u; a ^= y ^ (y >> 13);
    y = a * 0xC2B2AE3Du; b ^= y ^ (y >> 16);
    return ((uint64_t)b << 32) | a;
}
// first slot of the 64-byte bucket (4 slots of 16 bytes) a scrambled key belongs to; non-decreasing in h
__device__ __forceinline__ uint64_t bucket_slot(uint64_t h, uint64_t cap)
{
    const uint64_t nb = cap >> 2;
    if (nb >> 32) return __umul64hi(h, nb) << 2;                  // more than 2^34 slots (256 GB): not on one GPU; kept exact
    return (uint64_t)__umulhi((uint32_t)(h >> 32), (uint32_t)nb) << 2;
}

// Word and two-bit mask of a key in the bitmap in front of a stage-I dictionary (k_steps: most probes of a step find nothing; they end
// in this bitmap instead of in the table).  Word and bits come from the low word of the scrambled key (the bucket comes from the high
// word: keys that share a bucket do not share these).  The 64-byte LINE: nwin = 0 -- hashed from the key as well.  nwin > 0 (bitmaps larger
// than the Infinity Cache) -- chosen by the minimizer of the k-mer, the smallest m-mer inside it (bases re-coded so that C < T < A < G):
// the probes of one step are k-mers of the consensus at consecutive shifts, a dozen consecutive k-mers share their minimizer, so the 48
// probes of a batch fall into ~8 lines instead of 48 -- and what bounds the kernel there is the number of requests that miss L2, not
// bytes (tools/micro/gups.hip; PMC: 64 M -> 12 M misses per launch at configs[2]).
__device__ __forceinline__ void bloom_pos(uint64_t key, uint64_t hkey, uint32_t nlines, int nwin, uint32_t mmask, uint32_t *word, uint32_t *mask)
{
    uint32_t g = (uint32_t)hkey * 0x165667B1u;                   // hkey = key_scramble(key): its low word does not decide the bucket
    g ^= g >> 15;
    uint32_t hl;
    if (nwin > 0) {
        const uint32_t lo = (uint32_t)key ^ 0xAAAAAAAAu, hi = (uint32_t)(key >> 32) ^ 0xAAAAAAAAu;
        uint32_t best = 0xFFFFFFFFu;
        if (nwin == 17 && mmask == 0xFFFFFFFFu) {                    // 32-base keys, 16-base minimizers (reads of 100 bases and more): unrolled, no masks
            best = lo < hi ? lo : hi;
#pragma unroll
            for (int i = 1; i < 16; i += 3) {
                const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 2 * i), b = __builtin_amdgcn_alignbit(hi, lo, 2 * i + 2), c = __builtin_amdgcn_alignbit(hi, lo, 2 * i + 4);
                const uint32_t m = a < b ? a : b, n = c < best ? c : best;
                best = m < n ? m : n;
            }
        } else {
            const int n1 = nwin < 16 ? nwin : 16;
            for (int i = 0; i < n1; i++) { const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, 2 * i) & mmask; best = x < best ? x : best; }
            for (int i = 16; i < nwin; i++) { const uint32_t x = (hi >> (2 * i - 32)) & mmask; best = x < best ? x : best; }
        }
        hl = best * 0x9E3779B1u; hl ^= hl >> 15; hl *= 0x85EBCA77u; hl ^= hl >> 13;
    } else { hl = g ^ (g >> 16); hl *= 0x9E3779B1u; }
    *word = (__umulhi(hl, nlines) << 4) | ((g >> 10) & 15u);
    *mask = (1u << (g & 31)) | (1u << ((g >> 5) & 31));
}

// a[idx] for a small register array and a lane-varying idx; out-of-range -> 0
template <int W> __device__ __forceinline__ uint64_t sel0(const uint64_t (&a)[W], int idx)
{
    uint64_t r = 0;
#pragma unroll
    for (int k = 0; k < W; k++) r = (idx == k) ? a[k] : r;
    return r;
}

// nbits (<=64) starting at bit `off` of the W-word little-endian number a
template <int W> __device__ __forceinline__ uint64_t extract_bits(const uint64_t (&a)[W], int off, int nbits)
{
    const int wi = off >> 6, sh = off & 63;
    uint64_t lo = sel0<W>(a, wi), hi = sel0<W>(a, wi + 1);
    uint64_t v = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
    return v;
}

template <int W> __device__ __forceinline__ void shr_words(const uint64_t (&a)[W], int s, uint64_t (&o)[W])
{
    const int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int w = 0; w < W; w++) {
        uint64_t lo = sel0<W>(a, w + ws), hi = sel0<W>(a, w + ws + 1);
        o[w] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
}
template <int W> __device__ __forceinline__ void shl_words(const uint64_t (&a)[W], int s, uint64_t (&o)[W])
{
    const int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int w = 0; w < W; w++) {
        uint64_t hi = sel0<W>(a, w - ws), lo = sel0<W>(a, w - ws - 1);     // negative index -> 0
        o[w] = bs ? ((hi << bs) | (lo >> (64 - bs))) : hi;
    }
}
// word w of the mask with the low nb bits set
__device__ __forceinline__ uint64_t lowmask_word(int nb, int w)
{
    const int r = nb - 64 * w;
    return r >= 64 ? ~(uint64_t)0 : (r <= 0 ? 0 : (((uint64_t)1 << r) - 1));
}
// reverse the order of the 32 two-bit fields of x
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    x = __brevll(x);
    return ((x & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((x & 0x5555555555555555ULL) << 1);
}
// reverse complement of a packed read of L bases (complement of the 2-bit code is bitwise NOT: A0<->T3, G1<->C2)
template <int W> __device__ __forceinline__ void rc_words(const uint64_t (&a)[W], int L, uint64_t (&o)[W])
{
    uint64_t r[W];
#pragma unroll
    for (int w = 0; w < W; w++) r[w] = rev2(a[W - 1 - w]);
    const int s = 64 * W - 2 * L;                  // wave-uniform shift
    const int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int w = 0; w < W; w++) {
        uint64_t lo = sel0<W>(r, w + ws), hi = sel0<W>(r, w + ws + 1);
        uint64_t v = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        o[w] = (~v) & lowmask_word(2 * L, w);
    }
}
// spread the low 32 bits of a to the even bit positions
__device__ __forceinline__ uint64_t spread32(uint64_t a)
{
    a &= 0xFFFFFFFFULL;
    a = (a | (a << 16)) & 0x0000FFFF0000FFFFULL;
    a = (a | (a << 8)) & 0x00FF00FF00FF00FFULL;
    a = (a | (a << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    a = (a | (a << 2)) & 0x3333333333333333ULL;
    a = (a | (a << 1)) & 0x5555555555555555ULL;
    return a;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t x, int src)
{
    const uint32_t lo = __shfl((uint32_t)x, src, 64), hi = __shfl((uint32_t)(x >> 32), src, 64);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t shfl_u64_any(uint64_t x, int xormask)
{
    const uint32_t lo = __shfl_xor((uint32_t)x, xormask, 64), hi = __shfl_xor((uint32_t)(x >> 32), xormask, 64);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// exclusive prefix sum over the 64 lanes; *total = wave sum
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t *total)
{
    const int lane = threadIdx.x & 63;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    *total = __shfl(x, 63, 64);
    return x - v;
}
// block-wide exclusive scan for blockDim.x = NT (multiple of 64, <= 1024). sm must hold NT/64+1 words.
template <int NT> __device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *sm, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t wtot;
    uint32_t ex = wave_excl_scan_u32(v, &wtot);
    __syncthreads();
    if (lane == 0) sm[wv] = wtot;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) { uint32_t x = sm[k]; if (k < wv) base += x; tot += x; }
    *total = tot;
    return base + ex;
}
