// stage1.hip -- HARC stage I (hash-based read reordering) for gfx950.
//
// Reference: src/reorder.cpp.  constructdictionary :277-394 -> k_keygen2 + radix sort + k_table_place (an exact
// open-addressing key->bin table replaces BBHash: the MPHF value never reaches an output byte).  reorder() :434-703 ->
// the super-round-synchronous K-chain x S-step schedule of DESIGN.md: k_steps (one 64-lane wave per chain, the (shift,
// direction, dictionary) probes of a chain step spread over the lanes, priority = lane order, up to S steps per launch with
// the consensus counts in registers), k_resolve (smallest (step, chain) bid wins a read), k_reseed (one global descending
// cursor, reorder.cpp:652-668).  updaterefcount :863-915 -> cons_update inside k_steps.  writetofile :722-830 -> k_pages_out, k_s1_singles (+ k_orient for the in-HBM hand-over to stage II).
//
// Integer / HBM-latency bound; no MFMA.  Wave = 64 everywhere.
#include "devutil.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------ kernel argument block
struct S1Args {
    int L, maxmatch, thresh, maxsearch, Lp;
    int ds[2], de[2], kbits[2];
    uint32_t N, K;
    const uint64_t *reads;
    HashSlot *slots[2];
    uint64_t cap[2];
    const uint32_t *ids[2];
    int firstmax;                    // width cap of the first two probe batches of a step when many chains are in flight (the first batch: twice the running mean of the
                                     // priority index of the chain's hits + 16 probes, rounded up to 16)
    const uint32_t *bloom[2]; uint32_t bloom_lines; int bloom_nwin[2]; uint32_t bloom_mmask;   // bitmap over the keys of each dictionary (bloom_pos), 0 lines = none
    const uint2 *largetab;           // bins of more than HARC_LARGEBIN reads (SLOT_BIG; their slot's `start` indexes this table): x = first index into ids[], y = first row of `mirror`
    uint64_t *mirror;                // the reads of those bins once more, W words per entry, in bin order: their scan is one coalesced stream
    unsigned long long *claimed;     // bitmap, bit rid&63 of word rid>>6
    uint32_t *bid;                   // per read: smallest (step<<20 | chain) bidding for it this super-round
    ChainHdr *hdr;
    uint4 *cnt;                      // [2][K][Lp] column counts (A,C,G,T) -- reorder.cpp:467 `count`
    uint2 *steps;                    // [K][64] steps of the current super-round: {rid, shift | dir<<8}
    uint8_t *need;                   // per chain: wants a new seed (set by k_resolve, consumed by k_reseed)
    uint32_t *seedbuf;               // [K * (1 + HARC_NSUGG)] seeds, then look-ahead seeds, found by k_reseed, by rank
    uint32_t *needrank;              // [K] rank of a chain among those that want a seed (k_reseed -> the next k_steps, which applies the seed)
    uint32_t *rmeta;                 // [4] k_reseed's result: chains that wanted a seed, seeds found, look-ahead seeds found
    uint32_t own_mod, own_rem;       // design (R), chains partitioned over the ranks of a multi-GPU run: this rank walks the chains c with (c >> 2) % own_mod == own_rem (own_mod <= 1: all)
    uint32_t reseed_stress;          // tests (HARC_AMD_RESEED_STRESS): the odd workgroups of k_reseed_mg sleep this many times after every meeting; changes nothing but the timing
    uint32_t reseed_win;             // words of the claim bitmap one pass of k_reseed_mg looks at: RESEED_G * RESEED_NT (tests, HARC_AMD_RESEED_WIN: fewer, so that small inputs take several passes; same seeds)
    unsigned int *reseed_g;          // k_reseed_mg: meeting counter, flag, per-workgroup counts (k_resolve zeroes the first two words every round)
    uint2 *cst2;                     // per chain: x seeds taken (= unmatched reads, reorder.cpp:701), y lost bids
    uint32_t *bo;                    // back-off (null: off): per chain, bits 0-7 super-rounds it still sits out, bits 8-15 its walks cut in a row
    uint32_t *sugg;                  // [K][HARC_NSUGG] look-ahead seeds of every chain, highest id first
    int S;                           // speculative steps per super-round (1..64)
    int nsugg_per_seed;              // look-ahead seeds per reseed (HARC_NSUGG; 0 disables them: experiments only, the oracle uses the same value)
    int nsugg_stride;                // entries per chain in sugg[] and per seed in seedbuf[] (>= nsugg_per_seed)
    uint2 *slog;                     // [N] indexed by read id: {chain, index in the chain's singleton stream} of the reads that end as singletons (0xFF..: not one)
    // the main stream of a chain, in PAGES of 64 records {read id, pos | flag<<8 | rc<<9}: k_resolve appends the records a chain keeps in a super-round
    // side by side (up to 2 S records, one or two lines) instead of one 16-byte record per read at log[read id]; pages come from one counter, and
    // k_pages_out copies page after page to the chain's place in the output, coalesced on both sides
    uint2 *pg_rec;                   // [pages][64]
    uint2 *pg_hdr;                   // per page: {chain, number of the page inside the chain}
    uint32_t *pg_cur;                // per chain: the first page of its newest chunk (pages c * PG_CHUNK .. are chain c's first)
    unsigned int *pg_count;          // pages handed out (PG_CHUNK at a time)
    long long *cursor;               // reorder.cpp `remainingpos`, one for all chains
    unsigned long long *coopcnt;     // [HARC_COOPCNT] walks handed to the cooperative kernel so far (the host picks its workgroup size from them)
    unsigned long long *stats;       // [0] unmatched [1] conflicts [2] active chains [3] probes [4] candidates
    const uint16_t *probe_tab;       // the probes of one chain step in priority order: shift | dir<<8 | dict<<9
    const uint32_t *lds_tab;         // mask rows + probe descriptors as k_steps wants them in LDS (k_steps_tables)
    int nprobe;
    int budget;                      // HARC_SCAN_BUDGET (experiments may override it: HARC_AMD_BUDGET)
    int stepcap;                     // HARC_STEP_CAP (HARC_AMD_STEPCAP)
    int weedmin;                     // wave-uniform scan: single-read bins a batch must find before their claim bits are looked up ahead of the tests (HARC_AMD_WEEDMIN)
    int lazy;                        // 1: steps that agree with the consensus everywhere take the rows from their read and leave the counts to cons_flush (HARC_AMD_LAZY=0: every step applies its counts; same bytes)
    uint4 *cstat_coop;               // ... the cooperative kernel's own (k_chain_counts adds both up and keeps its share apart: bench.py prices the two kernels against different ceilings)
    uint2 *csteps;                   // per chain: steps walked by the launches of the main kernel (x) and of the cooperative kernel (y), kept or not
    uint4 *cstat;                    // per chain statistics: x slots inspected, y candidates tested, z sequential-equivalent key lookups, w sequential-equivalent candidates
    const uint2 *succ;               // runs with one chain or a few (k_succ): per (read, orientation) the first HARC_SUCC_N candidates of the step whose consensus IS that read; null = none
    unsigned long long *dbg;         // HARC_TIMING builds only: per-phase cycle sums of k_steps
    unsigned int *grp_ticket;        // k_steps_grp: the next chain a group takes when its walk is over (k_resolve sets it back to zero)
    uint32_t grp_wide_warn, grp_wide_limit;   // k_steps_grp (steps_group.h): largest column count the u16 form reports / walks itself (GRP_WIDE_WARN / GRP_WIDE_LIMIT; tests lower them)
};
#define PG_CHUNK 8u
#define HARC_SUCC_N 8              // entries of a successor list: one 64-byte line per (read, orientation)
#define HARC_SUCC_OPEN 0x80000000u // in every entry's y: the list stops before the probes of the step do (a large bin, or more candidates than entries)
#ifndef HARC_SEQ_EXTRA_WAVES
#define HARC_SEQ_EXTRA_WAVES 3     // waves per SIMD of the dense wave-uniform kernel above the base of 5: 8 (64 vector, 80 scalar registers)
#endif
#ifndef HARC_W0_MUL
#define HARC_W0_MUL 2
#endif
#ifndef HARC_W0_ADD
#define HARC_W0_ADD 16
#endif
#ifdef HARC_TIMING
#define TICK(k) do { const long long tn_ = clock64(); tacc[k] += (unsigned long long)(tn_ - tlast); tlast = tn_; } while (0)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(v, o, 64); v = x > v ? x : v; } return v; }
#else
#define TICK(k) do { } while (0)
#endif
enum { ST_UNMATCHED = 0, ST_CONFLICTS = 1, ST_ACTIVE = 2, ST_PROBES = 3, ST_CANDS = 4, ST_USEFUL = 5, ST_CANDS_SEQ = 6, ST_WIDE = 7 /* k_steps_grp: some chain's counts passed GRP_WIDE_WARN (sticky) */, /* 8: ST_WIDE + 1 */
       ST_COOP_USEFUL = 9, ST_COOP_CANDS_SEQ = 10, ST_COOP_CANDS = 11 /* the cooperative kernel's share of ST_USEFUL / ST_CANDS_SEQ / ST_CANDS */, ST_COOP_STEPS = 12, ST_DENSE_STEPS = 13 /* steps WALKED (kept or not) by the launches of either kernel */,
       ST_N = 16 /* the last word: entries of the singleton log */ };
#define HARC_COOPCNT 64   // counters of the walks handed to the cooperative kernel, spread over as many words (one word serialised 1300 atomics per round: +10 us)

// ------------------------------------------------------------------------------------------------ packing kernels
// ASCII -> std::bitset<2L> words (reorder.cpp:184-209). One thread per (read, word).
__global__ void k_pack2(const char *ascii, uint32_t n, uint32_t stride, int L, int W, uint64_t *out)
{
    const size_t gid = harc_gid();
    if (gid >= (size_t)n * W) return;
    const uint32_t i = (uint32_t)(gid / W); const int w = (int)(gid % W);
    const char *s = ascii + (size_t)i * stride;
    uint64_t v = 0;
    for (int k = 0; k < 32; k++) {
        const int b = 32 * w + k;
        if (b < L) {
            const char ch = s[b];
            const uint64_t pc = ch == 'A' ? 0 : ch == 'G' ? 1 : ch == 'C' ? 2 : 3;
            v |= pc << (2 * k);
        }
    }
    out[gid] = v;
}
// ASCII -> std::bitset<3L> words (encoder.cpp:729-749): A=0 N=1 G=2 C=4 T=6 at bits 3i.. One thread per (read, word).
__global__ void k_pack3(const char *ascii, uint32_t n, uint32_t stride, int L, int W3, uint64_t *out)
{
    const size_t gid = harc_gid();
    if (gid >= (size_t)n * W3) return;
    const uint32_t i = (uint32_t)(gid / W3); const int w = (int)(gid % W3);
    const char *s = ascii + (size_t)i * stride;
    uint64_t v = 0;
    const int b0 = (64 * w) / 3, b1 = (64 * w + 63) / 3;
    for (int b = b0; b <= b1 && b < L; b++) {
        const char ch = s[b];
        const uint64_t c3 = ch == 'A' ? 0 : ch == 'N' ? 1 : ch == 'G' ? 2 : ch == 'C' ? 4 : 6;
        const int sh = 3 * b - 64 * w;
        v |= sh >= 0 ? (c3 << sh) : (c3 >> (-sh));
    }
    out[gid] = v;
}
// 2-bit words -> text lines of L+1 bytes (reorder.cpp:832-846 bitsettostring). One thread per (read, word).
__global__ void k_unpack2(const uint64_t *reads, uint32_t n, int L, int W, char *out)
{
    const size_t gid = harc_gid();
    if (gid >= (size_t)n * W) return;
    const uint32_t i = (uint32_t)(gid / W); const int w = (int)(gid % W);
    uint64_t v = reads[gid];
    char *s = out + (size_t)i * (L + 1);
    for (int k = 0; k < 32; k++) {
        const int b = 32 * w + k;
        if (b < L) { const int pc = (int)(v & 3); s[b] = pc == 0 ? 'A' : pc == 1 ? 'G' : pc == 2 ? 'C' : 'T'; v >>= 2; }
    }
    if (w == W - 1) s[L] = '\n';
}

int s1_pack_ascii(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *d_out)
{
    if (!n) return HARC_AMD_OK;
    const size_t tot = (size_t)n * c->W;
    hipLaunchKernelGGL(k_pack2, harc_grid256(tot), dim3(256), 0, c->stream, d_ascii, n, stride, c->P.readlen, c->W, d_out);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}
int s1_pack3_ascii(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *d_out)
{
    if (!n) return HARC_AMD_OK;
    const size_t tot = (size_t)n * c->W3;
    hipLaunchKernelGGL(k_pack3, harc_grid256(tot), dim3(256), 0, c->stream, d_ascii, n, stride, c->P.readlen, c->W3, d_out);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}
int s1_unpack_to_ascii(harc_amd_ctx *c, const uint64_t *d_reads, uint32_t n, char *d_out)
{
    if (!n) return HARC_AMD_OK;
    const size_t tot = (size_t)n * c->W;
    hipLaunchKernelGGL(k_unpack2, harc_grid256(tot), dim3(256), 0, c->stream, d_reads, n, c->P.readlen, c->W, d_out);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}

// multi-GPU shard key: hash of the canonical minimizer (k=15) of the whole read, modulo the number of GPUs.  One thread per read.
__global__ void k_bucket(const uint64_t *reads, uint32_t n, int L, int W, uint32_t nb, uint32_t *out)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t *r = reads + (size_t)i * W;
    const int K = L < 15 ? L : 15;
    const uint64_t kmask = (K < 32) ? (((uint64_t)1 << (2 * K)) - 1) : ~(uint64_t)0;
    uint64_t fw = 0, rv = 0, best = ~(uint64_t)0;
    for (int b = 0; b < L; b++) {
        const uint64_t pc = (r[b >> 5] >> (2 * (b & 31))) & 3;
        fw = ((fw << 2) | pc) & kmask;
        rv = (rv >> 2) | ((3 - pc) << (2 * (K - 1)));
        if (b >= K - 1) { const uint64_t h = mix64(fw < rv ? fw : rv); best = h < best ? h : best; }
    }
    out[i] = (uint32_t)(best % nb);
}
int s1_bucket_reads(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint32_t *d_out)
{
    if (!n) return HARC_AMD_OK;
    hipLaunchKernelGGL(k_bucket, harc_grid256(n), dim3(256), 0, c->stream, d_packed, n, c->P.readlen, c->W, nb, d_out);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}

// stable partition of the packed reads by bucket (the send buffer of the all-to-all): bucket -> one radix pass on (bucket, index)
// -> gather; counts[b] = reads of bucket b
__global__ void k_bucket_keys(const uint32_t *bucket, uint32_t n, uint64_t *keys, uint32_t *idx, unsigned long long *counts)
{
    const uint32_t i = harc_gid32();
    const bool in = i < n;
    const uint32_t b = in ? bucket[i] : 0xFFFFFFFFu;
    if (in) { keys[i] = b; idx[i] = i; }
    // one atomic per distinct bucket per wave
    unsigned long long todo = __ballot(in);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lb = __shfl(b, leader, 64);
        const unsigned long long same = __ballot(in && b == lb);
        if ((threadIdx.x & 63) == leader) atomicAdd(&counts[lb], (unsigned long long)__popcll(same));
        todo &= ~same;
    }
}
__global__ void k_gather_reads(const uint64_t *reads, const uint32_t *idx, uint32_t n, int W, uint64_t *out)
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)n * W) return;
    const uint32_t i = (uint32_t)(gid / W); const int w = (int)(gid % W);
    out[gid] = reads[(size_t)idx[i] * W + w];
}
int s1_partition_reads(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint64_t *d_out, unsigned long long *d_counts)
{
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nb * 8, c->stream));
    if (!n) { HIP_TRY(hipStreamSynchronize(c->stream)); return HARC_AMD_OK; }
    PoolScope scope(c);
    uint32_t *b = nullptr, *i0 = nullptr, *i1 = nullptr; uint64_t *k0 = nullptr, *k1 = nullptr;
    RC_TRY(dalloc(c, &b, n)); RC_TRY(dalloc(c, &i0, n)); RC_TRY(dalloc(c, &i1, n)); RC_TRY(dalloc(c, &k0, n)); RC_TRY(dalloc(c, &k1, n));
    hipLaunchKernelGGL(k_bucket, harc_grid256(n), dim3(256), 0, c->stream, d_packed, n, c->P.readlen, c->W, nb, b);
    hipLaunchKernelGGL(k_bucket_keys, harc_grid256(n), dim3(256), 0, c->stream, (const uint32_t *)b, n, k0, i0, d_counts);
    unsigned bits = 1; while ((1u << bits) < nb) bits++;
    RC_TRY(prim_sort_pairs_u64_u32(c, k0, k1, i0, i1, n, bits));                 // stable: original order inside a bucket
    hipLaunchKernelGGL(k_gather_reads, harc_grid256((uint64_t)n * c->W), dim3(256), 0, c->stream, d_packed, (const uint32_t *)i1, n, c->W, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ index build
// key_l(read) = bases [ds_l, de_l] of the read (reorder.cpp:295-299)
// both dictionaries' keys in ONE pass over the reads (the second pass was 11 GB read again at configs[2]: 2.7 ms)
template <int W> __global__ void k_keygen2(const uint64_t *reads, uint32_t n, int off0, int nbits0, int off1, int nbits1, uint64_t *keys0, uint64_t *keys1, uint32_t *ids)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    uint64_t r[W];
#pragma unroll
    for (int w = 0; w < W; w++) r[w] = reads[(size_t)i * W + w];
    {
        const int wi = off0 >> 6, sh = off0 & 63;
        const uint64_t lo = sel0<W>(r, wi), hi = sel0<W>(r, wi + 1);
        uint64_t v = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        if (nbits0 < 64) v &= ((uint64_t)1 << nbits0) - 1;
        keys0[i] = v;
    }
    {
        const int wi = off1 >> 6, sh = off1 & 63;
        const uint64_t lo = sel0<W>(r, wi), hi = sel0<W>(r, wi + 1);
        uint64_t v = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        if (nbits1 < 64) v &= ((uint64_t)1 << nbits1) - 1;
        keys1[i] = v;
    }
    ids[i] = i;
}
// Bins of more than HARC_LARGEBIN reads (repeats, low-complexity sequence) get their reads copied once more in bin order, so that the
// cooperative scan of k_steps streams 64 candidates per round trip instead of gathering them ("LDS-staged reference reads, coalesced
// Hamming scan" of the north star, for the bins where it pays).  large_list was filled by k_table_insert: (slot index << 1) | dictionary.
__global__ void k_large_sizes(const unsigned long long *list, uint32_t nlist, HashSlot *s0, HashSlot *s1, uint32_t *sz)
{
    const uint32_t b = harc_gid32();
    if (b >= nlist) return;
    const unsigned long long e = list[b];
    const HashSlot *slot = ((e & 1) ? s1 : s0) + (e >> 1);
    sz[b] = slot->count & SLOT_CNT_MASK;
}
template <int W> __global__ __launch_bounds__(64) void k_large_fill(const unsigned long long *list, uint32_t nlist, HashSlot *s0, HashSlot *s1, const uint32_t *ids0, const uint32_t *ids1,
                                                                   const uint64_t *moff, const uint64_t *reads, uint2 *largetab, uint64_t *mirror)
{
    const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;      // (rows of at most 2^25 bins: wave_grid)
    if (b >= nlist) return;
    const int lane = threadIdx.x;
    const unsigned long long e = list[b];
    const int l = (int)(e & 1);
    HashSlot *slot = (l ? s1 : s0) + (e >> 1);
    const uint32_t st = slot->start, cnt = slot->count & SLOT_CNT_MASK;
    const uint32_t *ids = (l ? ids1 : ids0) + st;
    const uint64_t m0 = moff[b];
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t rid = ids[k];
#pragma unroll
        for (int w = 0; w < W; w++) mirror[(m0 + k) * W + w] = reads[(size_t)rid * W + w];
    }
    if (lane == 0) { largetab[b] = make_uint2(st, (uint32_t)m0); slot->start = b; }       // from now on the slot names its row of largetab
}
__global__ void k_mark_heads(const uint64_t *skeys, uint32_t n, uint32_t *head)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1u : 0u;
}
// ... and, with the keys at hand, what the placement's max-scan starts from: 4 b_i - i of bin i (see below), biased by n to stay unsigned
__global__ void k_bin_starts(const uint32_t *head, const uint32_t *binidx, uint32_t n, uint32_t *binstart, uint32_t *nbins, const uint64_t *skeys, uint64_t cap, uint64_t *v)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    if (head[i]) { const uint32_t b = binidx[i]; binstart[b] = i; v[b] = bucket_slot(skeys[i], cap) + (uint64_t)n - (uint64_t)b; }
    if (i == n - 1) *nbins = binidx[i] + head[i];
}
#define HARC_LARGEBIN 16u    // stage-I bins with more reads than this are compacted between super-rounds (k_compact_bins)
#define HARC_STEP_CAP 12    // schedule: a STEP that has made this many probes into such bins without a hit is put off -- the walk ends in front of it, and the next
                             // super-round takes the step up again behind the probes already made (they found nothing against fewer claims; ChainHdr.flags >> 16)
#define HARC_BO_FREE 1u      // schedule (repeat-rich input with more than 16 384 chains only; oracle: BO_FREE / BO_CAP): in the end phase of such an input every chain walks towards the
#define HARC_BO_CAP 3u       // same few reads and nine walks in ten are cut (profiles/r05/phantom_bids.txt); a chain whose walks keep being cut sits out 0, 1, 3, 7, 7 ... super-rounds
#define HARC_SCAN_BUDGET 12  // schedule: a walk ends after the step in which the probes it made into such bins (still holding unclaimed reads) reach this number
// The table is probed bucket by bucket (64 B = 4 slots); a search that finds a full bucket WITHOUT the overflow flag can stop.
// The reads arrive sorted by scrambled key, so the bins arrive in bucket order (bucket_slot is monotone): the slot of bin i is
// max(4 * bucket_i, slot_{i-1} + 1) -- the linear-probing invariant -- i.e. an inclusive max-scan of (4 * bucket_i - i), plus i.
// No atomics, and the stores walk the table front to back.  (A CAS insert ran at the random read-modify-write rate: 37 ms for 350 M bins.)
// The sort of the scrambled keys looks at their TOP bits only (40 of 64 for up to 2^28 reads: five radix passes instead of eight -- the sort is
// a sixth of the index build); two different keys that agree in those bits are rare (n^2 / 2^41 pairs: 55 000 among 350 M keys) and end up
// side by side, possibly interleaved.  k_mixed_find lists the places where a key follows a DIFFERENT key with the same top bits; k_mixed_fix
// sorts every such stretch by the whole key (stable: ids stay ascending inside a bin), one thread per stretch, started from its first listed
// place.  A stretch that would take too long (two large bins that collide), or a list that overflows, raises a flag and the host sorts again on
// all 64 bits.
#define MIXED_MAX (1u << 20)
#define MIXED_BUDGET 200000u
// (the keys go through the sort ROTATED, their top bits at the bottom: rocprim sorts the low `sbits` bits, begin_bit = 0 -- with begin_bit > 0
// its small-input path returned a wrong order here; this pass turns them back on the way)
__device__ __forceinline__ uint64_t rotl64(uint64_t x, unsigned r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }
__global__ void k_mixed_find(const uint64_t *rk, uint32_t n, unsigned sbits, uint64_t *sk, uint32_t *list, unsigned int *meta)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const int shift = (int)(64u - sbits);
    const uint64_t b = rotl64(rk[i], 64u - sbits);                // rotate right by sbits
    sk[i] = b;
    if (i == 0) return;
    const uint64_t a = rotl64(rk[i - 1], 64u - sbits);
    if (a != b && (a >> shift) == (b >> shift)) { const unsigned int at = atomicAdd(&meta[0], 1u); if (at < MIXED_MAX) list[at] = i; else meta[1] = 1u; }
}
// which listed place owns its stretch (the first one inside it); the others are struck off before anything moves
__global__ void k_mixed_own(const uint64_t *sk, int shift, uint32_t *list, unsigned int *meta)
{
    const uint32_t t = harc_gid32();
    const unsigned int nl = meta[0] < MIXED_MAX ? meta[0] : MIXED_MAX;
    if (t >= nl || meta[1]) return;
    const uint32_t i = list[t];
    const uint64_t top = sk[i] >> shift;
    for (uint32_t h = i; h > 0 && (sk[h - 1] >> shift) == top;) {
        h--;
        if (h > 0 && sk[h] != sk[h - 1] && (sk[h - 1] >> shift) == top) { list[t] = HARC_NONE; return; }      // an earlier listed place
        if (i - h > MIXED_BUDGET) { meta[1] = 1u; return; }
    }
}
__global__ void k_mixed_fix(uint64_t *sk, uint32_t *ids, uint32_t n, int shift, const uint32_t *list, unsigned int *meta)
{
    const uint32_t t = harc_gid32();
    const unsigned int nl = meta[0] < MIXED_MAX ? meta[0] : MIXED_MAX;
    if (t >= nl || meta[1]) return;
    const uint32_t i = list[t];
    if (i == HARC_NONE) return;
    const uint64_t top = sk[i] >> shift;
    uint32_t h = i;
    while (h > 0 && (sk[h - 1] >> shift) == top) h--;
    uint32_t e = i + 1;
    while (e < n && (sk[e] >> shift) == top) { e++; if (e - h > MIXED_BUDGET) { meta[1] = 1u; return; } }
    uint32_t moves = 0;
    for (uint32_t x = h + 1; x < e; x++) {                        // insertion sort: linear in the stretch plus its inversions
        const uint64_t kx = sk[x]; const uint32_t vx = ids[x];
        uint32_t y = x;
        while (y > h && sk[y - 1] > kx) { sk[y] = sk[y - 1]; ids[y] = ids[y - 1]; y--; if (++moves > MIXED_BUDGET) { meta[1] = 1u; return; } }
        if (y != x) { sk[y] = kx; ids[y] = vx; }
    }
}
__global__ void k_scramble_keys(uint64_t *keys, uint32_t n, unsigned rot)
{
    const uint32_t i = harc_gid32();
    if (i < n) keys[i] = rotl64(key_scramble(keys[i]), rot);
}
__global__ void k_rotate_keys(uint64_t *keys, uint32_t n, unsigned rot)
{
    const uint32_t i = harc_gid32();
    if (i < n) keys[i] = rotl64(keys[i], rot);
}
// pass 0: the bins whose slot lies inside the table, plain 16-byte stores, overflow flags included (round 2 set them in a second pass over all
// bins: 2 x 2.6 ms per dictionary at configs[2]).  pass 1 (after pass 0 has finished, over the last 16 384 bins only): the few bins at the very
// end whose slot falls past the table: they wrap around like a probe would.  Single-read bins (the common case) carry the read id in
// `start`: one dependent load less on every hit.
#ifndef TP_SPAN
#define TP_SPAN 1280u             // table slots a workgroup of the filling placement stages in LDS: 256 bins span 1024 +- 64 slots at load factor 1/4.  (2048 slots = 32 KB held
                                  // the kernel at 5 workgroups per CU, and a workgroup is a chain of five dependent loads before its first store: 10.75 ms per dictionary at
                                  // configs[2] for 22 GB, 2.1 TB/s; 20 KB lets the CU's 8 workgroups in: 6.4 ms.  A longer stretch -- 4 sigma -- takes the untiled path)
#endif
__global__ __launch_bounds__(256) void k_table_place(const uint64_t *skeys, const uint32_t *sids, const uint32_t *binstart, uint32_t nbins, uint32_t n, const uint64_t *q,
                              HashSlot *slots, uint64_t cap, uint32_t bigthresh,
                              unsigned long long *large_list, unsigned int *large_n, uint32_t large_max, uint32_t large_tag, uint32_t *nbins_p, int pass, uint32_t first_block, int fill)
{
    __shared__ uint4 tile[TP_SPAN];
    if (blockIdx.x < first_block) return;                                          // pass 1: the bins beyond the end of the table are among the last
    const uint32_t i = harc_gid32();
    const bool have = i < nbins;
    auto slot_of = [&](uint32_t k) -> uint64_t { return q[k] - (uint64_t)n + (uint64_t)k; };     // max-scan value (biased by n, k_bin_starts) + k
    // fill (pass 0): the table has NOT been cleared (22 GB per dictionary at configs[2]): slots grow with the bin index, so the 256 bins of a
    // workgroup own one contiguous stretch of the table -- from behind the last slot of the workgroup before to their own last slot, to the end
    // of the table for the last bins inside it.  The stretch is put together in LDS (zeros, then the bins' slots) and written out as ONE
    // coalesced stream of 16-byte stores: every slot of the table is written exactly once (round 3 let every bin write the gap in front of it
    // with stores of its own: 12.3 ms per dictionary at configs[2] for 22 GB; now 10.5).  A stretch longer than the LDS tile
    // (few bins, long gaps) falls back to that.
    const uint32_t b0 = blockIdx.x * blockDim.x, b1 = b0 + blockDim.x < nbins ? b0 + blockDim.x : nbins;
    bool tiled = false; uint64_t lo = 0, hi = 0;
    if (pass == 0 && fill && b0 < nbins) {
        lo = b0 ? slot_of(b0 - 1) + 1 : 0;
        const uint64_t last = slot_of(b1 - 1);
        if (lo < cap) {
            hi = (last >= cap || b1 == nbins || slot_of(b1) >= cap) ? cap : last + 1;     // the last bins inside the table clear what lies behind them
            tiled = hi - lo <= TP_SPAN;
        }
    }
    if (tiled) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t j = threadIdx.x; j < (uint32_t)(hi - lo); j += blockDim.x) tile[j] = z;
        __syncthreads();
    }
    uint64_t sl = have ? slot_of(i) : 0;
    bool mine = have;
    // pass 0 places the bins whose slot lies inside the table; pass 1 (launched over the last bins only) the few beyond its end
    if (mine && pass == 0 && sl >= cap && blockIdx.x + 64 < gridDim.x) { atomicAdd(nbins_p + 1, 1u); mine = false; }   // pass 1 would not reach it: the build fails loudly (never seen)
    if (mine && (pass == 1) != (sl >= cap)) mine = false;
    if (mine) {
        const uint32_t st = binstart[i];
        const uint64_t key = skeys[st];
        // the overflow flag of a bucket lives in its first slot: "a key whose home is this bucket or an earlier one sits beyond it".  Slots and
        // homes both grow with the bin index, so that is the case exactly when the bucket is full and the bin right behind it has its home
        // here or earlier -- decided by the thread that writes the first slot, without a second pass over all bins
        uint32_t ovf = 0;
        if (pass == 0 && (sl & 3) == 0 && (uint64_t)i + 4 < (uint64_t)nbins) {
            const uint64_t s3 = slot_of(i + 3);
            if (s3 == sl + 3 && (bucket_slot(skeys[binstart[i + 4]], cap) >> 2) <= (sl >> 2)) ovf = SLOT_OVF;
        }
        if (pass == 0 && fill && !tiled) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (uint64_t x = i ? slot_of(i - 1) + 1 : 0; x < sl; x++) *reinterpret_cast<uint4 *>(&slots[x]) = z;
            if (i + 1 == nbins || slot_of(i + 1) >= cap)
                for (uint64_t x = sl + 1; x < cap; x++) *reinterpret_cast<uint4 *>(&slots[x]) = z;
        }
        const uint32_t en = (i + 1 < nbins) ? binstart[i + 1] : n;
        const uint32_t cnt = en - st;
        if (cnt > SLOT_CNT_MASK) atomicAdd(nbins_p + 1, 1u);                       // does not fit the count field: the build fails loudly
        else {
            // Single-read bins (the common case) carry the read id in `start`: one dependent load less on every hit.
            const unsigned long long meta = cnt == 1 ? ((unsigned long long)sids[st] | ((unsigned long long)(1u | SLOT_EMB) << 32))
                                                     : ((unsigned long long)st | ((unsigned long long)((cnt & SLOT_CNT_MASK) | ((bigthresh && cnt > bigthresh) ? SLOT_BIG : 0u)) << 32));
            if (pass == 1) {
                // a bin beyond the end wraps around like a probe would: it has left its home bucket, every bucket up to the last, and whatever it passes at the start
                for (uint64_t bb = bucket_slot(key, cap) >> 2; bb < (cap >> 2); bb++) atomicOr(&slots[4 * bb].count, SLOT_OVF);
                sl = 0;
                for (;;) {
                    unsigned long long *mp = reinterpret_cast<unsigned long long *>(&slots[sl]) + 1;
                    if (atomicCAS(mp, 0ULL, meta) == 0ULL) break;
                    if ((sl & 3) == 3) atomicOr(&slots[sl - 3].count, SLOT_OVF);
                    if (++sl == cap) sl = 0;
                }
                slots[sl].key = key;
            } else {
                uint4 w; w.x = (uint32_t)key; w.y = (uint32_t)(key >> 32); w.z = (uint32_t)meta; w.w = (uint32_t)(meta >> 32) | ovf;
                if (tiled) tile[sl - lo] = w; else *reinterpret_cast<uint4 *>(&slots[sl]) = w;
            }
            if (large_list && cnt > HARC_LARGEBIN) {                              // remembered for k_compact_bins: (slot index, dictionary)
                const unsigned int at = atomicAdd(large_n, 1u);
                if (at < large_max) large_list[at] = ((unsigned long long)sl << 1) | large_tag;
            }
        }
    }
    if (tiled) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < (uint32_t)(hi - lo); j += blockDim.x) *reinterpret_cast<uint4 *>(&slots[lo + j]) = tile[j];
    }
}

// The same bitmap for a LARGE input, without the random atomics (configs[2]: 350 M of them into 700 MB, 13.6 ms per dictionary at the
// random-access ceiling): every key becomes (tile of the bitmap, word inside the tile, its two bits) in one u64, the u64s are sorted by tile
// (two radix passes over the tile bits), and one workgroup per tile ORs its keys into 64 KB of LDS and writes the tile out in one coalesced
// stream -- every byte of the bitmap is written exactly once, so it needs no clearing either.
__global__ void k_s1_bloom_keys(const uint64_t *keys, uint32_t n, uint32_t nlines, int nwin, uint32_t mmask, uint64_t *pk)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    uint32_t w, m;
    bloom_pos(keys[i], key_scramble(keys[i]), nlines, nwin, mmask, &w, &m);
    const uint32_t b0 = (uint32_t)__ffs((int)m) - 1u, b1 = 31u - (uint32_t)__clz((int)m);
    pk[i] = harc_bitmap_item(w, b0, b1);
}
__global__ __launch_bounds__(256) void k_s1_bloom_tile(const uint64_t *pk, uint32_t n, uint64_t nwords, uint32_t *bloom)
{
    __shared__ uint32_t tile[BL_TILE_WORDS];
    __shared__ uint32_t bounds[2];
    const uint32_t part = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < BL_TILE_WORDS; j += 256) tile[j] = 0u;
    if (threadIdx.x < 2) {                                        // first key of this tile and of the next one
        const uint64_t want = (uint64_t)part + threadIdx.x, pmask = ((uint64_t)1 << BL_PART_BITS) - 1;
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if ((pk[mid] & pmask) < want) lo = mid + 1; else hi = mid; }
        bounds[threadIdx.x] = lo;
    }
    __syncthreads();
    for (uint32_t i = bounds[0] + threadIdx.x; i < bounds[1]; i += 256) {
        const uint32_t v = (uint32_t)(pk[i] >> BL_PART_BITS);
        atomicOr(&tile[v >> 10], (1u << ((v >> 5) & 31u)) | (1u << (v & 31u)));
    }
    __syncthreads();
    const uint64_t base = (uint64_t)part * BL_TILE_WORDS;
    for (uint32_t j = threadIdx.x; j < BL_TILE_WORDS; j += 256) if (base + j < nwords) bloom[base + j] = tile[j];
}
// items[n] (harc_bitmap_item: tile, word in the tile, two bit positions; an item whose tile is harc_bitmap_tiles(nwords) sets nothing) -> the
// bitmap of `nwords` 32-bit words, every word written once (no clearing needed).  `tmp` holds n u64 as well.  Shared with stage II's combined bitmap.
int harc_bitmap_from_items(harc_amd_ctx *c, const uint64_t *items, uint64_t *tmp, size_t n, uint64_t nwords, uint32_t *bitmap)
{
    const uint64_t ntiles64 = (nwords + BL_TILE_WORDS - 1) / BL_TILE_WORDS;
    if (ntiles64 + 1 >= (1ull << BL_PART_BITS) || n > 0xFFFFFFFFull) { harc_set_error("bitmap of %llu words / %zu items: too large for the tiled build", (unsigned long long)nwords, n); return HARC_AMD_EINVAL; }
    const uint32_t ntiles = (uint32_t)ntiles64;
    if (ntiles == 0) return HARC_AMD_OK;
    unsigned tb = 1; while (((uint64_t)1 << tb) <= ntiles) tb++;  // the tile numbers 0 .. ntiles (the last one: items that set nothing)
    RC_TRY(prim_sort_keys_u64(c, items, tmp, n, tb));
    hipLaunchKernelGGL(k_s1_bloom_tile, dim3(ntiles), dim3(256), 0, c->stream, (const uint64_t *)tmp, (uint32_t)n, nwords, bitmap);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}
__global__ void k_s1_bloom_diff(const uint32_t *a, const uint32_t *b, uint64_t nwords, unsigned long long *ndiff)
{
    const uint64_t i = harc_gid();
    const bool d = i < nwords && a[i] != b[i];
    const unsigned long long m = __ballot(d);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(ndiff, (unsigned long long)__popcll(m));
}
__global__ void k_s1_bloom_set(const uint64_t *keys, uint32_t n, uint32_t *bloom, uint32_t nlines, int nwin, uint32_t mmask)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    uint32_t w, m;
    bloom_pos(keys[i], key_scramble(keys[i]), nlines, nwin, mmask, &w, &m);
    if ((bloom[w] & m) != m) atomicOr(&bloom[w], m);
}

// ------------------------------------------------------------------------------------------------ chain kernels
// Schedule (the oracle's stage1_run restates it on the CPU): K chains, super-rounds of S speculative steps.
//   k_steps   (A) every live chain -- one 64-lane wave -- walks up to S steps against the FROZEN claim bitmap with its
//                 consensus counts held in registers, bidding (step<<20 | chain) for every read it takes;
//   k_resolve (B) a read goes to the smallest bid; a chain keeps the steps before its first lost bid and is rolled back to there;
//   k_reseed  (C) chains that ran out of candidates take new seeds, in chain order, from ONE descending cursor (reorder.cpp:652-668).
__global__ void k_init_chains(S1Args s)
{
    const uint32_t c = harc_gid32();
    if (c >= s.K) return;
    ChainHdr h; memset(&h, 0, sizeof h);
    const uint32_t step = s.N / s.K;                             // reorder.cpp:490
    const bool act = s.N > 0 && (c == 0 || step > 0);            // a seed that is already taken makes the chain give up (:484)
    if (act) {
        const uint32_t seed = c * step;
        atomicOr(&s.claimed[seed >> 6], 1ULL << (seed & 63));
        h.cur = seed; h.prev = seed; h.flags = CH_ACTIVE | CH_PREVUNM; h.mode = 2;
    }
    s.hdr[c] = h;
    s.cst2[c] = make_uint2(act ? 1u : 0u, 0u);
    s.need[c] = 0;
    s.pg_cur[c] = c * PG_CHUNK;
    for (uint32_t k = 0; k < PG_CHUNK; k++) s.pg_hdr[(size_t)c * PG_CHUNK + k] = make_uint2(c, k);
    if (c == 0) *s.pg_count = s.K * PG_CHUNK;
}

// consensus state of one chain, spread over the wave: lane owns ring slots p = lane + 64 t (t < CT); the slot holding
// consensus column i is p = (i + base) mod LP, so a shift of the chain only moves `base` (reorder.cpp:887-908 without data movement)
template <int W, bool INLDS> struct ConsState {
    static constexpr int CT = (W + 1) / 2;
    static constexpr int LP = 64 * CT;
    // counts A,C,G,T of the slot (reorder.cpp:467 `count`).  INLDS (the dense main kernel, 7 waves per SIMD on 72 registers): in LDS, slot p
    // of the wave at ql[p - lane] (the pointer is the lane's own) -- in registers they were 8 of the 80 the dense build had, the compiler kept
    // them in scratch across every batch (436 MB of scratch writes per launch at configs[2]), and a scratch reload waits for every load
    // issued before it (vmcnt counts in order).  Otherwise (few chains, cooperative walks: latency-bound, registers to spare) in registers:
    // LDS there cost configs[1] 7 % and the repeat workloads 3-7 %.
    uint4 qr[INLDS ? 1 : CT];
    uint4 *ql;
    int v[CT];        // consensus base of the slot (first strict maximum)
    int base;
    __device__ __forceinline__ uint4 getq(int t) const { if constexpr (INLDS) return ql[64 * t]; else return qr[t]; }
    __device__ __forceinline__ void setq(int t, const uint4 &q) { if constexpr (INLDS) ql[64 * t] = q; else qr[t] = q; }
};
__device__ __forceinline__ int argmax4(const uint4 &q)
{
    uint32_t mx = 0; int ind = 0;                                // first strict maximum: ties A<C<G<T (reorder.cpp:893-899)
    if (q.x > mx) { mx = q.x; ind = 0; }
    if (q.y > mx) { mx = q.y; ind = 1; }
    if (q.z > mx) { mx = q.z; ind = 2; }
    if (q.w > mx) { mx = q.w; ind = 3; }
    return ind;
}
// base (count row A0 C1 G2 T3) of the oriented read at column i
template <int W> __device__ __forceinline__ int oriented_base(const uint64_t (&rw)[W], int L, int rev, int i)
{
    const int sc = rev ? (L - 1 - i) : i;
    const int pc = (int)((sel0<W>(rw, sc >> 5) >> (2 * (sc & 31))) & 3);
    const int b = ((pc & 1) << 1) | (pc >> 1);                   // packed code A0 G1 C2 T3 -> A0 C1 G2 T3
    return rev ? 3 - b : b;
}
template <int W, bool I> __device__ __forceinline__ void cons_reset(ConsState<W, I> &st, const uint64_t (&rw)[W], int L, int lane)
{
    st.base = 0;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        const int i = lane + 64 * t;
        int b = 0; uint4 q = make_uint4(0, 0, 0, 0);
        if (i < L) { b = oriented_base<W>(rw, L, 0, i); q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3); }
        st.setq(t, q); st.v[t] = b;
    }
}
// updaterefcount for a match at `shift` with the read oriented by `rev` (reorder.cpp:884-909)
template <int W, bool I> __device__ __forceinline__ void cons_update(ConsState<W, I> &st, const uint64_t (&rw)[W], int L, int rev, int shift, int lane)
{
    constexpr int LP = ConsState<W, I>::LP;
    int nb = st.base + shift; if (nb >= LP) nb -= LP;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        const int p = lane + 64 * t;
        int oldl = p - st.base; if (oldl < 0) oldl += LP;
        int newl = p - nb; if (newl < 0) newl += LP;
        uint4 q = st.getq(t); int v = 0;
        if (newl < L) {
            const int b = oriented_base<W>(rw, L, rev, newl);
            if (oldl < L && oldl >= shift) { q.x += (b == 0); q.y += (b == 1); q.z += (b == 2); q.w += (b == 3); v = argmax4(q); }
            else { q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3); v = b; }
        } else q = make_uint4(0, 0, 0, 0);
        st.setq(t, q); st.v[t] = v;
    }
    st.base = nb;
}
// The count matrices of a chain go to HBM between super-rounds (two halves by parity: the rollback point and the state after the walk).  A
// column is four u8 (4 bytes) while no count of the chain exceeds 255 -- the rule on ordinary data: a count is the number of the chain's reads
// over a column, i.e. about the coverage --, four u16 up to 65 535, four u32 beyond (collapsed repeats walked by a single chain): 26 instead of 105 MB
// per launch of 65 536 chains written, and as much less read by the next one.  `fmt` (0 / 1 / 2: ChainHdr.pad0 bits 24-25 for parity 0, 26-27
// for parity 1) says which form a half holds; the widest form is exact for any count, so nothing saturates.
template <int W, bool I> __device__ __forceinline__ void cons_load(ConsState<W, I> &st, const uint4 *src, int L, int lane, uint32_t fmt)
{
    st.base = 0;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        const int i = lane + 64 * t;
        uint4 q = make_uint4(0, 0, 0, 0); int v = 0;
        if (i < L) {
            if (fmt == 2u) q = src[i];
            else if (fmt == 1u) { const uint2 p = reinterpret_cast<const uint2 *>(src)[i]; q = make_uint4(p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16); }
            else { const uint32_t p = reinterpret_cast<const uint32_t *>(src)[i]; q = make_uint4(p & 0xFFu, (p >> 8) & 0xFFu, (p >> 16) & 0xFFu, p >> 24); }
            v = argmax4(q);
        }
        st.setq(t, q); st.v[t] = v;
    }
}
// returns the form that was written (wave-uniform)
template <int W, bool I> __device__ __forceinline__ uint32_t cons_store(const ConsState<W, I> &st, uint4 *dst, int L, int lane)
{
    constexpr int LP = ConsState<W, I>::LP;
    uint32_t mx = 0;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        int l = lane + 64 * t - st.base; if (l < 0) l += LP;
        if (l < L) { const uint4 q = st.getq(t); const uint32_t a = q.x > q.y ? q.x : q.y, b = q.z > q.w ? q.z : q.w; const uint32_t m = a > b ? a : b; mx = m > mx ? m : mx; }
    }
    const uint32_t fmt = __ballot(mx > 0xFFFFu) != 0 ? 2u : (__ballot(mx > 0xFFu) != 0 ? 1u : 0u);
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        int l = lane + 64 * t - st.base; if (l < 0) l += LP;
        if (l < L) {
            const uint4 q = st.getq(t);
            if (fmt == 2u) dst[l] = q;
            else if (fmt == 1u) reinterpret_cast<uint2 *>(dst)[l] = make_uint2(q.x | (q.y << 16), q.z | (q.w << 16));
            else reinterpret_cast<uint32_t *>(dst)[l] = q.x | (q.y << 8) | (q.z << 16) | (q.w << 24);
        }
    }
    return fmt;
}
// consensus -> packed 2-bit words in column order: ballot the two code bits per ring slot, then rotate the ring by `base`
template <int W, bool I> __device__ __forceinline__ void cons_pack(const ConsState<W, I> &st, int L, int lane, uint64_t (&ref)[W])
{
    constexpr int CT = ConsState<W, I>::CT, LP = ConsState<W, I>::LP, RW = 2 * CT;
    uint64_t ring[RW];
#pragma unroll
    for (int t = 0; t < CT; t++) {
        int l = lane + 64 * t - st.base; if (l < 0) l += LP;
        const bool valid = l < L;
        const int v = st.v[t];
        const unsigned long long b0 = __ballot(valid && (v >> 1)), b1 = __ballot(valid && (v & 1));   // code bit0 = row>>1, bit1 = row&1
        ring[2 * t] = spread32(b0) | (spread32(b1) << 1);
        ring[2 * t + 1] = spread32(b0 >> 32) | (spread32(b1 >> 32) << 1);
    }
    const int sb = 2 * st.base, ws = sb >> 6, bs = sb & 63;      // wave-uniform
#pragma unroll
    for (int w = 0; w < W; w++) {
        int i0 = w + ws; if (i0 >= RW) i0 -= RW;
        int i1 = i0 + 1; if (i1 >= RW) i1 -= RW;
        const uint64_t lo = sel0<RW>(ring, i0), hi = sel0<RW>(ring, i1);
        ref[w] = (bs ? ((lo >> bs) | (hi << (64 - bs))) : lo) & lowmask_word(2 * L, w);
    }
}

// The dominant kernel.  One wave per chain, 4 chains per 256-thread workgroup.  Per step:
//  (1) consensus -> packed words (ballots), reverse complement; both go to LDS, so that every lane can take the 64-bit key window and
//      the 2L-bit Hamming window of ITS shift with a few unaligned dword reads + v_alignbit instead of select chains over registers;
//  (2) the probes of the step (reorder.cpp:517-649) are dealt to the lanes in priority order, in batches: key -> bucketed open-addressing
//      table -> bin scan from the highest unclaimed id (<= maxsearch of them) -> XOR+popcount Hamming on the packed words against the
//      precomputed mask row of (direction, shift); the lowest lane with a hit is the step's read;
//  (3) lane 0 records the step and bids for the read with atomicMin(step<<20 | chain); the counts are updated in registers.
// The kernel is bound by instruction issue at large K (PMC: 85 % of the SIMD issue slots, profiles/r02) and by dependent HBM round
// trips at small K: both want few instructions per step.
// Waves per SIMD of the main kernel (launch bound -> register budget).  5 unless the launch is at least eight full rounds of waves long
// (DENSE: more than 16 384 chains, i.e. whenever the launch is not QUAD): there three more waves -- all eight a SIMD holds, 64 VGPRs -- fill issue slots that the others leave empty
// while they wait (round 3: the column counts in LDS and the chain header in scalar registers made a seventh wave pay, configs[2] 1252 ->
// 1137 us per launch; an eighth then still spilled 31 registers, 1661 us, until the wave-uniform scan of the small bins took the read's
// words out of the lanes: 1 spilled register, 1110 us), while at 24 k chains a sixth wave was 9 % slower (round 2).  Reads of more than 128 bases: one less.
#ifndef HARC_STEPS_WAVES
#define HARC_STEPS_WAVES 5
#endif
#ifndef HARC_COOP_WAVES
#define HARC_COOP_WAVES 4       // COOP kernel: register budget for 4 waves per SIMD (it sat at 127-128 registers; two more cost it a wave)
#endif
#ifndef HARC_STEPS_WAVES_Q
#define HARC_STEPS_WAVES_Q 3     // few chains (whole-bucket fetch): never more than ~3 waves / SIMD anyway
#endif
// LDS geometry of k_steps (dwords).  A packed read is NW = 2W dwords.  Window rows: [NW dwords before][NW dwords of the word][NW + 1 after]
// -- what lies outside the word is never zeroed, the mask row of (direction, shift) removes it.
template <int W> struct StepsLds {
    static constexpr int NW = 2 * W;
    static constexpr int ROW = 3 * NW + 1;
    static constexpr int MROW = (NW + 3) & ~3;                   // mask rows are 16-byte aligned
};
#ifndef HARC_SEQ_SCAN
#define HARC_SEQ_SCAN 1               // dense k_steps: the small bins of a batch are scanned one after the other by the whole wave (0: every lane its own bin, as the other kernels)
#endif
#ifndef HARC_SCAN_CH
#define HARC_SCAN_CH 1                // chunks of 256 bin entries the cooperative scan fetches per round trip (see wg_scan)
#endif
#define HARC_WGCMD_BYTES 3584          // >= sizeof(WgCmd), checked where the struct is defined
#define HARC_SK_E 4                    // wg_scan_sk: bin entries per lane and round trip (their 8-byte sketches fill the registers ONE whole read took)
#ifndef HARC_SCAN_SKETCH
#define HARC_SCAN_SKETCH 1             // the cooperative kernel in its ONE-wave form (many walks per super-round) scans large bins by sketch (wg_scan_sk); with 2 / 4 waves per walk, and with 0
                                       // (make variant) always: whole reads, 64 entries per wave and trip (wg_scan) -- measured: c2d 68.0 against 59.7 Mreads/s, c2r 50.6 / 47.4 with four waves
#endif
// the reads a chain has taken in the running super-round (they are not in the frozen claim bitmap yet), as a hash table in LDS: 128 slots per wave for
// at most 64 of them.  The wave-uniform scan weeds them out of its candidates by all lanes at once; walking the wave's register copy with
// v_readlane (one per read taken so far, every batch) was ~40 of the ~460 vector instructions of a step
#define HARC_OWN_SLOTS 128
#ifndef HARC_SEQ_OWNT
#define HARC_SEQ_OWNT 1               // the wave-uniform kernels keep the table too and their lanes weed the chain's own reads out of the single-read bins.  0 (make variant): no
                                      // table, the wave-uniform test asks the wave's registers before it fetches anything -- two LDS round trips less per step and 3 % SLOWER
                                      // (configs[2]: 1023 against 994 us per launch): at eight waves per SIMD the round trips are hidden, the two extra turns of the candidate loop are not
#endif
__device__ __forceinline__ void own_insert(uint32_t *tab, uint32_t id)
{
    uint32_t h = (id * 0x9E3779B1u) >> 25;
    while (tab[h] != HARC_NONE) h = (h + 1u) & (HARC_OWN_SLOTS - 1u);
    tab[h] = id;
}
__device__ __forceinline__ bool own_has(const uint32_t *tab, uint32_t id)
{
    uint32_t h = (id * 0x9E3779B1u) >> 25;
    for (;;) {
        const uint32_t v = tab[h];
        if (v == id) return true;
        if (v == HARC_NONE) return false;
        h = (h + 1u) & (HARC_OWN_SLOTS - 1u);
    }
}
static inline size_t steps_lds_bytes_coop(int W, int maxmatch, int nprobe)
{
    const int NW = 2 * W, ROW = 3 * NW + 1, MROW = (NW + 3) & ~3;
    return ((size_t)2 * maxmatch * MROW + (size_t)((2 * ROW + 3) & ~3) + (size_t)MROW + (size_t)8 * NW + (size_t)32 + (size_t)2 * nprobe + 8) * 4 + HARC_WGCMD_BYTES + 16;
}
static inline size_t steps_lds_bytes(int W, int maxmatch, int nprobe, bool seq = false)
{
    const int NW = 2 * W, ROW = 3 * NW + 1, MROW = (NW + 3) & ~3;
    return (size_t)4 * (64 * ((W + 1) / 2)) * 16 + ((size_t)2 * maxmatch * MROW + (size_t)4 * 2 * ROW + (size_t)4 * MROW + (size_t)4 * 8 * NW + (size_t)4 * 32 + (size_t)(seq ? 4 * HARC_OWN_SLOTS : 0) + (size_t)2 * nprobe + 8) * 4 + 32;      // (no command block: only the cooperative kernel scans by workgroup)
}
// Hamming distance between a candidate read (registers) and the consensus shifted by the lane's own amount: `row` = ref or rref window
// row in LDS, bitoff = 32 NW +- 2j, mrow = mask row of (direction, shift) (reorder.cpp:543,608 with mask[j] / revmask[j] of :706-718)
template <int W> __device__ __forceinline__ int ham_window(const uint32_t *row, int bitoff, const uint32_t *mrow, const uint32_t (&rd)[2 * W])
{
    constexpr int NW = 2 * W;
    const int i0 = bitoff >> 5, sh = bitoff & 31;
    uint32_t d[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; k++) d[k] = row[i0 + k];
    int hd = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) hd += __popc((__builtin_amdgcn_alignbit(d[k + 1], d[k], sh) ^ rd[k]) & mrow[k]);
    return hd;
}
template <int W> __device__ __forceinline__ void load_read32(const uint64_t *reads, uint32_t rid, uint32_t (&rd)[2 * W])
{
    if (W % 2 == 0) {                                            // 16 W bytes per read: 16-byte aligned rows
        const uint4 *p = reinterpret_cast<const uint4 *>(reads + (size_t)rid * W);
#pragma unroll
        for (int q = 0; q < W / 2; q++) { const uint4 v = p[q]; rd[4 * q] = v.x; rd[4 * q + 1] = v.y; rd[4 * q + 2] = v.z; rd[4 * q + 3] = v.w; }
    } else {
        const uint2 *p = reinterpret_cast<const uint2 *>(reads + (size_t)rid * W);
#pragma unroll
        for (int q = 0; q < W; q++) { const uint2 v = p[q]; rd[2 * q] = v.x; rd[2 * q + 1] = v.y; }
    }
}
// updaterefcount (reorder.cpp:884-909) with the accepted read's dwords in LDS (rdl): every lane picks the bases of its own columns
template <int W, bool I> __device__ __forceinline__ void cons_update_lds(ConsState<W, I> &st, const uint32_t *rdl, int L, int rev, int shift, int lane)
{
    constexpr int LP = ConsState<W, I>::LP;
    int nb = st.base + shift; if (nb >= LP) nb -= LP;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        const int p = lane + 64 * t;
        int oldl = p - st.base; if (oldl < 0) oldl += LP;
        int newl = p - nb; if (newl < 0) newl += LP;
        uint4 q = st.getq(t); int v = 0;
        if (newl < L) {
            const int sc = rev ? (L - 1 - newl) : newl;
            const int pc = (int)((rdl[sc >> 4] >> (2 * (sc & 15))) & 3u);
            int b = ((pc & 1) << 1) | (pc >> 1);                 // packed code A0 G1 C2 T3 -> count row A0 C1 G2 T3
            if (rev) b = 3 - b;
            if (oldl < L && oldl >= shift) { q.x += (b == 0); q.y += (b == 1); q.z += (b == 2); q.w += (b == 3); v = argmax4(q); }
            else { q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3); v = b; }
        } else q = make_uint4(0, 0, 0, 0);
        st.setq(t, q); st.v[t] = v;
    }
    st.base = nb;
}
template <int W, bool I> __device__ __forceinline__ void cons_reset_lds(ConsState<W, I> &st, const uint32_t *rdl, int L, int lane)
{
    st.base = 0;
#pragma unroll
    for (int t = 0; t < ConsState<W, I>::CT; t++) {
        const int i = lane + 64 * t;
        int b = 0; uint4 q = make_uint4(0, 0, 0, 0);
        if (i < L) {
            const int pc = (int)((rdl[i >> 4] >> (2 * (i & 15))) & 3u);
            b = ((pc & 1) << 1) | (pc >> 1);
            q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3);
        }
        st.setq(t, q); st.v[t] = b;
    }
}

// consensus -> the wave's window rows in LDS: rowF = packed consensus (reorder.cpp `ref`), rowR = its reverse complement (`revref`).
// Every lane drops the 2-bit codes of its ring slots as bytes at their column (and, complemented, at the mirrored column); 2 NW lanes
// then squeeze 16 bytes into one dword each.  (The ballot + bit-spread formulation cost 380 vector instructions per step.)
template <int W, bool I> __device__ __forceinline__ void cons_rows(const ConsState<W, I> &st, int L, int lane, uint8_t *tmp, uint32_t *rowF, uint32_t *rowR)
{
    constexpr int CT = ConsState<W, I>::CT, LP = ConsState<W, I>::LP, NW = 2 * W;
#pragma unroll
    for (int t = 0; t < CT; t++) {
        int l = lane + 64 * t - st.base; if (l < 0) l += LP;
        if (l < L) {
            const int v = st.v[t];
            const int pc = ((v & 1) << 1) | (v >> 1);              // count row A0 C1 G2 T3 -> packed code A0 G1 C2 T3
            tmp[l] = (uint8_t)pc;
            tmp[16 * NW + (L - 1 - l)] = (uint8_t)(3 - pc);        // complement of the packed code is 3 - code
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 2 * NW) {
        const uint4 x = *reinterpret_cast<const uint4 *>(tmp + 16 * lane);
        auto pk = [](uint32_t b) -> uint32_t { b = (b | (b >> 6)) & 0x000F000Fu; return (b | (b >> 12)) & 0xFFu; };   // 4 bytes of 2 bits -> 8 bits
        const uint32_t d = pk(x.x) | (pk(x.y) << 8) | (pk(x.z) << 16) | (pk(x.w) << 24);
        (lane < NW ? rowF + NW + lane : rowR + lane)[0] = d;       // rowR + NW + (lane - NW)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A step whose read agrees with the consensus on EVERY column of the overlap (Hamming distance 0: all steps on error-free data, a third of
// them at 1 % errors) leaves a consensus that IS the oriented read: the overlap columns keep their majority (the count of the base that already
// led goes up by one), the columns behind it are the read's (reorder.cpp:884-909).  Such a step takes the window rows straight from the read
// -- one row is the read as stored, the other its reverse complement, 2 NW lanes make a dword each -- instead of the ~45 vector instructions of
// cons_rows, and it does NOT touch the column counts: it only notes its shift (cons_flush applies a whole run of such steps at once).
template <int W> __device__ __forceinline__ void rows_from_read(const uint32_t *rdl, int L, int rev, int lane, uint32_t *rowF, uint32_t *rowR)
{
    constexpr int NW = 2 * W;
    // (the lane number goes through an empty asm: everything below would otherwise be hoisted out of the step loop as a dozen loop-invariant
    // lane masks and addresses, and the kernel has no register left for them -- they came back as scratch reloads behind s_waitcnt vmcnt(0))
    int ln = lane; asm volatile("" : "+v"(ln));
    if (ln < 2 * NW) {
        const bool copy = ln < NW;
        const int k = copy ? ln : ln - NW;                        // dword of the row
        // field i of the reverse complement = 3 - field (L - 1 - i) of the read: the 16 fields of dword k come from the 32-bit window of the
        // read at bit P = 2 (L - 16 k - 16), field order reversed, complemented; what lies outside the read is zero on both sides
        const int P = 2 * (L - 16 * k - 16), i0 = P >> 5;
        const int ia = i0 < 0 ? 0 : (i0 > NW - 1 ? NW - 1 : i0), ib = i0 + 1 < 0 ? 0 : (i0 + 1 > NW - 1 ? NW - 1 : i0 + 1);
        const uint32_t ra = rdl[copy ? k : ia], rb = rdl[ib];
        const uint32_t lo = (i0 >= 0 && i0 < NW) ? ra : 0u, hi = (i0 + 1 >= 0 && i0 + 1 < NW) ? rb : 0u;
        uint32_t w = __brev(__builtin_amdgcn_alignbit(hi, lo, P & 31));
        w = ((w & 0xAAAAAAAAu) >> 1) | ((w & 0x55555555u) << 1);
        const int vb = 2 * L - 32 * k;                             // bits of the 2L-bit row that fall into this dword
        const uint32_t d = copy ? ra : (~w & (vb >= 32 ? 0xFFFFFFFFu : (vb <= 0 ? 0u : ((1u << vb) - 1u))));
        (copy != (rev != 0) ? rowF : rowR)[NW + k] = d;            // forward match: rowF = read; reverse match: rowF = its reverse complement
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// The column counts after `pend` steps of that kind, whose cumulative shifts are ps[0 .. pend) (ptot = the last): column l of the consensus of
// now (its base: rowF) was covered by step k exactly when ptot - ps[k] < L - l; where it existed before the run (l + ptot < L: the same ring
// slot, nothing moved) its old counts stay and the leading base gains one per step, else it starts from the steps that cover it.  Identical
// to applying cons_update_lds step by step -- the tests force both ways (HARC_AMD_LAZY=0) onto the same bytes.
template <int W, bool I> __device__ __forceinline__ void cons_flush(ConsState<W, I> &st, const uint16_t *ps, int pend, int ptot, const uint32_t *rowF, int L, int lane)
{
    constexpr int LP = ConsState<W, I>::LP, CT = ConsState<W, I>::CT, NW = 2 * W;
    int nb = 0;
    if (ptot < L) { nb = st.base + ptot; if (nb >= LP) nb -= LP; }         // beyond that no old column is left and any base will do
    int ln = lane; asm volatile("" : "+v"(ln));                    // nothing of this is to be hoisted out of the step loop (see rows_from_read)
    int newl[CT], cnt[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) { newl[t] = ln + 64 * t - nb; if (newl[t] < 0) newl[t] += LP; cnt[t] = 0; }
    for (int k = pend - 1; k >= 0; k--) {                         // the latest steps first: those more than L behind cover nothing
        const int d = __builtin_amdgcn_readfirstlane(ptot - (int)ps[k]);     // wave-uniform
        if (d >= L) break;
#pragma unroll
        for (int t = 0; t < CT; t++) cnt[t] += (d < L - newl[t]) ? 1 : 0;
    }
#pragma unroll
    for (int t = 0; t < CT; t++) {
        uint4 q = make_uint4(0, 0, 0, 0); int v = 0;
        if (newl[t] < L) {
            const int pc = (int)((rowF[NW + (newl[t] >> 4)] >> (2 * (newl[t] & 15))) & 3u);
            v = ((pc & 1) << 1) | (pc >> 1);                       // packed code A0 G1 C2 T3 -> count row A0 C1 G2 T3
            if (newl[t] + ptot < L) q = st.getq(t);
            const uint32_t c = (uint32_t)cnt[t];
            q.x += v == 0 ? c : 0u; q.y += v == 1 ? c : 0u; q.z += v == 2 ? c : 0u; q.w += v == 3 ? c : 0u;
        }
        st.setq(t, q); st.v[t] = v;
    }
    st.base = nb;
}

// ---- the COOP kernel gives a whole workgroup to one chain: wave 0 walks the chain (the same code as the main kernel), waves 1..3
//      wait for scan commands and share the scans of the large bins -- 256 candidates per round trip.  (One wave did up to ~600
//      chunk iterations in the heaviest walk of a super-round while the chip idled.)
struct WgCmd {
    int op;                          // 1 = scan, 0 = leave
    int l, t, fast;
    uint32_t ids0, m0, cnt;
    uint32_t rot;                    // the scan starts at the rot-th entry from the top and takes the rot entries above it last (schedule rule: below)
    unsigned long long grp;          // the probes (lanes of wave 0) that look into this bin
    uint32_t found;                  // the read of the best probe that found one
    uint32_t nc;                     // candidates tested by the helpers (statistics)
    unsigned long long hit[2][HARC_SCAN_CH][4];
    unsigned long long um[2][HARC_SCAN_CH][4];     // which entries of each wave count for the maxsearch window
    uint8_t mj[64], mdir[64];        // shift and direction of every probe
    uint32_t own[64];                // reads the chain took earlier in this super-round
    // wg_scan_sk (the scan by sketch): which entries of every (quarter, wave) count for the maxsearch window; every wave's earliest hit; every wave's queue of
    // the entries whose sketch did not settle them
    unsigned long long um4[2][HARC_SK_E][4];
    uint32_t best[2][4];
    uint16_t queue[4][64 * HARC_SK_E];
};
struct WgResult { int besthit, fj, fdir; uint32_t iters, tests, nc;
#ifdef HARC_SKETCH_STATS
    uint32_t rej1, rej2, acc;         // experiment (make variant VFLAGS=-DHARC_SKETCH_STATS): tests a 16- / 32-base sketch of the candidate would have rejected; tests that passed
#endif
};
static_assert(sizeof(WgCmd) <= HARC_WGCMD_BYTES, "HARC_WGCMD_BYTES too small");
// every wave of the workgroup: scan the bin of the command; same order, same maxsearch window as a lane-serial scan (reorder.cpp:540-552).
// No claim bit is asked for: k_compact_bins runs after the last change of the claim bitmap of a super-round (k_resolve, k_reseed), so every
// entry of a large bin is unclaimed in the frozen state this kernel walks against; what remains to be excluded are the reads the chain
// took earlier in this super-round (cmd->own).  One chunk of 256 entries = ONE round trip to HBM (ids and the bin-ordered reads
// together) and ONE barrier: the waves post their hit masks -- and, above maxsearch, which of their entries count for the window -- in
// the same exchange (double-buffered); only the chunk in which the window closes needs the exact ranks and a second exchange.
template <int W, int NWV> __device__ __forceinline__ WgResult wg_scan(WgCmd *cmd, int role, int lane, const uint32_t *const *ids, const uint64_t *mirror,
                                                            const uint32_t *rowF, const uint32_t *s_mask, uint32_t *rdl, int maxsearch, int maxmatch, int thresh)
{
    constexpr int NW = StepsLds<W>::NW, ROW = StepsLds<W>::ROW, MROW = StepsLds<W>::MROW, CH = HARC_SCAN_CH;
    WgResult r; r.besthit = 64; r.fj = 0; r.fdir = 0; r.iters = 0; r.tests = 0; r.nc = 0;
#ifdef HARC_SKETCH_STATS
    r.rej1 = r.rej2 = r.acc = 0;
#endif
    const uint32_t *oids = ids[cmd->l];
    const uint32_t ids0 = cmd->ids0, m0 = cmd->m0;
    const int t = cmd->t; const bool fast = cmd->fast != 0;
    unsigned long long grp = cmd->grp;
    // Schedule rule of round 3 (the oracle's scan_bin has the same): the scan of a large bin starts at entry number rot from the top -- rot =
    // (chain * 0x9E3779B1 >> 8) mod min(live entries, maxsearch), 0 for chain 0: one chain stays the reference at -t 1 -- goes down to the
    // end of the bin or of the maxsearch window, and takes the rot entries above its start last.  The chains that sit in one repeat family
    // all wanted the highest unclaimed id of the same bin; all but one lost the bid and the rest of their walk (c2r: 4.26 M steps walked to
    // keep 2.91 M).  Two segments: entries [0, cnt - rot) from their top, then [cnt - rot, cnt) from theirs.
    int seen = 0, par = 0;
    const uint32_t rot = cmd->rot;
    uint32_t pos = cmd->cnt - rot, lo = 0; bool second = rot == 0;
    uint32_t mrd[CH][NW], rid[CH];
    for (;;) {
        if (pos == lo) { if (second) break; second = true; pos = cmd->cnt; lo = cmd->cnt - rot; }
        if (!(seen < maxsearch && grp)) break;
        // CH chunks of 256 entries per round trip; entry c * 256 + 64 * role + lane from the top of the bin, so priority = (c, role, lane).
        // CH = 1: with 2 or 4 (a bin of diverged repeat copies is scanned to the end of the window, four dependent round trips) the kernel
        // needs 153 / 209 registers instead of 124 and loses a wave or two per SIMD: c3sd chains 419 -> 489 / 703 ms.  A chunk costs ~1.5 us
        // whatever it waits for: the walking wave is alone on its SIMD lane of the workgroup and bound by instruction latency.
        bool valid[CH], own[CH], un[CH], cand[CH]; unsigned long long um[CH];
        r.iters++;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t off = 64u * NWV * (uint32_t)c + 64u * (uint32_t)role + (uint32_t)lane;
            valid[c] = off < pos - lo; rid[c] = 0; own[c] = false;
            if (valid[c]) { const uint32_t at = pos - 1 - off; rid[c] = oids[ids0 + at]; load_read32<W>(mirror, m0 + at, mrd[c]); }
        }
        // fast (the bin fits the maxsearch window, which then never closes): the chain's own reads are only looked up for candidates that
        // pass the Hamming test.  Otherwise they do not count for the window either: looked up for every entry.
        bool ownknown = !fast;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (!fast && valid[c]) for (int k = 0; k < t; k++) own[c] |= (cmd->own[k] == rid[c]);
            un[c] = valid[c] && !own[c]; cand[c] = un[c];
            um[c] = __ballot(un[c]);
        }
        int total = 0;
        unsigned long long gm = grp; r.tests += (uint32_t)__popcll(grp);
        bool first = true;
        while (gm) {                                               // the probes of this bin, highest priority first
            const int g = __ffsll((long long)gm) - 1;
            gm &= gm - 1;
            const int g_j = cmd->mj[g], g_dir = cmd->mdir[g];
            const uint32_t *const omrow = s_mask + (size_t)(g_dir * maxmatch + g_j) * MROW;
            const int obit = g_dir * ROW * 32 + 32 * NW + (g_dir ? -2 * g_j : 2 * g_j);
            bool ok[CH]; bool anyok = false;
#pragma unroll
            for (int c = 0; c < CH; c++) {
                ok[c] = false;
                if (cand[c]) { r.nc++; ok[c] = ham_window<W>(rowF, obit, omrow, mrd[c]) <= thresh; }
#ifdef HARC_SKETCH_STATS
                if (cand[c]) {   // the Hamming distance over any part of the overlap is a lower bound of the distance over all of it: would 16 / 32 bases of the candidate
                                 // that every probe of this direction overlaps (forward: the first ones; reverse: bases 16 (NW - 4) ... of a read of >= 16 (NW - 2) + 50 bases) have settled it?
                    const int i0 = obit >> 5, sh = obit & 31, k0 = g_dir ? NW - 4 : 0;
                    const int p1 = __popc((__builtin_amdgcn_alignbit(rowF[i0 + k0 + 1], rowF[i0 + k0], sh) ^ mrd[c][k0]) & omrow[k0]);
                    const int p2 = p1 + __popc((__builtin_amdgcn_alignbit(rowF[i0 + k0 + 2], rowF[i0 + k0 + 1], sh) ^ mrd[c][k0 + 1]) & omrow[k0 + 1]);
                    r.rej1 += p1 > thresh; r.rej2 += p2 > thresh; r.acc += ok[c];
                }
#endif
                anyok |= ok[c];
            }
            if (anyok && !ownknown) {
#pragma unroll
                for (int c = 0; c < CH; c++) if (ok[c]) for (int k = 0; k < t; k++) own[c] |= (cmd->own[k] == rid[c]);
                ownknown = true;
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                ok[c] = ok[c] && !own[c];
                const unsigned long long pm = __ballot(ok[c]);
                if (lane == 0) { cmd->hit[par][c][role] = pm; if (first && !fast) cmd->um[par][c][role] = um[c]; }
            }
            __syncthreads();
            if (first && !fast) {
                int before[CH];                                    // entries that count, ahead of this wave's entries of chunk c
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    before[c] = total;
                    for (int w = 0; w < NWV; w++) { const int x = __popcll(cmd->um[par][c][w]); if (w < role) before[c] += x; total += x; }
                }
                if (seen + total > maxsearch) {                    // the window closes inside these chunks: exact ranks, once more
                    par ^= 1;
#pragma unroll
                    for (int c = 0; c < CH; c++) {
                        cand[c] = un[c] && (seen + before[c] + __popcll(um[c] & ((1ULL << lane) - 1ULL)) < maxsearch);
                        ok[c] = ok[c] && cand[c];
                        const unsigned long long pm = __ballot(ok[c]);
                        if (lane == 0) cmd->hit[par][c][role] = pm;
                    }
                    __syncthreads();
                }
            }
            int winc = -1, winw = 0; unsigned long long winm = 0;
#pragma unroll
            for (int c = CH - 1; c >= 0; c--)
                for (int w = NWV - 1; w >= 0; w--) { const unsigned long long h = cmd->hit[par][c][w]; if (h) { winc = c; winw = w; winm = h; } }
            first = false;
            par ^= 1;
            if (winc >= 0) {                                       // the highest id: first chunk, first wave, first lane
                const int wl = __ffsll((long long)winm) - 1;
                if (role == winw && lane == wl) {
#pragma unroll
                    for (int c = 0; c < CH; c++) if (c == winc) {
#pragma unroll
                        for (int k = 0; k < NW; k++) rdl[k] = mrd[c][k];
                        cmd->found = rid[c];
                    }
                }
                r.fj = g_j; r.fdir = g_dir; r.besthit = g;
                grp &= (1ULL << g) - 1ULL;                         // this probe is settled, the ones behind it are beaten; the ones before it go on
                break;
            }
        }
        if (!fast) seen += total;
        pos -= pos - lo > 64u * NWV * CH ? 64u * NWV * CH : pos - lo;
    }
    __syncthreads();                                               // the winner's words are in place; the command block is free for the next scan
    return r;
}

// ---- the same scan BY SKETCH (round 6).  On repeat-rich input nearly every Hamming test of these scans fails: configs[2] with human-like repeats makes
//      1.56e11 of them per step and 0.22 % pass (profiles/r06/sketch_rejection_c3r.txt) -- up to maxsearch of them per probe into a bin of a diverged
//      repeat family is what reorder.cpp:540-556 asks for -- and a walk waits a trip to L2 per 64 NWV entries: sixteen dependent trips through a bin of a
//      thousand with one wave per walk (the shape the kernel takes at 33 000 walks per super-round).  The Hamming distance over ANY part of the overlap is a
//      lower bound of the distance over all of it: a lane fetches, of HARC_SK_E = 4 entries at once, only the two dwords of the read that every probe of the
//      direction overlaps (forward: bases 0..31; reverse: the 32 bases from dword NW - 4 on) -- as many registers as ONE whole read -- and 83 % of the
//      entries are settled there, exactly.  The survivors are queued in LDS in priority order and take the whole test, a lane each, in a second trip:
//      two dependent trips per 256 NWV entries instead of four, a quarter of the barriers.  Same order, same maxsearch window, same winner as wg_scan:
//      the bytes are the oracle's either way (make variant VFLAGS=-DHARC_SCAN_SKETCH=0 builds the other one; both are held to the oracle by the suite).
template <int W, int NWV> __device__ __forceinline__ WgResult wg_scan_sk(WgCmd *cmd, int role, int lane, const uint32_t *const *ids, const uint64_t *mirror,
                                                               const uint32_t *rowF, const uint32_t *s_mask, uint32_t *rdl, int maxsearch, int maxmatch, int thresh)
{
    constexpr int NW = StepsLds<W>::NW, ROW = StepsLds<W>::ROW, MROW = StepsLds<W>::MROW, E = HARC_SK_E;
    constexpr int K0R = NW >= 4 ? NW - 4 : 0;                     // reverse probes keep the bits from 2j on, j < maxmatch <= L / 2: these two dwords lie behind every j
    WgResult r; r.besthit = 64; r.fj = 0; r.fdir = 0; r.iters = 0; r.tests = 0; r.nc = 0;
#ifdef HARC_SKETCH_STATS
    r.rej1 = r.rej2 = r.acc = 0;
#endif
    const uint32_t *oids = ids[cmd->l];
    const uint32_t ids0 = cmd->ids0, m0 = cmd->m0;
    const int t = cmd->t; const bool fast = cmd->fast != 0;
    unsigned long long grp = cmd->grp;
    // the chain's own reads of this super-round (they are not in the frozen claim bitmap): asked for an entry only when its id lies between the smallest and the largest of them
    uint32_t omin = 0xFFFFFFFFu, omax = 0u;
    for (int k = 0; k < t; k++) { const uint32_t x = cmd->own[k]; omin = x < omin ? x : omin; omax = x > omax ? x : omax; }
    auto is_own = [&](uint32_t rid) -> bool { bool o = false; if (rid >= omin && rid <= omax) for (int k = 0; k < t; k++) o |= (cmd->own[k] == rid); return o; };
    uint16_t *const queue = cmd->queue[role];
    int seen = 0, par = 0;
    const uint32_t rot = cmd->rot;                                // (the two segments of wg_scan: schedule rule of round 3)
    uint32_t pos = cmd->cnt - rot, lo = 0; bool second = rot == 0;
    for (;;) {
        if (pos == lo) { if (second) break; second = true; pos = cmd->cnt; lo = cmd->cnt - rot; }
        if (!(seen < maxsearch && grp)) break;
        r.iters++;
        // entry off = (e NWV + role) 64 + lane from the top of what is left of the segment: priority = (e, role, lane) = ascending off
        unsigned long long um[E];                                  // per quarter: the lanes whose entry counts for the window (valid, not the chain's own)
        uint2 sk[E];
        const unsigned long long gfirst = grp & (0ULL - grp);
        const int dir0 = cmd->mdir[__ffsll((long long)gfirst) - 1];      // the probes into one bin share key and dictionary; their direction differs only for a palindromic window
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t off = 64u * (uint32_t)(e * NWV + role) + (uint32_t)lane;
            const bool valid = off < pos - lo;
            bool un = valid; sk[e] = make_uint2(0u, 0u);
            if (valid) {
                const uint32_t at = pos - 1 - off;
                const uint32_t *row = reinterpret_cast<const uint32_t *>(mirror + (size_t)(m0 + at) * W);
                sk[e] = *reinterpret_cast<const uint2 *>(row + (dir0 ? K0R : 0));
                if (!fast && t > 0) un = !is_own(oids[ids0 + at]);    // above maxsearch the chain's own reads do not count for the window either: asked of every entry
            }
            um[e] = __ballot(un);
        }
        // the maxsearch window (reorder.cpp:540): entries that count, in priority order, up to maxsearch of them.  While the bin fits the window it never closes.
        unsigned long long cm[E];                                  // candidates: count and lie inside the window
        int total = 0;
#pragma unroll
        for (int e = 0; e < E; e++) cm[e] = um[e];
        if (!fast) {
            if (NWV > 1) {
                if (lane == 0) {
#pragma unroll
                    for (int e = 0; e < E; e++) cmd->um4[par][e][role] = um[e];
                }
                __syncthreads();
            }
            int before[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                before[e] = total;
                for (int w = 0; w < NWV; w++) { const int x = NWV > 1 ? __popcll(cmd->um4[par][e][w]) : __popcll(um[e]); if (w < role) before[e] += x; total += x; }
            }
            if (NWV > 1) par ^= 1;
            if (seen + total > maxsearch) {                        // the window closes inside this stretch: exact ranks
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int rank = seen + before[e] + __popcll(um[e] & ((1ULL << lane) - 1ULL));
                    cm[e] = __ballot(((um[e] >> lane) & 1ULL) && rank < maxsearch);
                }
            }
        }
        unsigned long long gm = grp; r.tests += (uint32_t)__popcll(grp);
        while (gm) {                                               // the probes of this bin, highest priority first
            const int g = __ffsll((long long)gm) - 1;
            gm &= gm - 1;
            const int g_j = cmd->mj[g], g_dir = cmd->mdir[g];
            const uint32_t *const omrow = s_mask + (size_t)(g_dir * maxmatch + g_j) * MROW;
            const int obit = g_dir * ROW * 32 + 32 * NW + (g_dir ? -2 * g_j : 2 * g_j);
            // (1) the sketch: two dwords of the candidate against the same two of the shifted consensus, under the probe's mask
            unsigned long long sm[E]; int nsurv = 0;
            {
                const int i0 = obit >> 5, sh = obit & 31, k0 = g_dir ? K0R : 0;
                const bool same = g_dir == dir0;                   // (else this probe's dwords were not fetched: nothing is settled by sketch)
                const uint32_t cA = __builtin_amdgcn_alignbit(rowF[i0 + k0 + 1], rowF[i0 + k0], sh), cB = __builtin_amdgcn_alignbit(rowF[i0 + k0 + 2], rowF[i0 + k0 + 1], sh);
                const uint32_t mA = omrow[k0], mB = omrow[k0 + 1];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool c = (cm[e] >> lane) & 1ULL;
                    const int lb = __popc((cA ^ sk[e].x) & mA) + __popc((cB ^ sk[e].y) & mB);
                    if (c) r.nc++;
                    sm[e] = __ballot(c && (!same || lb <= thresh));
                    // the survivors of this wave, in priority order, into its queue
                    if ((sm[e] >> lane) & 1ULL) queue[nsurv + __popcll(sm[e] & ((1ULL << lane) - 1ULL))] = (uint16_t)(64u * (uint32_t)(e * NWV + role) + (uint32_t)lane);
                    nsurv += __popcll(sm[e]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (2) the survivors take the whole test, a lane each; the earliest one that passes (and is not the chain's own) is this wave's hit
            uint32_t mybest = 0xFFFFFFFFu, myrid = HARC_NONE; uint32_t full[NW];
            for (int b = 0; b < nsurv; b += 64) {
                const int i = b + lane;
                bool ok = false; uint32_t o = 0, rid = 0;
                if (i < nsurv) {
                    o = queue[i];
                    const uint32_t at = pos - 1 - o;
                    load_read32<W>(mirror, m0 + at, full);
                    ok = ham_window<W>(rowF, obit, omrow, full) <= thresh;
                    if (ok) { rid = oids[ids0 + at]; if (fast && t > 0) ok = !is_own(rid); }      // (below maxsearch the own reads are only asked for what passes)
#ifdef HARC_SKETCH_STATS
                    r.acc += ok;
#endif
                }
                const unsigned long long pm = __ballot(ok);
                if (pm) {
                    const int wl = __ffsll((long long)pm) - 1;
                    mybest = (uint32_t)__builtin_amdgcn_readlane((int)o, wl);
                    if (lane == wl) myrid = rid; else myrid = HARC_NONE;      // the lane that holds the winner's words keeps them in `full`
                    break;
                }
            }
            // (3) the earliest hit of the workgroup
            uint32_t wbest = mybest; int wrole = role;
            if (NWV > 1) {
                if (lane == 0) cmd->best[par][role] = mybest;
                __syncthreads();
                wbest = 0xFFFFFFFFu; wrole = 0;
                for (int w = 0; w < NWV; w++) { const uint32_t x = cmd->best[par][w]; if (x < wbest) { wbest = x; wrole = w; } }
                par ^= 1;
            }
            if (wbest != 0xFFFFFFFFu) {
                if (role == wrole && myrid != HARC_NONE) {
#pragma unroll
                    for (int k = 0; k < NW; k++) rdl[k] = full[k];
                    cmd->found = myrid;
                }
                r.fj = g_j; r.fdir = g_dir; r.besthit = g;
                grp &= (1ULL << g) - 1ULL;                         // this probe is settled, the ones behind it are beaten; the ones before it go on
                break;
            }
        }
        if (!fast) seen += total;
        pos -= pos - lo > 64u * NWV * E ? 64u * NWV * E : pos - lo;
    }
    __syncthreads();                                               // the winner's words are in place; the command block is free for the next scan
    return r;
}

// What every workgroup of k_steps needs in LDS and used to compute for itself at every launch (with integer divisions: a fifth of the
// kernel's vector instructions): out[0 .. 2 maxmatch MROW) = the mask rows -- mask[j] keeps the low 2(L-j) bits, revmask[j] the bits >= 2j
// below 2L (reorder.cpp:706-718); then one uint2 per probe p of a step (reorder.cpp:517-649 order): x = bit offset of its key window inside
// the wave's two rows | dir << 13 | dict << 14 | shift << 16;  y = bit offset of its Hamming window | dword offset of its mask row << 16
template <int W> __global__ void k_steps_tables(S1Args s, uint32_t *out)
{
    constexpr int NW = StepsLds<W>::NW, ROW = StepsLds<W>::ROW, MROW = StepsLds<W>::MROW;
    const int L = s.L, nm = 2 * s.maxmatch * MROW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
        const int k = i % MROW, r = i / MROW, dir = r / s.maxmatch, j = r % s.maxmatch;
        const int lo = dir ? 2 * j : 0, hi = dir ? 2 * L : 2 * (L - j);      // bits [lo, hi)
        const int a = lo - 32 * k, b = hi - 32 * k;
        const uint32_t mh = b >= 32 ? 0xFFFFFFFFu : (b <= 0 ? 0u : ((1u << b) - 1u)), ml = a >= 32 ? 0xFFFFFFFFu : (a <= 0 ? 0u : ((1u << a) - 1u));
        out[i] = k < NW ? (mh & ~ml) : 0u;
    }
    uint2 *const pinfo = reinterpret_cast<uint2 *>(out + nm);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.nprobe; i += gridDim.x * blockDim.x) {
        const uint32_t e = s.probe_tab[i];
        const int j = (int)(e & 0xFF), dir = (int)((e >> 8) & 1), l = (int)((e >> 9) & 1);
        const int koff = dir * ROW * 32 + 32 * NW + (dir ? 2 * (s.ds[l] - j) : 2 * (s.ds[l] + j));
        const int hoff = dir * ROW * 32 + 32 * NW + (dir ? -2 * j : 2 * j);      // the consensus is shifted by 2j bits (reorder.cpp:647-648)
        pinfo[i] = make_uint2((uint32_t)koff | ((uint32_t)dir << 13) | ((uint32_t)l << 14) | ((uint32_t)j << 16),
                              (uint32_t)hoff | ((uint32_t)((dir * s.maxmatch + j) * MROW) << 16));
    }
}

// ---- Exact mode (ONE chain: the reference at -t 1, byte for byte) and runs with a handful of chains leave the chip idle while a single wave
//      chases its chain read by read, ~700 dependent instructions, ~30 LDS round trips and three trips to memory per step.  Nearly all of that
//      answers a question that does not depend on the walk: a step that agreed with the consensus on every column of the overlap (Hamming
//      distance 0: every step on error-free reads, most of them at a fraction of a percent of errors) leaves a consensus that IS the read just
//      taken, in the orientation it was taken in (rows_from_read) -- so which reads the NEXT step may take, in the reference's priority order
//      (reorder.cpp:517-649: shift by shift, dictionary by dictionary, every bin from its highest id down, Hamming distance <= thresh), is a
//      function of (read, orientation) and of nothing else.  k_succ asks it for every read and both orientations at once, with the same probe
//      table, bitmaps, tables and mask rows as k_steps: one wave per (read, orientation), HARC_SUCC_N entries {read, shift | dir << 8 |
//      distance << 9 | probe << 16}.  What the walk then has to add is what does depend on it -- the first entry that is neither claimed nor
//      taken by the chain in the running super-round: one line of 64 bytes, the claim words, no key, no table, no read (k_steps' FASTP).  A list
//      that stops early (HARC_SUCC_OPEN: a bin of more than HARC_LARGEBIN reads among the probes -- their scan order is the walk's business --,
//      or more candidates than entries) sends the walk back to the probes when all its entries are taken.
template <int W> __global__ __launch_bounds__(256) void k_succ(S1Args s, uint2 *succ, uint32_t n)
{
    constexpr int NW = StepsLds<W>::NW, ROW = StepsLds<W>::ROW, MROW = StepsLds<W>::MROW;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int nm = 2 * s.maxmatch * MROW;
    uint32_t *const s_mask = lds;
    uint2 *const s_pinfo = reinterpret_cast<uint2 *>(s_mask + nm);
    uint32_t *const s_rows = reinterpret_cast<uint32_t *>(s_pinfo + s.nprobe);
    uint32_t *const s_rdl = s_rows + 4 * 2 * ROW;
    uint2 *const s_out = reinterpret_cast<uint2 *>(s_rdl + 4 * MROW);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nm; i += 256) s_mask[i] = s.lds_tab[i];
    { const uint2 *const pt = reinterpret_cast<const uint2 *>(s.lds_tab + nm); for (int i = threadIdx.x; i < s.nprobe; i += 256) s_pinfo[i] = pt[i]; }
    for (int i = threadIdx.x; i < 4 * 2 * ROW; i += 256) s_rows[i] = 0u;
    __syncthreads();
    const uint64_t g = harc_bid() * 4 + wv;                      // (a wave per (read, orientation): 128 n work-items pass 2^32 at 33.5 M reads -- folded grid)
    if (g >= 2ull * n) return;
    const uint32_t r = (uint32_t)(g >> 1); const int o = (int)(g & 1);
    const int L = s.L;
    uint32_t *const rowF = s_rows + (size_t)wv * 2 * ROW, *const rowR = rowF + ROW, *const rdl = s_rdl + (size_t)wv * MROW;
    uint2 *const out = s_out + (size_t)wv * HARC_SUCC_N;
    if (lane < NW) rdl[lane] = reinterpret_cast<const uint32_t *>(s.reads)[(size_t)r * NW + lane];
    if (lane < HARC_SUCC_N) out[lane] = make_uint2(HARC_NONE, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rows_from_read<W>(rdl, L, o, lane, rowF, rowR);
    const uint64_t kmask0 = s.kbits[0] >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << s.kbits[0]) - 1), kmask1 = s.kbits[1] >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << s.kbits[1]) - 1);
    const uint64_t cap = s.cap[0];
    uint32_t total = 0; bool open = false;
    for (int base = 0; base < s.nprobe && total < HARC_SUCC_N && !open; base += 64) {
        const int p = base + lane;
        int state = 0, j = 0, dir = 0, l = 0; uint32_t sst = 0, cw = 0; uint2 pi = make_uint2(0, 0);
        if (p < s.nprobe && cap) {
            pi = s_pinfo[p];
            j = (int)(pi.x >> 16); dir = (int)((pi.x >> 13) & 1); l = (int)((pi.x >> 14) & 1);
            uint64_t key;
            {
                const int i0 = (int)((pi.x & 0x1FFF) >> 5), shb = (int)(pi.x & 31);
                const uint32_t d0 = rowF[i0], d1 = rowF[i0 + 1], d2 = rowF[i0 + 2];
                key = ((uint64_t)__builtin_amdgcn_alignbit(d1, d0, shb) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, shb) << 32)) & (l ? kmask1 : kmask0);
            }
            const HashSlot *const tab = l ? s.slots[1] : s.slots[0];
            const uint64_t hk = key_scramble(key);
            uint64_t sl = bucket_slot(hk, cap);
            if (s.bloom_lines) {
                uint32_t bw, bm;
                bloom_pos(key, hk, s.bloom_lines, s.bloom_nwin[0], s.bloom_mmask, &bw, &bm);
                if (((l ? s.bloom[1] : s.bloom[0])[bw] & bm) != bm) state = 1;
            }
            while (state == 0) {                                       // the bucketed search of k_steps: a full bucket ends it unless its overflow flag is set
                uint32_t w0 = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(&tab[sl + q]);
                    if (q == 0) w0 = raw.w;
                    if (state == 0) {
                        if (raw.w == 0) state = 1;
                        else if (raw.x == (uint32_t)hk && raw.y == (uint32_t)(hk >> 32)) { state = 2; sst = raw.z; cw = raw.w; }
                    }
                }
                if (state == 0 && !(w0 & SLOT_OVF)) state = 1;
                sl += 4; if (sl >= cap) sl = 0;
            }
        }
        const bool big = state == 2 && (cw & SLOT_BIG) != 0;
        const unsigned long long mb = __ballot(big);
        const int firstbig = mb ? __ffsll((long long)mb) - 1 : 64;
        // the candidates of this lane's bin, from the highest id down; the list wants them in (probe, position in the bin) order: count, then place
        const bool scan = state == 2 && !big && lane < firstbig;
        const uint32_t cntb = cw & SLOT_CNT_MASK; const bool emb = (cw & SLOT_EMB) != 0;
        const uint32_t *const ids = l ? s.ids[1] : s.ids[0];
        const uint32_t *const mrow = s_mask + (pi.y >> 16);
        const int bitoff = (int)(pi.y & 0xFFFF);
        uint32_t cnt = 0;
        if (scan) for (uint32_t i = cntb; i > 0; i--) {
            const uint32_t rid = emb ? sst : ids[sst + i - 1];
            if (rid == r) continue;                                    // the read itself: taken by the walk that asks
            uint32_t mrd[NW];
            load_read32<W>(s.reads, rid, mrd);
            if (ham_window<W>(rowF, bitoff, mrow, mrd) <= s.thresh) cnt++;
        }
        uint32_t incl = cnt;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += y; }
        const uint32_t off = total + incl - cnt;
        if (scan && cnt && off < HARC_SUCC_N) {
            uint32_t k = 0;
            for (uint32_t i = cntb; i > 0; i--) {
                const uint32_t rid = emb ? sst : ids[sst + i - 1];
                if (rid == r) continue;
                uint32_t mrd[NW];
                load_read32<W>(s.reads, rid, mrd);
                const int hd = ham_window<W>(rowF, bitoff, mrow, mrd);
                if (hd <= s.thresh) { if (off + k < HARC_SUCC_N) out[off + k] = make_uint2(rid, (uint32_t)j | ((uint32_t)dir << 8) | ((uint32_t)hd << 9) | ((uint32_t)p << 16)); k++; }
            }
        }
        total += (uint32_t)__shfl((int)incl, 63, 64);
        if (firstbig < 64) open = true;
    }
    if (total >= HARC_SUCC_N) open = true;                             // there may be more
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < HARC_SUCC_N) { uint2 e = out[lane]; if (open) e.y |= HARC_SUCC_OPEN; succ[g * HARC_SUCC_N + lane] = e; }
}
static inline size_t succ_lds_bytes(int W, int maxmatch, int nprobe)
{
    const int NW = 2 * W, ROW = 3 * NW + 1, MROW = (NW + 3) & ~3;
    return ((size_t)2 * maxmatch * MROW + (size_t)2 * nprobe + (size_t)4 * 2 * ROW + (size_t)4 * MROW + (size_t)4 * HARC_SUCC_N * 2) * 4 + 16;
}

// NWV (COOP only): waves per workgroup = the walking wave + NWV - 1 helpers that share its scans (64 NWV candidates per round trip).  Few
// walks per super-round are bound by the longest one: 4 waves.  More walks than the chip holds workgroups are bound by wave slots, most of
// which helpers idle in: fewer helpers, more walkers (stage1_run_w picks it from the walks of the last rounds; what is computed is the same).
// SPEC (the dense wave-uniform kernel on reads of 100 bases and more, i.e. every headline configuration): 64-bit keys in both dictionaries, bitmap lines
// by 16-base minimizers over 17 windows, a table of fewer than 2^34 slots, maxsearch above HARC_LARGEBIN, full-width batches -- what the general
// kernel asks its argument block about at run time is a constant here: no key masks, no 64-bit bucket arithmetic, one minimizer loop, and a
// dozen scalar registers less to spill (stage1_run_w checks the conditions; same bytes either way, HARC_AMD_SPEC=0 forces the general kernel)
template <int W, bool QUAD, bool COOP, bool DENSE = false, int NWV = 4, bool SEQ = false, bool SPEC = false> __global__ __launch_bounds__(256, COOP ? HARC_COOP_WAVES : (QUAD ? HARC_STEPS_WAVES_Q : (W <= 4 ? HARC_STEPS_WAVES : HARC_STEPS_WAVES - 1) + (DENSE ? (SEQ ? HARC_SEQ_EXTRA_WAVES : 2) : 0))) void k_steps(S1Args s)
{
    constexpr int NW = StepsLds<W>::NW, ROW = StepsLds<W>::ROW, MROW = StepsLds<W>::MROW;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // [mask rows: (dir, shift) -> NW dwords][window rows: 4 waves x {ref, rref}][accepted read: 4 waves][column bytes: 4 waves x 2 x 16 NW][probes]
    // [column counts: 4 waves x LP x uint4][mask rows ...]
    // (the cooperative kernel walks ONE chain per workgroup with its counts in registers: no counts area and one set of rows -- 16 KB per workgroup held
    // its one-wave form at 10 walks per compute unit where the registers allow 16)
    constexpr int NSL = COOP ? 1 : 4;                             // chains a workgroup walks
    uint32_t *const s_mask = lds + (COOP ? (size_t)0 : (size_t)4 * ConsState<W, DENSE>::LP * 4);
    uint32_t *const s_rows = s_mask + (size_t)2 * s.maxmatch * MROW;
    uint32_t *const s_rdl = s_rows + ((NSL * 2 * ROW + 3) & ~3);      // (the column bytes behind it are read 16 bytes at a time)
    uint32_t *const s_tmp = s_rdl + NSL * MROW;
    uint32_t *const s_pend = s_tmp + NSL * 8 * NW;                // per chain 64 u16: cumulative shifts of the steps whose counts are still to be applied (cons_flush)
    uint32_t *const s_own = s_pend + NSL * 32;
    // OWNT: the kernels that keep the reads a walk has taken in the LDS hash table -- the wave-uniform ones, and the whole-bucket kernel of runs with few
    // chains (there LDS is no limit, and exact mode walks 64 steps per launch: comparing every candidate with up to 63 earlier reads by v_readlane was
    // what made long walks slower than short ones)
    constexpr bool OWNT = (SEQ && HARC_SEQ_OWNT) || (QUAD && !COOP);
    uint2 *const s_pinfo = reinterpret_cast<uint2 *>(s_own + (OWNT ? 4 * HARC_OWN_SLOTS : 0));      // the table exists in the kernel that asks it only (2 KB more LDS cost the 150-bp kernel a workgroup per CU)
    WgCmd *const cmd = reinterpret_cast<WgCmd *>(reinterpret_cast<char *>(s_pinfo + s.nprobe) + 8 - ((size_t)(s_pinfo + s.nprobe) & 7));
    // main kernel: 4 chains per workgroup, one wave each.  COOP: one chain per workgroup, wave 0 walks it (role 0), waves 1..3 help with the scans
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6, wv = COOP ? 0 : role;
    const int L = s.L;
    // design (R): the four chains of a workgroup of the main kernel belong to ONE rank; the others leave before anything is set up
    if (s.own_mod > 1 && ((COOP ? (blockIdx.x >> 2) : blockIdx.x) % s.own_mod) != s.own_rem) return;
    if (COOP) {                                                   // uniform for the workgroup: nothing to do for this chain
        if (blockIdx.x >= s.K) return;
        const uint32_t fl = s.hdr[blockIdx.x].flags;
        if (!(fl & CH_ACTIVE) || !(fl & CH_COOP)) return;
    }
    {   // the whole workgroup, before any wave leaves: mask rows and probe descriptors (k_steps_tables) -> LDS
        const int nm = 2 * s.maxmatch * MROW;
        for (int i = threadIdx.x; i < nm; i += 64 * NWV) s_mask[i] = s.lds_tab[i];
        const uint2 *const pt = reinterpret_cast<const uint2 *>(s.lds_tab + nm);
        for (int i = threadIdx.x; i < s.nprobe; i += 64 * NWV) s_pinfo[i] = pt[i];
        for (int i = threadIdx.x; i < NSL * 2 * ROW; i += 64 * NWV) s_rows[i] = 0u;
        for (int i = threadIdx.x; i < NSL * 8 * NW; i += 64 * NWV) s_tmp[i] = 0u;
        if constexpr (OWNT) for (int i = threadIdx.x; i < 4 * HARC_OWN_SLOTS; i += 64 * NWV) s_own[i] = HARC_NONE;
        __syncthreads();
    }
    if (COOP && role != 0) {                                      // helpers: wait for a scan, take part, until the walk is over
        const uint32_t *const idp[2] = { s.ids[0], s.ids[1] };
        uint32_t hnc = 0;
        for (;;) {
            __syncthreads();
            if (cmd->op == 0) break;
            const WgResult r = (HARC_SCAN_SKETCH && NWV == 1) ? wg_scan_sk<W, NWV>(cmd, role, lane, idp, s.mirror, s_rows, s_mask, s_rdl, s.maxsearch, s.maxmatch, s.thresh)
                                                : wg_scan<W, NWV>(cmd, role, lane, idp, s.mirror, s_rows, s_mask, s_rdl, s.maxsearch, s.maxmatch, s.thresh);
            hnc += r.nc;
#ifdef HARC_SKETCH_STATS
            if (s.dbg) { const uint32_t a1 = wave_sum_u32(r.rej1), a2 = wave_sum_u32(r.rej2), a3 = wave_sum_u32(r.acc), a0 = wave_sum_u32(r.nc); if (lane == 0) { atomicAdd(&s.dbg[44], (unsigned long long)a0); atomicAdd(&s.dbg[45], (unsigned long long)a1); atomicAdd(&s.dbg[46], (unsigned long long)a2); atomicAdd(&s.dbg[47], (unsigned long long)a3); } }
#endif
        }
        hnc = wave_sum_u32(hnc);
        if (lane == 0 && hnc) { atomicAdd(&s.cstat_coop[blockIdx.x].y, hnc); atomicAdd(&s.cstat_coop[blockIdx.x].w, hnc); }
        return;
    }
    const long long dbg_t0 = (COOP && s.dbg) ? wall_clock64() : 0;
    // trace (HARC_AMD_TRACE): wall time of a cooperative walk by phase (setup, consensus rows, probes + small bins, large-bin scans, bid + update,
    // tail), and time and number of its scans by live entries of the bin (<= 64, 256, 1024, more)
    // (a build with -DHARC_COOP_TRACE only: the accumulators cost the cooperative kernel 30 registers and a wave per SIMD)
#ifdef HARC_COOP_TRACE
    long long dbg_tl = dbg_t0; unsigned long long dbg_ph[6] = { 0, 0, 0, 0, 0, 0 }, dbg_kt[4] = { 0, 0, 0, 0 }, dbg_kn[4] = { 0, 0, 0, 0 };
#define PH(k) do { if (COOP && s.dbg) { const long long tn_ = wall_clock64(); dbg_ph[k] += (unsigned long long)(tn_ - dbg_tl); dbg_tl = tn_; } } while (0)
#else
#define PH(k) do { } while (0)
#endif
    uint32_t ownreg = HARC_NONE;                                 // lane t: the read this chain took at step t of this super-round
    // everything about the chain is the same in all lanes of its wave: told to the compiler, so that it lives in scalar registers
    // (with the counts in LDS: 26 -> 2 spilled vector registers in the 6-wave build)
    const uint32_t c = COOP ? blockIdx.x : (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    if (c >= s.K) return;
    ChainHdr h;
    {
        const uint4 *hp = reinterpret_cast<const uint4 *>(&s.hdr[c]);
        const uint4 h0 = hp[0], h1 = hp[1];
        h.cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0.x); h.prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0.y);
        h.flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0.z); h.mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0.w);
        h.n_main = (uint32_t)__builtin_amdgcn_readfirstlane((int)h1.x); h.n_sing = (uint32_t)__builtin_amdgcn_readfirstlane((int)h1.y);
        h.nsteps = (uint32_t)__builtin_amdgcn_readfirstlane((int)h1.z); h.pad0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)h1.w);
    }
    uint4 cst = s.cstat[c];
    cst.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)cst.x); cst.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)cst.y);
    cst.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)cst.z); cst.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)cst.w);
    if (!(h.flags & CH_ACTIVE)) return;
    if (!COOP && s.bo && (s.bo[c] & 0xFFu)) return;             // sits this super-round out (back-off; resolve_body counts the rounds down)
    if (COOP && !(h.flags & CH_COOP)) return;
    if (!COOP && s.need[c]) {
        // the chain asked for a seed last super-round and k_reseed ranked it: take seed number `rank` (reorder.cpp:650-688), or finish
        const uint32_t r = s.needrank[c], R = s.rmeta[0], assigned = s.rmeta[1], got = s.rmeta[2];
        if (h.flags & CH_PREVUNM) {                               // previous seed found nothing: singleton (reorder.cpp:672-684)
            if (lane == 0) s.slog[h.prev] = make_uint2(c, h.n_sing);
            h.n_sing++;
        }
        if (r < assigned) {
            const uint32_t id = s.seedbuf[r];
            h.cur = id; h.prev = id; h.flags = ((h.flags | CH_PREVUNM) & ~CH_NEEDSEED) & 0xFFFFu; h.mode = 2;
            const uint32_t first = r * (uint32_t)s.nsugg_per_seed;
            const uint32_t ng = first >= got ? 0u : (got - first < (uint32_t)s.nsugg_per_seed ? got - first : (uint32_t)s.nsugg_per_seed);
            if ((uint32_t)lane < ng) s.sugg[(size_t)c * s.nsugg_stride + lane] = s.seedbuf[R + first + lane];
            h.nsteps = ng << 24;                                  // nothing walked, nothing to replay, look-ahead position 0
            if (lane == 0) { uint2 q = s.cst2[c]; q.x++; s.cst2[c] = q; s.need[c] = 0; }
        } else {                                                  // no reads left (reorder.cpp:670-677)
            h.flags &= ~(CH_ACTIVE | CH_PREVUNM | CH_NEEDSEED);
            if (lane == 0) { atomicAdd(&s.stats[ST_ACTIVE], ~0ULL); s.need[c] = 0; s.hdr[c] = h; }
            return;
        }
    }
    constexpr int LP = ConsState<W, DENSE>::LP;
    const uint32_t par = (h.flags & CH_PARITY) ? 1u : 0u;
    uint4 *B0 = s.cnt + ((size_t)par * s.K + c) * LP;            // state at the start of this super-round (rollback point)
    uint4 *B1 = s.cnt + ((size_t)(par ^ 1u) * s.K + c) * LP;     // state at its end
    ConsState<W, DENSE> st;
    st.ql = reinterpret_cast<uint4 *>(lds) + (size_t)wv * ConsState<W, DENSE>::LP + lane;
    const int T0 = COOP ? (int)(h.nsteps & 0xFF) : 0;           // COOP: the one step the main kernel stopped in front of
    if (COOP) {
        cons_load(st, T0 > 0 ? B1 : B0, L, lane, (h.pad0 >> (24 + 2 * (T0 > 0 ? (par ^ 1u) : par))) & 3u);
        if (lane < T0) ownreg = s.steps[(size_t)c * 64 + lane].x;
    } else if (h.mode == 2) {                                    // fresh seed (reorder.cpp:875-883)
        uint64_t rw[W];
#pragma unroll
        for (int w = 0; w < W; w++) rw[w] = s.reads[(size_t)h.cur * W + w];
        cons_reset(st, rw, L, lane);
    } else {
        cons_load(st, B0, L, lane, (h.pad0 >> (24 + 2 * par)) & 3u);
        if (h.mode == 1) {                                       // rolled back last time: replay the steps that were kept
            const int nrep = (int)((h.nsteps >> 8) & 0xFF);
            uint2 sp = make_uint2(0, 0);
            uint64_t mrw[W];
#pragma unroll
            for (int w = 0; w < W; w++) mrw[w] = 0;
            if (lane < nrep) {
                sp = s.steps[(size_t)c * 64 + lane];
#pragma unroll
                for (int w = 0; w < W; w++) mrw[w] = s.reads[(size_t)sp.x * W + w];
            }
            for (int t = 0; t < nrep; t++) {
                uint64_t rw[W];
#pragma unroll
                for (int w = 0; w < W; w++) rw[w] = shfl_u64(mrw[w], t);
                const uint32_t y = __shfl(sp.y, t, 64);
                if ((y >> 16) & 1) cons_reset(st, rw, L, lane);          // the step took a look-ahead seed
                else cons_update(st, rw, L, (int)((y >> 8) & 1), (int)(y & 0xFF), lane);
            }
        }
    }
    uint32_t widebits = (h.pad0 >> 24) & 15u;                    // the form of the two halves of the chain's count matrices (cons_store), two bits each
    if (!COOP && h.mode != 0) { const uint32_t w0 = cons_store(st, B0, L, lane); widebits = (widebits & ~(3u << (2 * par))) | (w0 << (2 * par)); }    // B0 now holds the rollback point of this super-round

    uint32_t *const rowF = s_rows + (size_t)wv * 2 * ROW, *const rowR = rowF + ROW, *const rdl = s_rdl + (size_t)wv * MROW;
    uint8_t *const coltmp = reinterpret_cast<uint8_t *>(s_tmp + (size_t)wv * 8 * NW);
    uint32_t *const ownt = s_own + (size_t)wv * HARC_OWN_SLOTS;
    uint16_t *const pshift = reinterpret_cast<uint16_t *>(s_pend + (size_t)wv * 32);
    const uint64_t kmask0 = (SPEC || s.kbits[0] >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << s.kbits[0]) - 1), kmask1 = (SPEC || s.kbits[1] >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << s.kbits[1]) - 1);
    const uint64_t cap = s.cap[0];                               // both dictionaries have the same geometry (stage1_run_w)
    uint32_t dbg_bins = 0, dbg_iter = 0, dbg_miss = 0, dbg_surv = 0, dbg_batches = 0;   // coop scans / their 64-entry chunks / steps without a hit / (unused) / batches
    uint32_t np = 0, nc = 0, nuse = 0, ncu = 0;                   // ncu: candidates a strictly sequential scan (reorder.cpp:517-649) would have tested too
    int nst = 0; bool needseed = false, defer = false;
    int lastp = (int)(h.pad0 & 0xFFFF);                          // 16 x running mean (weight 1/4) of the priority index of this chain's hits
    int spos = (int)((h.nsteps >> 16) & 0xFF); const int nsugg = (int)(h.nsteps >> 24);   // look-ahead seeds: next to try / held
    int bigprobes = 0;                                           // COOP: probes into large live bins made by this walk so far
    const int resume = (int)(h.flags >> 16);                     // the first step of this super-round was put off before: where it takes up again
    bool stalled = false; int resume_next = 0;
    // steps that agreed with the consensus everywhere (rows_from_read): the rows are current, `pend` of them wait for cons_flush, ptot = their shifts
    bool rows_ok = false; int pend = 0, ptot = 0;
    // (compiled into the wave-uniform kernels only: there the run of agreeing steps is the rule -- dense launches over error-free or nearly error-free
    // reads -- and the code fits; in the other kernels the second path cost registers: the 150-bp dense kernel lost 3.7 % to it)
    const bool lazy = (SEQ || (QUAD && !COOP)) && s.lazy != 0;
    // FASTP (the kernel of runs with few chains; k_succ): while the consensus IS a read -- `lastrid`, taken in orientation `lastdir`: after a fresh
    // seed and after every step with Hamming distance 0 -- the step reads its candidates from that read's successor list instead of asking the
    // tables.  Such a step does not even fetch the read it takes: the window rows are made from it only when something asks for them
    // (rows_lazy: a step that has to go back to the probes, a step that disagrees with the consensus somewhere, the end of the launch).
    constexpr bool FASTP = QUAD && !COOP;
    uint32_t lastrid = HARC_NONE; int lastdir = 0; bool rows_lazy = false;
    // ... and one trip to memory per step instead of two: together with the claim words of the list's entries the step asks for the lists of ALL of them
    // (64 lanes, eight lines); the one of the entry that wins is the next step's list (enext)
    uint2 enext = make_uint2(HARC_NONE, 0u); bool have_next = false;
    // with the lists the reads a walk has taken reach the LDS table only when something asks the table (the probes' bin scans, the look-ahead seeds):
    // a step by list compares its candidate with the wave's register copy (lane t: the read of step t) -- no LDS round trip in such a step
    const bool ownlate = FASTP && s.succ != nullptr;
    int own_ins = T0;                                             // steps [T0, own_ins) of this launch are in the table
    auto own_sync = [&](int upto) {
        if (lane >= own_ins && lane < upto) {
            uint32_t hh = (ownreg * 0x9E3779B1u) >> 25;
            while (atomicCAS(&ownt[hh], HARC_NONE, ownreg) != HARC_NONE) hh = (hh + 1u) & (HARC_OWN_SLOTS - 1u);
        }
        own_ins = upto;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // ... and a look-ahead seed taken by such a walk does not fetch its read either: the counts become the seed's when something asks for them (seedrid)
    uint32_t seedrid = HARC_NONE;
    auto counts_sync = [&]() {
        if (seedrid == HARC_NONE) return;
        if (lane < NW) rdl[lane] = reinterpret_cast<const uint32_t *>(s.reads)[(size_t)seedrid * NW + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cons_reset_lds(st, rdl, L, lane);
        seedrid = HARC_NONE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (FASTP && s.succ && h.mode == 2) lastrid = h.cur;
    auto rows_materialise = [&]() {
        if (lane < NW) rdl[lane] = reinterpret_cast<const uint32_t *>(s.reads)[(size_t)lastrid * NW + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rows_from_read<W>(rdl, L, lastdir, lane, rowF, rowR);
        rows_ok = true; rows_lazy = false;
    };
    PH(0);
    for (int t = T0; t < s.S; t++) {
        uint32_t found = HARC_NONE; int fj = 0, fdir = 0, fhd = -1;   // fhd: Hamming distance of the accepted read where the scan that found it knows it
        bool viafast = false, nohit = false;
        if constexpr (FASTP) {
            if (s.succ && lastrid != HARC_NONE && !(t == 0 && resume > 0)) {
                uint2 e = make_uint2(HARC_NONE, 0u);
                if (have_next) e = enext;
                else if (lane < HARC_SUCC_N) e = s.succ[((size_t)lastrid * 2 + (size_t)lastdir) * HARC_SUCC_N + lane];
                have_next = false;
                static_assert(HARC_SUCC_N == 8, "lane >> 3 = entry, lane & 7 = entry of its list");
                const uint32_t ck = (uint32_t)__shfl((int)e.x, lane >> 3, 64), cy = (uint32_t)__shfl((int)e.y, lane >> 3, 64);
                uint2 e2 = make_uint2(HARC_NONE, 0u);
                if (ck != HARC_NONE) e2 = s.succ[((size_t)ck * 2 + (size_t)((cy >> 8) & 1u)) * HARC_SUCC_N + (lane & 7)];
                bool ok = false;
                if (e.x != HARC_NONE) ok = !((s.claimed[e.x >> 6] >> (e.x & 63)) & 1ULL);
                unsigned long long m = __ballot(ok);
                while (m) {                                            // ... and not taken by this walk since the bitmap was frozen (nearly always the first one is not)
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)e.x, __ffsll((long long)m) - 1);
                    if (!__ballot(ownreg == x)) break;
                    m &= m - 1ULL;
                }
                if (m) {
                    const int f = __ffsll((long long)m) - 1;
                    enext.x = (uint32_t)__shfl((int)e2.x, 8 * f + (lane & 7), 64); enext.y = (uint32_t)__shfl((int)e2.y, 8 * f + (lane & 7), 64);
                    if (lane >= HARC_SUCC_N) enext = make_uint2(HARC_NONE, 0u);
                    found = (uint32_t)__builtin_amdgcn_readlane((int)e.x, f);
                    const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)e.y, f);
                    fj = (int)(meta & 0xFFu); fdir = (int)((meta >> 8) & 1u); fhd = (int)((meta >> 9) & 0x7Fu);
                    const int pp = (int)((meta >> 16) & 0xFFFu);
                    nuse += (uint32_t)(pp + 1); lastp += 4 * pp - (lastp >> 2);
                    if (lane == 0) { nc++; ncu++; }
                    viafast = true; have_next = fhd == 0;
                } else if (!((uint32_t)__builtin_amdgcn_readlane((int)e.y, 0) & HARC_SUCC_OPEN)) nohit = true;     // the whole list is taken and nothing lies behind it
            }
        }
        if (!viafast && !nohit) {
        if (ownlate && own_ins < t) own_sync(t);
        if (ownlate) counts_sync();
        if (!rows_ok) { if (FASTP && rows_lazy) rows_materialise(); else cons_rows(st, L, lane, coltmp, rowF, rowR); }   // consensus and its reverse complement -> the wave's window rows
        PH(1);
        // Probes are issued in priority order in batches: every probe behind the first hit of a batch is speculative traffic,
        // every extra batch is a serial round trip to HBM.
        int base = (t == 0 && resume > 0) ? resume : 0;          // the probes before `resume` were made in an earlier super-round: nothing then, nothing now
        int stepbig = 0;                                         // COOP: probes into large live bins made by this STEP so far
        for (int bi = 0; base < s.nprobe; bi++) {
            // first batch: twice the running mean of the priority index of the chain's hits + 16 (high coverage -> hits at small shifts ->
            // narrow first batch; the mean, not the last hit: where reads start is Poisson, the last hit says little about the next),
            // at most firstmax probes; every later batch is a full wave.  A probe behind the winner is speculation: without the bitmap in
            // front of the tables it cost a table fetch at the random-access ceiling and 32 was the best cap; with the bitmap it costs a
            // 4-byte lookup (48: configs[2] chains 768 -> 728 ms); with the bitmap's lines chosen by minimizer it shares the lines of the
            // useful probes (64).  With few chains a round trip costs more than the probes (QUAD: 64).
            int bend;
            { int w0 = bi == 0 ? ((HARC_W0_MUL * (lastp >> 4) + HARC_W0_ADD + 15) & ~15) : 64; const int wmax = (QUAD || SPEC) ? 64 : (bi <= 1 ? s.firstmax : 64); if (w0 > wmax) w0 = wmax; bend = base + w0; }
            if (bend > s.nprobe) bend = s.nprobe;
            if (bend <= base) bend = s.nprobe;
            const int p = base + lane; dbg_batches++;
            uint32_t mine = HARC_NONE; int j = 0, dir = 0, l = 0, myhd = 0;
            uint32_t ncb = 0;                                             // candidates this lane tests in this batch
            bool big = false; uint32_t b_cnt = 0; uint64_t b_slot = 0;   // a bin too large for one lane: scanned by the whole wave below
            uint2 b_lt = make_uint2(0, 0);                               // COOP: its row of largetab, fetched by the lane that found it (all bins of the batch in ONE round trip)
            bool cand = false; uint32_t c_sst = 0, c_cw = 0, c_piy = 0; uint64_t c_slot = 0;     // SEQ (see "ONE AFTER THE OTHER" below): the small bin this lane's probe found
            uint32_t mrd[NW];                                             // !SEQ: only the lane that wins has loaded them, and only it reads them
            if (p < bend && cap) {
                const uint2 pi = s_pinfo[p];
                j = (int)(pi.x >> 16); dir = (int)((pi.x >> 13) & 1);
                l = (int)((pi.x >> 14) & 1);
                uint64_t key;
                {   // key = kbits bits of the (reverse) consensus at bit 2 (ds +- j)
                    const int i0 = (int)((pi.x & 0x1FFF) >> 5), shb = (int)(pi.x & 31);
                    const uint32_t d0 = rowF[i0], d1 = rowF[i0 + 1], d2 = rowF[i0 + 2];
                    key = (uint64_t)__builtin_amdgcn_alignbit(d1, d0, shb) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, shb) << 32);
                    if constexpr (!SPEC) key &= (l ? kmask1 : kmask0);
                }
                HashSlot *const tab = l ? s.slots[1] : s.slots[0];
                // The table is bucketed (64 B = 4 slots).  QUAD (few chains, latency-bound): the whole bucket in one round trip; otherwise
                // (many chains) two slots at a time -- the second pair is the same 64-B sector.  A full bucket ends the
                // search unless its overflow flag says that keys went on to the next one.
                // phase 1: every lane finishes its slot search (the dependent bucket fetches of all lanes overlap) ...
                const uint64_t hk = key_scramble(key);                    // what the table stores and compares
                uint64_t sl;
                if constexpr (SPEC) sl = (uint64_t)__umulhi((uint32_t)(hk >> 32), (uint32_t)(cap >> 2)) << 2; else sl = bucket_slot(hk, cap);
                int state = 0, qhit = 0;                                  // 1 = the key is not in the table, 2 = key found
                uint32_t sst = 0, cw = 0;
                if (s.bloom_lines) {                                      // most keys of a step are in neither: they stop at the bitmap
                    uint32_t bw, bm;
                    if constexpr (SPEC) bloom_pos(key, hk, s.bloom_lines, 17, 0xFFFFFFFFu, &bw, &bm);
                    else bloom_pos(key, hk, s.bloom_lines, s.bloom_nwin[0], s.bloom_mmask, &bw, &bm);      // both dictionaries have keys of the same width when nwin > 0
                    if (((l ? s.bloom[1] : s.bloom[0])[bw] & bm) != bm) state = 1;
                }
                if (state == 0) for (;;) {
                    constexpr int NQ = QUAD ? 4 : 2;
                    uint32_t w0 = 0;
#pragma unroll
                    for (int hp = 0; hp < 4 / NQ; hp++) {
                        if (state == 0) {
                            uint4 rawq[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; q++) rawq[q] = *reinterpret_cast<const uint4 *>(&tab[sl + hp * NQ + q]);
                            if (hp == 0) w0 = rawq[0].w;
#pragma unroll
                            for (int q = 0; q < NQ; q++) {
                                if (state == 0) {
                                    np++;
                                    if (rawq[q].w == 0) state = 1;
                                    else if (rawq[q].x == (uint32_t)hk && rawq[q].y == (uint32_t)(hk >> 32)) { state = 2; qhit = hp * NQ + q; sst = rawq[q].z; cw = rawq[q].w; }
                                }
                            }
                        }
                    }
                    if (state == 0 && !(w0 & SLOT_OVF)) state = 1;
                    if (state) break;
                    sl += 4; if (sl >= cap) sl = 0;
                }
                // ... phase 2: the lanes that found their key.  HARC_SEQ_SCAN (default): they only say so -- the bins are looked at below, one after the
                // other in priority order, by the whole wave.  Otherwise (round 2) every such lane scans its bin itself, all at once.
                if (state == 2 && !(cw & SLOT_DEAD)) {                    // SLOT_DEAD: every read of this bin is already claimed
                    // bins of more than HARC_LARGEBIN reads (at build time) are scanned by the whole wave, and only by the COOP kernel
                    if (cw & SLOT_BIG) { big = true; b_cnt = cw & (SLOT_CNT_MASK | SLOT_OVF | SLOT_BIG); b_slot = sl + qhit; if (COOP) b_lt = s.largetab[sst]; }
                    else if constexpr (SEQ) {
                        cand = true; c_sst = sst; c_cw = cw; c_piy = pi.y; c_slot = sl + qhit;
                        if (cw & SLOT_EMB) {
                            // single-read bins (nearly all): claimed reads and the chain's own reads of this super-round are weeded out HERE, by all
                            // lanes at once -- tested one after the other below, each of the chain's last few reads (they sit at the small shifts,
                            // in front of the read the step is looking for) would cost a dependent round trip of its own
                            // -- but the look at the claim bitmap is a round trip to memory of its own, in front of the one that fetches the read: it is
                            // made only when the batch found at least `weedmin` such bins (one or two are tested faster than they are weeded: the test
                            // asks for the claim word anyway).  __ballot here counts the lanes that are in this branch.
                            if (__popcll(__ballot(true)) >= s.weedmin && ((reinterpret_cast<const uint32_t *>(s.claimed)[sst >> 5] >> (sst & 31u)) & 1u)) { cand = false; atomicOr(reinterpret_cast<uint32_t *>(&tab[sl + qhit]) + 3, SLOT_DEAD); }
                            else if (OWNT && own_has(ownt, sst)) cand = false;
                        }
                    }
                    else {
                        const uint32_t cntb = cw & SLOT_CNT_MASK;
                        const bool emb = (cw & SLOT_EMB) != 0;                // single-read bin: `start` IS the read id
                        const uint32_t *const ids = l ? s.ids[1] : s.ids[0];
                        const uint32_t *const mrow = s_mask + (pi.y >> 16);
                        const int bitoff = (int)(pi.y & 0xFFFF);
                        // at most HARC_LARGEBIN reads: the maxsearch window (reorder.cpp:540) cannot close unless maxsearch itself is that small
                        const bool exactwin = !SPEC && s.maxsearch < (int)HARC_LARGEBIN;
                        uint32_t lead = 0; bool alltop = true; int seen = 0;
                        for (uint32_t i = cntb; i > 0 && seen < s.maxsearch; i--) {
                            const uint32_t rid = emb ? sst : ids[sst + i - 1];
                            // claim bit and read words are fetched together (one dependent hop instead of two)
                            const unsigned long long cwd = s.claimed[rid >> 6];
                            load_read32<W>(s.reads, rid, mrd);
                            if ((cwd >> (rid & 63)) & 1ULL) { if (alltop) lead++; continue; }
                            alltop = false;
                            const int hd = ham_window<W>(rowF, bitoff, mrow, mrd);
                            // taken by this chain earlier in this super-round? (not in the frozen bitmap).  Such a read is not a candidate at
                            // all (it does not count); asked only when the answer matters
                            bool own = false;
                            if (exactwin || hd <= s.thresh) {
                                if constexpr (OWNT) own = own_has(ownt, rid);
                                else for (int k = 0; k < t; k++) own |= ((uint32_t)__builtin_amdgcn_readlane((int)ownreg, k) == rid);
                            }
                            if (own) continue;
                            nc++; ncb++; if (exactwin) seen++;
                            if (hd <= s.thresh) { mine = rid; myhd = hd; break; }
                        }
                        // hints only (the claim bitmap stays the truth): claimed reads at the top of a bin are never looked at again
                        if (lead) {
                            uint32_t *cp = reinterpret_cast<uint32_t *>(&tab[sl + qhit]) + 3;
                            if (lead == cntb) atomicOr(cp, SLOT_DEAD); else if (!emb) atomicMin(cp, (cntb - lead) | (cw & SLOT_OVF));
                        }
                    }
                }
            }
            int winlane = 64;
            if constexpr (SEQ) {
            // ---- the small bins the probes of this batch found, ONE AFTER THE OTHER in priority order (the sequential order of reorder.cpp:517-552),
            //      every candidate tested by the whole wave: NW lanes fetch one dword of the read each and take their share of the Hamming
            //      distance, the claim word and the bin's ids are wave-uniform loads, the chain's own reads one compare across the lanes.  A step
            //      tests one candidate on average (C-bar ~ 1) -- lane-serial scans of every bin at once kept the whole wave busy with the ~85
            //      vector instructions of a scan for the two or three lanes that had a bin, and cost 8 registers per lane for the read.
            {
                unsigned long long mc = __ballot(cand);
                if (!COOP) { const unsigned long long mb = __ballot(big); if (mb) mc &= (mb & (0ULL - mb)) - 1ULL; }      // behind a large bin the main kernel stops anyway
                const bool exactwin = !SPEC && s.maxsearch < (int)HARC_LARGEBIN;      // else the maxsearch window (reorder.cpp:540) cannot close inside a small bin
                uint32_t ntest = 0;
                while (mc) {
                    const int w = __ffsll((long long)mc) - 1;
                    mc &= mc - 1ULL;
                    const uint32_t o_sst = (uint32_t)__builtin_amdgcn_readlane((int)c_sst, w), o_cw = (uint32_t)__builtin_amdgcn_readlane((int)c_cw, w);
                    const uint32_t o_piy = (uint32_t)__builtin_amdgcn_readlane((int)c_piy, w);
                    const int o_l = __builtin_amdgcn_readlane(l, w);
                    const uint32_t cntb = o_cw & SLOT_CNT_MASK;
                    const bool emb = (o_cw & SLOT_EMB) != 0;                  // single-read bin: `start` IS the read id
                    const uint32_t *const ids = o_l ? s.ids[1] : s.ids[0];
                    const int bitoff = (int)(o_piy & 0xFFFF), i0 = bitoff >> 5, sh = bitoff & 31;
                    const uint32_t *const mrow = s_mask + (o_piy >> 16);
                    const uint64_t o_slot = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c_slot, w) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(c_slot >> 32), w) << 32);
                    uint32_t lead = 0, hit = HARC_NONE, rd = 0; bool alltop = true; int seen = 0, hit_hd = -1;
                    for (uint32_t i = cntb; i > 0 && seen < s.maxsearch; i--) {
                        const uint32_t rid = emb ? o_sst : (uint32_t)__builtin_amdgcn_readfirstlane((int)ids[o_sst + i - 1]);
                        // taken by this chain earlier in this super-round? (not in the frozen bitmap: the read the consensus came from sits in the first bins of
                        // every step).  Such a read is not a candidate at all; asked of the wave's registers BEFORE anything is fetched for it
                        if (!OWNT && __ballot(ownreg == rid && lane < t)) { alltop = false; continue; }
                        // claim bit and read words are fetched together (one dependent hop instead of two)
                        const uint32_t cwd = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t *>(s.claimed)[rid >> 5]);
                        rd = lane < NW ? reinterpret_cast<const uint32_t *>(s.reads)[(size_t)rid * NW + lane] : 0u;
                        if ((cwd >> (rid & 31u)) & 1u) { if (alltop) lead++; continue; }
                        alltop = false;
                        if (OWNT && __ballot(ownreg == rid && lane < t)) continue;     // (with the LDS table the lanes have weeded most of these out already)
                        uint32_t hdp = 0;
                        if (lane < NW) hdp = (uint32_t)__popc((__builtin_amdgcn_alignbit(rowF[i0 + lane + 1], rowF[i0 + lane], sh) ^ rd) & mrow[lane]);
                        // NW <= 16 lanes: the sum over a row of 16 (row_shr 8, 4, 2, 1: lane 15 holds it)
                        hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x118, 0xF, 0xF, true);
                        hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x114, 0xF, 0xF, true);
                        hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x112, 0xF, 0xF, true);
                        hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x111, 0xF, 0xF, true);
                        const int hd = __builtin_amdgcn_readlane((int)hdp, 15);
                        ntest++; if (exactwin) seen++;
                        if (hd <= s.thresh) { hit = rid; hit_hd = hd; break; }
                    }
                    // hints only (the claim bitmap stays the truth): claimed reads at the top of a bin are never looked at again
                    if (lead && lane == 0) {
                        uint32_t *cp = reinterpret_cast<uint32_t *>(&(o_l ? s.slots[1] : s.slots[0])[o_slot]) + 3;
                        if (lead == cntb) atomicOr(cp, SLOT_DEAD); else if (!emb) atomicMin(cp, (cntb - lead) | (o_cw & SLOT_OVF));
                    }
                    if (hit != HARC_NONE) {
                        winlane = w; found = hit; fj = __builtin_amdgcn_readlane(j, w); fdir = __builtin_amdgcn_readlane(dir, w); fhd = hit_hd;
                        if (lane < NW) rdl[lane] = rd;                         // the accepted read, for updaterefcount
                        break;
                    }
                }
                if (lane == 0) { nc += ntest; ncu += ntest; }
            }
            } else {
            // ---- the best hit of the lanes that scanned their (small) bins themselves
                const unsigned long long msmall = __ballot(mine != HARC_NONE);
                if (msmall) {
                    winlane = __ffsll((long long)msmall) - 1;
                    found = (uint32_t)__builtin_amdgcn_readlane((int)mine, winlane); fj = __builtin_amdgcn_readlane(j, winlane); fdir = __builtin_amdgcn_readlane(dir, winlane);
                    fhd = __builtin_amdgcn_readlane(myhd, winlane);
                    if (lane == winlane) {
#pragma unroll
                        for (int k = 0; k < NW; k++) rdl[k] = mrd[k];
                    }
                }
            }
            // ---- bins of more than HARC_LARGEBIN reads (low-complexity k-mers, repeats).  Schedule rule (the oracle has the same
            //      one): the probes a walk makes into such bins -- not yet exhausted, up to and including the winning probe of the step --
            //      are counted, and the walk ends after the step in which they reach HARC_SCAN_BUDGET (a boundary of a low-complexity run
            //      costs dozens of such probes per step, a run of duplicates one).  The main kernel stops in front of the first such step
            //      (CH_COOP); k_steps<W, QUAD, COOP = true> then goes on with the walk, the whole wave scanning the bins: 64 candidates per round trip from the bin-ordered copy of
            //      their reads (k_large_fill), in the same order and with the same maxsearch window as a lane-serial scan, and the probes
            //      of the step that look into the SAME bin (every shift of a poly-A consensus has the same key) share one pass over it.
            PH(2);
            {
                const unsigned long long bigall = __ballot(big);
                unsigned long long bigm = bigall;
                if (winlane < 64) bigm &= (1ULL << winlane) - 1ULL;
                if (!COOP) { if (bigm) { defer = true; break; } }
                // the probe at which the step reaches HARC_STEP_CAP probes into large live bins: if nothing is found up to and including
                // it, the step is put off there (what lies behind it is not looked at: the oracle has not either)
                int caplane = 64;
                if (COOP) {
                    const int need = s.stepcap - stepbig;
                    if (need >= 1 && __popcll(bigall) >= need) { unsigned long long m = bigall; for (int k = 1; k < need; k++) m &= m - 1; caplane = __ffsll((long long)m) - 1; }
                    if (caplane < 64 && caplane < winlane) bigm &= (2ULL << caplane) - 1ULL;
                }
                if (COOP) while (bigm) {
                    const int bl = __ffsll((long long)bigm) - 1;
                    const uint32_t o_raw = (uint32_t)__builtin_amdgcn_readlane((int)b_cnt, bl), o_cnt = o_raw & SLOT_CNT_MASK;
                    const int o_l = __builtin_amdgcn_readlane(l, bl);
                    const uint64_t o_slot = shfl_u64(b_slot, bl);
                    unsigned long long grp = __ballot(big && l == o_l && b_slot == o_slot) & bigm;     // the probes into this bin, bl among them
                    bigm &= ~grp;
                    dbg_bins++;
#ifdef HARC_COOP_TRACE
                    const int dbg_kind = o_cnt <= 64 ? 0 : o_cnt <= 256 ? 1 : o_cnt <= 1024 ? 2 : 3;
                    const long long dbg_s0 = s.dbg ? wall_clock64() : 0;
#endif
                    const uint2 lt = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)b_lt.x, bl), (uint32_t)__builtin_amdgcn_readlane((int)b_lt.y, bl));
                    // while the bin fits the maxsearch window (reorder.cpp:540) the window never closes: the claim bit is only asked of the
                    // candidates that pass the Hamming test.  Above it the unclaimed reads are counted exactly, as the lane-serial scan would.
                    cmd->mj[lane] = (uint8_t)j; cmd->mdir[lane] = (uint8_t)dir; cmd->own[lane] = ownreg;
                    if (lane == 0) {
                        cmd->op = 1; cmd->l = o_l; cmd->t = t; cmd->fast = o_cnt <= (uint32_t)s.maxsearch ? 1 : 0;
                        cmd->ids0 = lt.x; cmd->m0 = lt.y; cmd->cnt = o_cnt; cmd->grp = grp; cmd->found = HARC_NONE;
                        { const uint32_t m = o_cnt < (uint32_t)s.maxsearch ? o_cnt : (uint32_t)s.maxsearch; cmd->rot = m ? ((c * 0x9E3779B1u) >> 8) % m : 0u; }
                    }
                    __syncthreads();                                           // the helpers start
                    const uint32_t *const idp[2] = { s.ids[0], s.ids[1] };
                    const WgResult wr = (HARC_SCAN_SKETCH && NWV == 1) ? wg_scan_sk<W, NWV>(cmd, 0, lane, idp, s.mirror, rowF, s_mask, rdl, s.maxsearch, s.maxmatch, s.thresh)
                                                         : wg_scan<W, NWV>(cmd, 0, lane, idp, s.mirror, rowF, s_mask, rdl, s.maxsearch, s.maxmatch, s.thresh);
                    dbg_iter += wr.iters; dbg_surv += wr.tests; nc += wr.nc; ncu += wr.nc;
#ifdef HARC_SKETCH_STATS
                    if (s.dbg) { const uint32_t a1 = wave_sum_u32(wr.rej1), a2 = wave_sum_u32(wr.rej2), a3 = wave_sum_u32(wr.acc), a0 = wave_sum_u32(wr.nc); if (lane == 0) { atomicAdd(&s.dbg[44], (unsigned long long)a0); atomicAdd(&s.dbg[45], (unsigned long long)a1); atomicAdd(&s.dbg[46], (unsigned long long)a2); atomicAdd(&s.dbg[47], (unsigned long long)a3); } }
#endif
                    const int besthit = wr.besthit;
                    if (besthit < 64) { found = cmd->found; fj = wr.fj; fdir = wr.fdir; fhd = -1; }     // the helpers' scan does not report the distance
                    if (besthit < 64) { winlane = besthit; bigm &= (1ULL << besthit) - 1ULL; }
#ifdef HARC_COOP_TRACE
                    if (s.dbg) { dbg_kt[dbg_kind] += (unsigned long long)(wall_clock64() - dbg_s0); dbg_kn[dbg_kind]++; }
#endif
                }
                if (COOP && caplane < 64 && caplane < winlane) {               // nothing up to the cap: the step is put off
                    stalled = true; resume_next = base + caplane + 1;
                    break;
                }
                if (COOP) { const int nb = __popcll(bigall & (winlane < 64 ? ((2ULL << winlane) - 1ULL) : ~0ULL)); bigprobes += nb; stepbig += nb; }   // the probes up to and including the winning one
            }
            PH(3);
            if (lane <= winlane) ncu += ncb;                              // lanes behind the winner were speculation (winlane = 64: no hit, all count)
            if (found != HARC_NONE) {
                nuse += (uint32_t)(base + winlane + 1);
                lastp += 4 * (base + winlane) - (lastp >> 2);
                break;
            }
            base = bend;
        }
        }
        if (defer || stalled) break;
        if (found == HARC_NONE) {
            // no candidate: go on from the chain's look-ahead seeds (highest id first, skipping what was claimed meanwhile) -- the new
            // seed of reorder.cpp:652-668 without waiting for the next k_reseed
            nuse += (uint32_t)s.nprobe; dbg_miss++;
            if (ownlate && own_ins < t) own_sync(t);
            uint32_t sid = HARC_NONE;
            if (spos < nsugg) {
                const int idx = spos + lane;
                uint32_t id = 0; bool okc = false;
                if (idx < nsugg) {
                    id = __hip_atomic_load(&s.sugg[(size_t)c * s.nsugg_stride + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // may have been written by this wave above
                    okc = !((s.claimed[id >> 6] >> (id & 63)) & 1ULL);
                    if constexpr (OWNT) okc = okc && !own_has(ownt, id);
                    else for (int k = 0; k < t; k++) okc = okc && ((uint32_t)__builtin_amdgcn_readlane((int)ownreg, k) != id);
                }
                const unsigned long long sm = __ballot(okc);
                if (sm) { const int f = __ffsll((long long)sm) - 1; sid = __shfl(id, f, 64); spos += f + 1; } else spos = nsugg;
            }
            if (sid == HARC_NONE) { needseed = true; break; }
            if (lane == 0) s.steps[(size_t)c * 64 + t] = make_uint2(sid, (1u << 16) | ((uint32_t)spos << 24));     // (the bid: after the walk, below)
            if (lane == t) ownreg = sid;
            if constexpr (OWNT) { if (!ownlate && lane == 0) own_insert(ownt, sid); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
            if (ownlate) { seedrid = sid; lastrid = sid; lastdir = 0; rows_lazy = true; }      // rows and counts from the read, when somebody wants them
            else {
                if (lane < NW) { const uint32_t *rp = reinterpret_cast<const uint32_t *>(s.reads + (size_t)sid * W); rdl[lane] = rp[lane]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                cons_reset_lds(st, rdl, L, lane);                  // every count is replaced: what was pending is gone with the old consensus
            }
            rows_ok = false; pend = 0; ptot = 0;
            nst++;
            if (COOP && bigprobes >= s.budget) break;
            continue;
        }
        if (lane == 0) s.steps[(size_t)c * 64 + t] = make_uint2(found, (uint32_t)fj | ((uint32_t)fdir << 8));
        if (lane == t) ownreg = found;
        if constexpr (OWNT) { if (!ownlate && lane == 0) own_insert(ownt, found); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (FASTP && viafast && fhd == 0) {                        // (k_succ) ... and nobody has asked for its rows yet
            ptot += fj;
            if (lane == 0) pshift[pend] = (uint16_t)ptot;
            pend++; rows_ok = false; rows_lazy = true; lastrid = found; lastdir = fdir;
        } else {
            if (FASTP && viafast) {                                // the list's read disagrees with the consensus somewhere: the counts want the rows of the consensus and the read's words after all
                counts_sync();
                if (pend && !rows_ok) rows_materialise();
                if (lane < NW) rdl[lane] = reinterpret_cast<const uint32_t *>(s.reads)[(size_t)found * NW + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (lazy && fhd == 0) {                                // the read agrees with the consensus on the whole overlap: the new consensus is the read
                rows_from_read<W>(rdl, L, fdir, lane, rowF, rowR);
                ptot += fj;
                if (lane == 0) pshift[pend] = (uint16_t)ptot;
                pend++; rows_ok = true;
                if (FASTP && s.succ) { lastrid = found; lastdir = fdir; rows_lazy = false; }
            } else {
                if (pend) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); cons_flush(st, pshift, pend, ptot, rowF, L, lane); pend = 0; ptot = 0; }
                cons_update_lds(st, rdl, L, fdir, fj, lane);
                rows_ok = false;
                if (FASTP) { lastrid = HARC_NONE; rows_lazy = false; }
            }
        }
        nst++;
        PH(4);
        if (COOP && bigprobes >= s.budget) break;
    }
    // the bids of the steps walked in this launch, all at once (lane t holds the read of step t): k_resolve is their only reader, and an atomic on a cold
    // line of bid[] in every step sat in front of the next step's first wait for memory (vmcnt counts loads and atomics alike)
    if (lane >= T0 && lane < T0 + nst) atomicMin(&s.bid[ownreg], ((uint32_t)lane << 20) | c);
    if (ownlate) counts_sync();
    if (pend) {
        if (FASTP && rows_lazy) rows_materialise();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); cons_flush(st, pshift, pend, ptot, rowF, L, lane); pend = 0; ptot = 0;
    }
    PH(4);
    if (COOP) { if (lane == 0) cmd->op = 0; __syncthreads(); }      // the walk is over: the helpers leave
    if (nst > 0) { const uint32_t w1 = cons_store(st, B1, L, lane); widebits = (widebits & ~(3u << (2 * (par ^ 1u)))) | (w1 << (2 * (par ^ 1u))); }
    np = wave_sum_u32(np); nc = wave_sum_u32(nc); ncu = wave_sum_u32(ncu);
    PH(5);
#undef PH
#ifdef HARC_COOP_TRACE
    if (COOP && lane == 0 && s.dbg) {
        for (int k = 0; k < 6; k++) atomicAdd(&s.dbg[16 + k], dbg_ph[k]);
        for (int k = 0; k < 4; k++) { atomicAdd(&s.dbg[36 + k], dbg_kn[k]); atomicAdd(&s.dbg[40 + k], dbg_kt[k]); }
        const unsigned long long dt0 = (unsigned long long)(wall_clock64() - dbg_t0);
        atomicAdd(&s.dbg[24 + (dt0 / 2500 < 11 ? dt0 / 2500 : 11)], 1ULL);
    }
#endif
    if (COOP && lane == 0 && s.dbg) { const unsigned long long dt = (unsigned long long)(wall_clock64() - dbg_t0); atomicAdd(&s.dbg[9], 1ULL); atomicAdd(&s.dbg[10], dt); atomicMax(&s.dbg[11], dt); atomicAdd(&s.dbg[12], (unsigned long long)(nst)); }
    if (!COOP && lane == 0 && s.dbg) { atomicAdd(&s.dbg[13], 1ULL); if (defer) atomicAdd(&s.dbg[15], 1ULL); if (defer && nst == 0) atomicAdd(&s.dbg[14], 1ULL); }      // trace: walks of the main kernel; those that end in front of a large bin; of them, without a step
    if (lane == 0 && s.dbg) { atomicAdd(&s.dbg[0], (unsigned long long)dbg_bins); atomicAdd(&s.dbg[1], (unsigned long long)dbg_iter); atomicAdd(&s.dbg[2], (unsigned long long)dbg_miss); atomicAdd(&s.dbg[3], (unsigned long long)dbg_surv); atomicAdd(&s.dbg[4], (unsigned long long)nst); atomicAdd(&s.dbg[5], (unsigned long long)dbg_batches); atomicMax(&s.dbg[6], (unsigned long long)dbg_iter); atomicMax(&s.dbg[7], (unsigned long long)dbg_surv); atomicMax(&s.dbg[8], (unsigned long long)dbg_bins); }
    if (lane == 0) {
        if (COOP) { atomicAdd(&s.cstat_coop[c].x, np); atomicAdd(&s.cstat_coop[c].y, nc); atomicAdd(&s.cstat_coop[c].z, nuse); atomicAdd(&s.cstat_coop[c].w, ncu); s.csteps[c].y += (uint32_t)nst; }   // the helpers add to it too
        else { cst.x += np; cst.y += nc; cst.z += nuse; cst.w += ncu; s.cstat[c] = cst; s.csteps[c].x += (uint32_t)nst; }
        h.mode = 0;
        h.nsteps = (h.nsteps & 0xFFFF0000u) | (uint32_t)(T0 + nst);
        h.pad0 = ((uint32_t)lastp & 0xFFFFu) | (((uint32_t)spos & 0xFFu) << 16) | (widebits << 24);
        h.flags = needseed ? (h.flags | CH_NEEDSEED) : (h.flags & ~CH_NEEDSEED);
        h.flags = defer ? (h.flags | CH_COOP) : (h.flags & ~CH_COOP);
        // a step put off (COOP), or one still waiting for its turn in the cooperative kernel (main kernel, nothing walked yet), keeps its place
        h.flags = (h.flags & 0xFFFFu) | ((COOP ? (stalled ? (uint32_t)resume_next : 0u) : ((defer && nst == 0) ? (uint32_t)resume : 0u)) << 16);
        if (!COOP && defer) atomicAdd(&s.coopcnt[c & (HARC_COOPCNT - 1)], 1ULL);
        s.hdr[c] = h;
    }
}

// k_steps_grp (steps_group.h): the dense SPEC kernel with two to four chains per wave, round 5's structural attempt -- byte-identical, 1001 us per launch
// against k_steps' 992 and 18 % slower at 1 % errors.  Not part of the default build since round 6 (`make GRP=1` compiles it in, HARC_AMD_GRP=1 / 2 then
// selects it, and the tests that ask for it run; harc_amd_build_has("grp") tells).
#ifdef HARC_AMD_WITH_GRP
#include "steps_group.h"
#endif

// ---- design (R): replicate the reads and the index, partition the chains (harc_amd_replicate_exchange).  Everything a super-round
// changes outside the walking wave's own state is either recomputed identically on every rank (k_resolve, k_reseed, the compaction of the
// large bins: functions of replicated state) or travels with the all-gather of the walked steps: the chain header and the steps of the
// super-round (k_pack_chains -> all-to-all(v) -> k_unpack_chains, which also places the bids of the other ranks' chains).  The column
// counts, the statistics and the hints in the table slots stay with the owner.
__device__ __forceinline__ bool chain_owned(uint32_t c, uint32_t mod, uint32_t rem) { return mod <= 1 || ((c >> 2) % mod) == rem; }
// the seed a chain takes at the top of k_steps (reorder.cpp:650-688), on the ranks that do NOT walk it: same records, same counters
__global__ void k_apply_seed(S1Args s)
{
    const uint32_t c = harc_gid32();
    if (c >= s.K || chain_owned(c, s.own_mod, s.own_rem)) return;
    ChainHdr h = s.hdr[c];
    if (!(h.flags & CH_ACTIVE) || !s.need[c]) return;
    const uint32_t r = s.needrank[c], R = s.rmeta[0], assigned = s.rmeta[1], got = s.rmeta[2];
    if (h.flags & CH_PREVUNM) {                                   // previous seed found nothing: singleton (reorder.cpp:672-684)
        s.slog[h.prev] = make_uint2(c, h.n_sing);
        h.n_sing++;
    }
    if (r < assigned) {
        const uint32_t id = s.seedbuf[r];
        h.cur = id; h.prev = id; h.flags = ((h.flags | CH_PREVUNM) & ~CH_NEEDSEED) & 0xFFFFu; h.mode = 2;
        const uint32_t first = r * (uint32_t)s.nsugg_per_seed;
        const uint32_t ng = first >= got ? 0u : (got - first < (uint32_t)s.nsugg_per_seed ? got - first : (uint32_t)s.nsugg_per_seed);
        for (uint32_t k = 0; k < ng; k++) s.sugg[(size_t)c * s.nsugg_stride + k] = s.seedbuf[R + first + k];
        h.nsteps = ng << 24;
        uint2 q = s.cst2[c]; q.x++; s.cst2[c] = q;
    } else {                                                      // no reads left (reorder.cpp:670-677)
        h.flags &= ~(CH_ACTIVE | CH_PREVUNM | CH_NEEDSEED);
        atomicAdd(&s.stats[ST_ACTIVE], ~0ULL);
    }
    s.need[c] = 0;
    s.hdr[c] = h;                                                 // until the owner's header arrives with the all-gather
}
// record of a chain: its header (8 dwords) + the S steps of the super-round (2 dwords each).  Slot li of rank p's block = chain
// ((li >> 2) * own_mod + p) * 4 + (li & 3).  One thread per dword.
__global__ void k_pack_chains(S1Args s, uint32_t *buf, uint32_t nper)
{
    const uint32_t recw = 8u + 2u * (uint32_t)s.S;
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)nper * recw) return;
    const uint32_t li = (uint32_t)(gid / recw), d = (uint32_t)(gid % recw);
    const uint32_t c = ((li >> 2) * s.own_mod + s.own_rem) * 4u + (li & 3u);
    uint32_t v = 0;
    if (c < s.K) v = d < 8u ? reinterpret_cast<const uint32_t *>(&s.hdr[c])[d] : reinterpret_cast<const uint32_t *>(&s.steps[(size_t)c * 64])[d - 8u];
    buf[gid] = v;
}
__global__ void k_unpack_chains(S1Args s, const uint32_t *buf, uint32_t nper)
{
    const uint32_t recw = 8u + 2u * (uint32_t)s.S, p = blockIdx.y;
    if (p == s.own_rem) return;                                   // this rank's own chains are in place
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (blockIdx.y is the peer; K * (8 + 2 S) dwords < 2^24)
    if (gid >= (uint64_t)nper * recw) return;
    const uint32_t li = (uint32_t)(gid / recw), d = (uint32_t)(gid % recw);
    const uint32_t c = ((li >> 2) * s.own_mod + p) * 4u + (li & 3u);
    if (c >= s.K) return;
    const uint32_t *rec = buf + ((size_t)p * nper + li) * recw;
    const uint32_t v = rec[d];
    if (d < 8u) reinterpret_cast<uint32_t *>(&s.hdr[c])[d] = v;
    else {
        reinterpret_cast<uint32_t *>(&s.steps[(size_t)c * 64])[d - 8u] = v;
        const uint32_t t = (d - 8u) >> 1;
        // the bid the walking wave made for the read of step t (k_steps), on this rank too: k_resolve arbitrates over all of them
        if (((d - 8u) & 1u) == 0 && (rec[2] & CH_ACTIVE) && t < (rec[6] & 0xFFu)) atomicMin(&s.bid[v], (t << 20) | c);
    }
}

// What every rank of a design-(R) run must agree on after a super-round, folded into a few words (sums of per-element mixes, so the order of
// the additions does not matter): [0] claim bitmap [1] chain headers [2] chains that asked for a seed [3] cursor [4..6] what k_reseed handed out.
// The host compares them over the ranks after every batch of super-rounds: replicas that drift apart would otherwise hang in the next
// all-gather or end with a wrong archive.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) v += shfl_u64(v, (__lane_id() + o) & 63); return v; }
__device__ __forceinline__ unsigned long long digest_mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; return x ^ (x >> 33); }
__global__ void k_replica_digest(S1Args s, unsigned long long nwords, unsigned long long *out)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long a = 0, b = 0, d = 0;
    for (unsigned long long i = gid; i < nwords; i += nth) { const unsigned long long w = s.claimed[i]; if (w) a += digest_mix(w ^ (i * 0x9E3779B97F4A7C15ULL)); }
    for (unsigned long long i = gid; i < (unsigned long long)s.K * 8; i += nth) b += digest_mix((unsigned long long)reinterpret_cast<const uint32_t *>(s.hdr)[i] ^ (i * 0x9E3779B97F4A7C15ULL));
    for (unsigned long long i = gid; i < s.K; i += nth) if (s.need[i]) d += digest_mix(i + 1);
    a = wave_sum_u64(a); b = wave_sum_u64(b); d = wave_sum_u64(d);
    if ((threadIdx.x & 63) == 0) { if (a) atomicAdd(&out[0], a); if (b) atomicAdd(&out[1], b); if (d) atomicAdd(&out[2], d); }
    if (gid == 0) { out[3] = (unsigned long long)*s.cursor; out[4] = s.rmeta[0]; out[5] = s.rmeta[1]; out[6] = s.rmeta[2]; }
}

// (B) G lanes per chain (G = 16/32/64 >= S), lane t of the group = step t of the super-round
template <int G> __device__ __forceinline__ void resolve_body(const S1Args &s, const uint32_t bx)
{
    constexpr int CPW = 64 / G;                                   // chains per wave
    const int lane = threadIdx.x & 63, sub = lane / G, sl = lane % G, g0 = sub * G;
    if (bx == 0 && threadIdx.x == 0) { s.reseed_g[0] = 0u; s.reseed_g[1] = 0u; s.grp_ticket[0] = 0u; }      // the meeting counter and the flag of the k_reseed_mg that follows
    const uint32_t c = (bx * 4 + (threadIdx.x >> 6)) * CPW + sub;
    ChainHdr h; h.flags = 0; h.nsteps = 0; h.n_main = 0; h.n_sing = 0; h.prev = 0; h.pad0 = 0;
    if (c < s.K) h = s.hdr[c];
    // back-off: a chain that sat this super-round out is not touched by it (its header keeps the roll-back of the round in which it was cut)
    const uint32_t bo0 = (s.bo && c < s.K) ? s.bo[c] : 0u;
    const bool asleep = (bo0 & 0xFFu) != 0;
    if (asleep && sl == 0) s.bo[c] = bo0 - 1u;
    const bool act = c < s.K && (h.flags & CH_ACTIVE) && !asleep;
    const int n = act ? (int)(h.nsteps & 0xFF) : 0;
    uint2 sp = make_uint2(HARC_NONE, 0);
    bool mineb = false;
    if (sl < n) {
        sp = s.steps[(size_t)c * 64 + sl];
        mineb = s.bid[sp.x] == (((uint32_t)sl << 20) | c);
    }
    unsigned long long lost = __ballot(sl < n && !mineb);
    if (G < 64) lost = (lost >> g0) & ((1ULL << (G & 63)) - 1ULL);
    const int v = lost ? (__ffsll((long long)lost) - 1) : n;     // steps kept: those before the first lost bid
    const bool cut = v < n;
    // what every kept step emits (reorder.cpp:560-578 for a match, :678-687 for a new seed)
    const bool kept = sl < v;
    // a bid held for a step behind the cut is withdrawn; the read of a kept step is claimed below and nobody bids for a claimed read again, so
    // its bid stays where it is (one random write less per kept step: most of them)
    if (mineb && !kept) s.bid[sp.x] = HARC_NONE;
    const bool seedk = kept && ((sp.y >> 16) & 1);
    const uint32_t pkind = __shfl_up((uint32_t)seedk, 1, 64), prid = __shfl_up(sp.x, 1, 64);
    const bool punm = sl == 0 ? ((h.flags & CH_PREVUNM) != 0) : (pkind != 0);      // a seed is pending before this step
    const uint32_t pid = sl == 0 ? h.prev : prid;
    const uint32_t nm = kept ? (seedk ? 0u : (punm ? 2u : 1u)) : 0u;               // main-stream records
    const uint32_t ns = (kept && seedk && punm) ? 1u : 0u;                          // singleton-stream records
    uint32_t tm, ts;
    const uint32_t em = wave_excl_scan_u32(nm, &tm), es = wave_excl_scan_u32(ns, &ts);
    const uint32_t em0 = __shfl(em, g0, 64), es0 = __shfl(es, g0, 64);
    // group totals and the last kept step
    const int lastl = g0 + (v > 0 ? v - 1 : 0);
    const uint32_t gm = __shfl(em + nm, lastl, 64) - em0, gs = __shfl(es + ns, lastl, 64) - es0;
    // the pages the chain's gm new main records go to: its newest page holds record (n_main - 1), new ones come from the counter
    // (pages are handed out PG_CHUNK at a time: one atomic on the shared counter per 512 records of a chain -- one per page was 14 000 atomics on
    // one address per super-round at configs[2] and doubled this kernel's time)
    uint32_t pcur = 0, pnew = 0, qcur = 0, qlast = 0;
    bool wantchunk = false;
    if (act && sl == 0 && v > 0 && gm > 0) {
        pcur = s.pg_cur[c];                                       // first page of the chunk that holds record n_main - 1
        qcur = ((h.n_main ? h.n_main - 1u : 0u) >> 6) / PG_CHUNK; // number of that chunk inside the chain
        qlast = ((h.n_main + gm - 1u) >> 6) / PG_CHUNK;           // at most the next one: 2 S <= 128 records a round, 512 a chunk
        wantchunk = qlast != qcur;
    }
    // The chains of a run march in step: they all fill their chunk of 512 records within a few super-rounds of each other, and every 32nd round or so
    // nearly all 65 536 of them asked the ONE counter for their next chunk in the same launch -- 65 536 atomics on one address, +150 us of this kernel
    // (profiles/r04/ksteps_profile_c3.txt: k_resolve 61 -> 210 us at rounds 32, 64, 96 ..., fading as the chains drift apart).  The workgroup asks once
    // for all its chains (which page a chain gets is not visible in the output: k_pages_out goes by pg_hdr).
    {
        __shared__ uint32_t sh_cnt, sh_base;
        if (__syncthreads_or(wantchunk ? 1 : 0)) {
            if (threadIdx.x == 0) sh_cnt = 0u;
            __syncthreads();
            uint32_t myrank = 0;
            if (wantchunk) myrank = atomicAdd(&sh_cnt, 1u);
            __syncthreads();
            if (threadIdx.x == 0) sh_base = atomicAdd(s.pg_count, sh_cnt * (unsigned int)PG_CHUNK);
            __syncthreads();
            if (wantchunk) {
                pnew = sh_base + myrank * PG_CHUNK;
                for (uint32_t k = 0; k < PG_CHUNK; k++) s.pg_hdr[pnew + k] = make_uint2(c, qlast * PG_CHUNK + k);
                s.pg_cur[c] = pnew;
            }
        }
    }
    pcur = __shfl(pcur, g0, 64); pnew = __shfl(pnew, g0, 64); qcur = __shfl(qcur, g0, 64);
    if (kept) {
        atomicOr(&s.claimed[sp.x >> 6], 1ULL << (sp.x & 63));
        if (seedk) {
            if (punm) s.slog[pid] = make_uint2(c, h.n_sing + (es - es0));
        } else {
            uint32_t seq = h.n_main + (em - em0);
            if (punm) {                                           // the pending seed opens a contig
                const uint32_t pn = seq >> 6, pg = (pn / PG_CHUNK == qcur ? pcur : pnew) + pn % PG_CHUNK;
                s.pg_rec[(size_t)pg * 64 + (seq & 63u)] = make_uint2(pid, (uint32_t)(s.L & 0xFF));
                seq++;
            }
            const uint32_t pn = seq >> 6, pg = (pn / PG_CHUNK == qcur ? pcur : pnew) + pn % PG_CHUNK;
            s.pg_rec[(size_t)pg * 64 + (seq & 63u)] = make_uint2(sp.x, (sp.y & 0xFF) | (1u << 8) | (((sp.y >> 8) & 1u) << 9));
        }
    }
    const uint32_t lastrid = __shfl(sp.x, lastl, 64), lastseed = __shfl((uint32_t)seedk, lastl, 64);
    unsigned long long seedmask = __ballot(seedk);
    if (G < 64) seedmask = (seedmask >> g0) & ((1ULL << (G & 63)) - 1ULL);
    const int nseed = __popcll(seedmask);
    const int lastseedlane = seedmask ? (63 - __clzll((long long)seedmask)) : 0;
    const uint32_t lastsidx = __shfl(sp.y >> 24, g0 + lastseedlane, 64);
    if (act && sl == 0) {
        if (v > 0) {
            h.n_main += gm; h.n_sing += gs;
            h.cur = lastrid;
            if (lastseed) { h.prev = lastrid; h.flags |= CH_PREVUNM; } else h.flags &= ~CH_PREVUNM;
        }
        uint32_t spos = (h.nsteps >> 16) & 0xFF;
        if (cut) { if (nseed) spos = lastsidx; }                 // look-ahead seeds of dropped steps stay available
        else spos = (h.pad0 >> 16) & 0xFF;
        const uint32_t keep = (h.nsteps & 0xFF000000u) | (spos << 16);
        if (cut) {
            h.mode = v > 0 ? 1u : 0u; h.nsteps = keep | ((uint32_t)v << 8); h.flags &= ~CH_NEEDSEED; h.flags &= 0xFFFFu;   // rolled back: a step put off belonged to a state that is gone
        } else {
            if (n > 0) { h.flags ^= CH_PARITY; h.mode = 0; }
            h.nsteps = keep;
        }
        s.hdr[c] = h;
        if (nseed || cut) { uint2 q = s.cst2[c]; q.x += (uint32_t)nseed; q.y += cut ? 1u : 0u; s.cst2[c] = q; }
        s.need[c] = (!cut && (h.flags & CH_NEEDSEED)) ? 1 : 0;
        if (s.bo) {   // a walk cut at a lost bid: the chain sits out 2^(k - HARC_BO_FREE) - 1 super-rounds, k = its cuts in a row (at most HARC_BO_FREE + HARC_BO_CAP); a walk that was kept whole clears k
            uint32_t k = (bo0 >> 8) & 0xFFu, zs = 0u;
            if (cut) { if (k < HARC_BO_FREE + HARC_BO_CAP) k++; zs = k > HARC_BO_FREE ? (1u << (k - HARC_BO_FREE)) - 1u : 0u; }
            else if (n > 0) k = 0;
            s.bo[c] = zs | (k << 8);
        }
    }
}
template <int G> __global__ __launch_bounds__(256) void k_resolve(S1Args s) { resolve_body<G>(s, blockIdx.x); }

// (C) one workgroup: new seeds, in chain order, from the single descending cursor over unclaimed reads (reorder.cpp:650-688);
// when the cursor runs out the remaining chains finish.  Every reseeded chain also gets HARC_NSUGG look-ahead seeds: the next
// unclaimed ids below the cursor (not claimed; the cursor moves below them, so they are this chain's until it takes them or finds
// them taken by a walk), which k_steps uses when the chain is stuck again.
// The kernel only ranks the chains and finds the ids; every chain applies its own seed at the top of the next k_steps.
// NT threads: 1024, or 256 with few chains (the block scans and barriers of sixteen waves were most of the kernel's 11.5 us on a
// 2048-chain input: a tenth of a super-round).  The look-ahead range is HARC_LOOK_CHUNKS x 1024 words either way.
template <int NT> __device__ __forceinline__ void reseed_body(const S1Args &s)
{
    __shared__ uint32_t sm[20];
    __shared__ long long scursor;
    const int t = threadIdx.x;
    // chains per thread, contiguous and a multiple of 8 so that the need bytes are read as 64-bit words (the array is padded with zeros)
    const uint32_t chunk = (((s.K + NT - 1) / NT) + 7) & ~7u;
    const uint32_t c0 = (uint32_t)t * chunk;
    uint32_t mycnt = 0;
    if (c0 < s.K) for (uint32_t c = c0; c < c0 + chunk; c += 8) mycnt += (uint32_t)__popcll(*(const unsigned long long *)(s.need + c));
    uint32_t R; const uint32_t rbase = block_excl_scan_u32<NT>(mycnt, sm, &R);
    if (R == 0) return;
    if (mycnt) {                                                  // rank of every chain that wants a seed, ascending chain id
        uint32_t r = rbase;
        for (uint32_t c = c0; c < c0 + chunk; c += 8) {
            unsigned long long w = *(const unsigned long long *)(s.need + c);
            while (w) { const int b = __ffsll((long long)w) - 1; w &= w - 1; s.needrank[c + (uint32_t)(b >> 3)] = r++; }
        }
    }

    long long cursor = *s.cursor;
    uint32_t assigned = 0;
    while (assigned < R && cursor >= 0) {
        const long long cwd = cursor >> 6, wi = cwd - t;
        unsigned long long bits = 0;
        if (wi >= 0) {
            bits = ~s.claimed[wi];
            if (wi == cwd) { const int top = (int)(cursor & 63); if (top < 63) bits &= (2ULL << top) - 1ULL; }
        }
        uint32_t total; const uint32_t off = block_excl_scan_u32<NT>((uint32_t)__popcll(bits), sm, &total);
        unsigned long long newclaim = 0; uint32_t k = 0;
        while (bits && assigned + off + k < R) {
            const int b = 63 - __clzll((long long)bits);
            bits &= ~(1ULL << b);
            const uint32_t id = (uint32_t)(wi * 64 + b), r = assigned + off + k;
            k++; newclaim |= 1ULL << b;
            s.seedbuf[r] = id;
            if (r == R - 1) scursor = (long long)id - 1;
        }
        if (newclaim) s.claimed[wi] |= newclaim;                  // this workgroup is the only writer of the bitmap in this launch
        __syncthreads();
        if (assigned + total >= R) { assigned = R; cursor = scursor; }
        else { assigned += total; cursor = (cwd - (NT - 1)) * 64 - 1; }
        __syncthreads();
    }
    // look-ahead: the next assigned * HARC_NSUGG unclaimed ids below the cursor; nothing is claimed, the cursor moves below them
    const uint32_t want = assigned * (uint32_t)s.nsugg_per_seed;
    uint32_t got = 0;
    const long long look = cursor;
    __threadfence();
    __syncthreads();
    if (look < 0) got = want;                                     // nothing left below the cursor: no look-ahead (got is reset below)
    // at most HARC_LOOK_CHUNKS chunks of 1024 words below the cursor (bounded cost; the oracle has the same limit).  One chunk was enough
    // while few chains reseed per super-round; a minimizer-bucket shard of a multi-GPU run leaves a third of its reads unmatched, tens
    // of thousands of chains ask for seeds every round, and those that got no look-ahead seed walked ONE read per round: 1296 rounds
    // instead of ~400 on an 8-way shard of configs[2].
    for (int ch = 0; ch < HARC_LOOK_CHUNKS * (1024 / NT) && got < want; ch++) {
        const long long cwd = (look >> 6) - (long long)NT * ch, wi = cwd - t;
        if (cwd < 0) break;
        unsigned long long bits = 0;
        if (wi >= 0) {
            bits = ~s.claimed[wi];
            if (ch == 0 && wi == cwd) { const int top = (int)(look & 63); if (top < 63) bits &= (2ULL << top) - 1ULL; }
        }
        uint32_t total; const uint32_t off = block_excl_scan_u32<NT>((uint32_t)__popcll(bits), sm, &total);
        const uint32_t take = total >= want - got ? want - got : total;
        uint32_t k = 0;
        while (bits && off + k < take) {
            const int b = 63 - __clzll((long long)bits);
            bits &= ~(1ULL << b);
            const uint32_t id = (uint32_t)(wi * 64 + b);
            s.seedbuf[R + got + off + k] = id;
            if (off + k == take - 1) scursor = (long long)id - 1;   // the cursor goes below the last look-ahead seed handed out:
            k++;                                                    // those reads belong to their chains now, later reseeds do not hand them out again
        }
        got += take;
    }
    if (look < 0) got = 0;
    __threadfence();
    __syncthreads();
    if (got) cursor = scursor;
    // the chains take their seeds themselves at the top of the next k_steps (rank -> seedbuf)
    if (t == 0) { s.rmeta[0] = R; s.rmeta[1] = assigned; s.rmeta[2] = got; }
    if (t == 0) *s.cursor = cursor < -1 ? -1 : cursor;
}
template <int NT> __global__ __launch_bounds__(NT) void k_reseed(S1Args s) { reseed_body<NT>(s); }
// (B) + (C) in one launch when ONE workgroup of k_resolve holds every chain (exact mode: one chain; a super-round of a single wave's walk is ~60 us, of which
// the two launches were ten)
template <int G> __global__ __launch_bounds__(256) void k_resolve_reseed(S1Args s)
{
    resolve_body<G>(s, 0u);
    __threadfence();
    __syncthreads();
    reseed_body<256>(s);
}

// (C) with many chains: the same, by RESEED_G workgroups of RESEED_NT threads (16384 threads: one word of the claim bitmap each per pass,
// the look-ahead window of HARC_LOOK_CHUNKS x 1024 words in ONE pass).  The single workgroup took 59 us per super-round at configs[2]
// (21 000 seeds + 170 000 look-ahead ids per round: a dozen dependent iterations of load, block scan, hand out) and 167 us on a
// minimizer-bucket shard -- a tenth of the chain phase there.  The workgroups meet at a counter in global memory (they are all resident:
// 64 workgroups on 256 CUs): once after counting (chains that want a seed per workgroup + unclaimed reads per workgroup's words; more
// passes, one meeting each, only when a million bits below the cursor do not hold enough unclaimed reads), once for the look-ahead; the
// read id where the seeds end travels through a flag.  What is computed is the single workgroup's result (the oracle's), id for id.
// g = [0] meeting counter [1] flag: 2 + cursor after the seeds [2] a wait ran out (sticky) [3] unused [4 .. 4+4G) per workgroup: chains wanting a seed, unclaimed
// reads of the pass (two sets, even and odd passes), unclaimed reads of the look-ahead window.  k_resolve zeroes g[0..1] before every launch.
#define RESEED_NT 256
#define RESEED_G 64
#ifndef RESEED_ONE_SET
#define RESEED_ONE_SET 0     // 1 (make variant VFLAGS=-DRESEED_ONE_SET=1): the race of round 3 back in, to see tests/test_gpu_config_size.py's stress test fail
#endif
// (every wait is bounded: a meeting that does not happen within ~2^24 polls -- seconds; it takes microseconds -- raises g[2] and lets everybody
// through, the kernel ends with a meaningless result and the host fails the run at its next look at the flag, instead of a GPU that never
// comes back: with one set of per-pass counts workgroups could disagree on the last pass and wait for each other for ever)
#define RESEED_SPIN_LIMIT (1u << 24)
__device__ __forceinline__ void grid_meet(unsigned int *cnt, unsigned int target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(cnt, 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > RESEED_SPIN_LIMIT || __hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(RESEED_NT) void k_reseed_mg(S1Args s, unsigned int *g)
{
    constexpr int NT = RESEED_NT, G = RESEED_G;
    __shared__ uint32_t sm[20];
    __shared__ uint32_t spre[3][G + 1];
    const int t = threadIdx.x, b = blockIdx.x;
    const uint32_t gid = (uint32_t)b * NT + (uint32_t)t;
    unsigned int *const gA = g + 4, *const gB0 = g + 4 + G, *const gC = g + 4 + 3 * G;
    unsigned int meet = 0;
    // chains per thread, contiguous and a multiple of 8 so that the need bytes are read as 64-bit words (the array is padded with zeros)
    const uint32_t chunk = (((s.K + G * NT - 1) / (G * NT)) + 7) & ~7u;
    const uint32_t c0 = gid * chunk;
    uint32_t mycnt = 0;
    if (c0 < s.K) for (uint32_t c = c0; c < c0 + chunk; c += 8) mycnt += (uint32_t)__popcll(*(const unsigned long long *)(s.need + c));
    uint32_t totA; const uint32_t offA = block_excl_scan_u32<NT>(mycnt, sm, &totA);
    const long long cursor0 = *s.cursor;
    long long top = cursor0;                                       // the pass looks at the G * NT words from the word of `top` downwards
    uint32_t R = 0, assigned = 0;
    long long look = -1;
    int pass = 0;
    for (;; pass++) {
        const long long cwd = top >> 6, wi = cwd - (long long)gid;
        unsigned long long bits = 0;
        if (top >= 0 && wi >= 0 && gid < s.reseed_win) {
            bits = ~s.claimed[wi];
            if (wi == cwd) { const int tb = (int)(top & 63); if (tb < 63) bits &= (2ULL << tb) - 1ULL; }
        }
        uint32_t totB; const uint32_t offB = block_excl_scan_u32<NT>((uint32_t)__popcll(bits), sm, &totB);
        // two sets of per-workgroup counts, by the parity of the pass: a workgroup that is already counting pass p + 1 must not overwrite what
        // a slower one still has to read of pass p (it cannot get further ahead than that: the next meeting waits for everybody)
        unsigned int *const gB = gB0 + (RESEED_ONE_SET ? 0 : (pass & 1)) * G;
        if (t == 0) {
            if (pass == 0) __hip_atomic_store(gA + b, totA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gB + b, totB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_meet(g, (unsigned int)G * ++meet);
        if (s.reseed_stress && (b & 1)) for (uint32_t i = 0; i < s.reseed_stress; i++) __builtin_amdgcn_s_sleep(127);     // tests: late readers of the counts
        if (t < 64) {                                              // one wave: prefix sums over the workgroups
            const uint32_t va = (pass == 0 && t < G) ? __hip_atomic_load(gA + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t vb = t < G ? __hip_atomic_load(gB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t ta, tb2;
            const uint32_t ea = wave_excl_scan_u32(va, &ta), eb = wave_excl_scan_u32(vb, &tb2);
            if (t < G) { if (pass == 0) spre[0][t] = ea; spre[1][t] = eb; }
            if (t == 0) { if (pass == 0) spre[0][G] = ta; spre[1][G] = tb2; }
        }
        __syncthreads();
        if (pass == 0) {
            R = spre[0][G];
            if (R == 0) return;                                    // nobody wants a seed (every workgroup sees the same R)
            if (mycnt) {                                           // rank of every chain that wants a seed, ascending chain id
                uint32_t r = spre[0][b] + offA;
                for (uint32_t c = c0; c < c0 + chunk; c += 8) {
                    unsigned long long w = *(const unsigned long long *)(s.need + c);
                    while (w) { const int bb = __ffsll((long long)w) - 1; w &= w - 1; s.needrank[c + (uint32_t)(bb >> 3)] = r++; }
                }
            }
        }
        const uint32_t total = spre[1][G], off = spre[1][b] + offB;
        unsigned long long newclaim = 0; uint32_t k = 0;
        while (bits && assigned + off + k < R) {
            const int bb = 63 - __clzll((long long)bits);
            bits &= ~(1ULL << bb);
            const uint32_t id = (uint32_t)(wi * 64 + bb), r = assigned + off + k;
            k++; newclaim |= 1ULL << bb;
            s.seedbuf[r] = id;
            if (r == R - 1) __hip_atomic_store(g + 1, id + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // 2 + (id - 1): where the seeds end
        }
        if (newclaim) s.claimed[wi] |= newclaim;                   // every word has one owner
        if (assigned + total >= R) {                               // the seeds are all found: everybody waits for the place of the last one
            assigned = R;
            if (t == 0) {
                unsigned int f, spins = 0;
                while ((f = __hip_atomic_load(g + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > RESEED_SPIN_LIMIT || __hip_atomic_load(g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(g + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); f = 2u; break; }
                }
                spre[2][0] = f;
            }
            __syncthreads();
            look = (long long)spre[2][0] - 2;
            break;
        }
        assigned += total;
        top = (cwd - (long long)s.reseed_win + 1) * 64 - 1;        // below the words of this pass
        if (top < 0) { look = -1; break; }                         // the bitmap is exhausted: the remaining chains finish
        __syncthreads();
    }
    // look-ahead: the next assigned * HARC_NSUGG unclaimed ids below the cursor, in HARC_LOOK_CHUNKS x 1024 words; nothing is claimed, the cursor moves below them
    const uint32_t want = assigned * (uint32_t)s.nsugg_per_seed;
    uint32_t got = 0;
    long long cursor = look;
    if (look >= 0 && want > 0) {
        const long long cwd = look >> 6, wi = cwd - (long long)gid;
        unsigned long long bits = 0;
        if (wi >= 0 && gid < (uint32_t)HARC_LOOK_CHUNKS * 1024u) {
            bits = ~s.claimed[wi];
            if (wi == cwd) { const int tb = (int)(look & 63); if (tb < 63) bits &= (2ULL << tb) - 1ULL; }
        }
        uint32_t totC; const uint32_t offC = block_excl_scan_u32<NT>((uint32_t)__popcll(bits), sm, &totC);
        if (t == 0) __hip_atomic_store(gC + b, totC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_meet(g, (unsigned int)G * ++meet);
        if (t < 64) {
            const uint32_t vc = t < G ? __hip_atomic_load(gC + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t tc; const uint32_t ec = wave_excl_scan_u32(vc, &tc);
            if (t < G) spre[2][t] = ec;
            if (t == 0) spre[2][G] = tc;
        }
        __syncthreads();
        const uint32_t total = spre[2][G], off = spre[2][b] + offC;
        const uint32_t take = total >= want ? want : total;
        uint32_t k = 0;
        while (bits && off + k < take) {
            const int bb = 63 - __clzll((long long)bits);
            bits &= ~(1ULL << bb);
            const uint32_t id = (uint32_t)(wi * 64 + bb);
            s.seedbuf[R + off + k] = id;
            if (off + k == take - 1) *s.cursor = (long long)id - 1;    // the cursor goes below the last look-ahead seed handed out (everybody has read it long ago)
            k++;
        }
        got = take;
    }
    // the chains take their seeds themselves at the top of the next k_steps (rank -> seedbuf)
    if (gid == 0) {
        s.rmeta[0] = R; s.rmeta[1] = assigned; s.rmeta[2] = got; s.rmeta[3] += (uint32_t)pass;     // [3]: passes beyond the first, whole run (trace)
        if (got == 0) *s.cursor = cursor < -1 ? -1 : cursor;
    }
}

// ------------------------------------------------------------------------------------------------ big bins stay short
// The reference deletes a read from its bins when it is claimed (reorder.cpp:396-432); here the claim bitmap is the truth and a bin
// only shrinks from the top (hints).  In repeats and low-complexity sequence a bin holds thousands of reads and every scan would wade
// through the claimed ones again: between super-rounds one wave per large bin packs the unclaimed ids to the front of the bin (order
// kept) and lowers the count.  What a scan sees -- the unclaimed reads of the bin, highest id first -- does not change.
#define HARC_HUGEBIN 512u    // bins that start with more reads than this are compacted by a workgroup of 1024 threads (k_compact_huge), the others by a wave
__global__ void k_huge_list(const uint32_t *sz, uint32_t nlarge, uint32_t *huge, unsigned int *nhuge)
{
    const uint32_t b = harc_gid32();
    if (b < nlarge && sz[b] > HARC_HUGEBIN) huge[atomicAdd(nhuge, 1u)] = b;
}
// A wave per bin walks a bin of 17 000 reads in 270 dependent passes while the chip idles: the kernel lasted as long as its largest bin
// (114 us per super-round on c3sd, a tenth of the chain phase).  Sixteen waves per such bin, order kept by a block scan; every thread has
// its entry in registers before the first one of the pass is overwritten.
template <int W> __global__ __launch_bounds__(1024) void k_compact_huge(S1Args s, const unsigned long long *list, const uint32_t *huge, uint32_t nhuge)
{
    __shared__ uint32_t sm[20];
    if (blockIdx.x >= nhuge) return;
    const uint32_t b = huge[blockIdx.x];
    const int t = threadIdx.x;
    const unsigned long long e = list[b];
    const int l = (int)(e & 1); const uint64_t si = e >> 1;
    HashSlot *slot = &s.slots[l][si];
    const uint32_t cw = slot->count;
    if (cw & SLOT_DEAD) return;
    const uint32_t cnt = cw & SLOT_CNT_MASK;
    const uint2 lt = s.largetab[b];
    uint32_t *ids = const_cast<uint32_t *>(s.ids[l]) + lt.x;
    uint64_t *mir = s.mirror + (size_t)lt.y * W;
    // Only the TOP of the bin is looked at: a scan starts at the highest id and ends with the maxsearch-th unclaimed entry (reorder.cpp:540), so all it
    // can ever reach is the last stretch of the bin that holds maxsearch unclaimed entries.  That stretch [A, cnt) is found from the top in growing
    // pieces and compacted in place; what lies below A keeps its claimed entries until the stretch above it has thinned out and the next passes reach
    // down to it (A = 0: the whole bin, as before).  A bin of 130 000 reads of a diverged repeat family was 127 dependent passes of this workgroup
    // every super-round -- the kernel lasts as long as its largest bin: 540 us per round, a seventh of the chain phase of configs[2] with repeats.
    uint32_t A = 0;
    for (uint32_t T = 4096; T < cnt && T <= (1u << 30); T *= 4) {      // (T *= 4 would wrap to 0 beyond 2^30 and never end)
        uint32_t u = 0;
        for (uint32_t pos = cnt - T; pos < cnt; pos += 1024) {
            if (pos + t < cnt) { const uint32_t rid = ids[pos + t]; u += ((s.claimed[rid >> 6] >> (rid & 63)) & 1ULL) ? 0u : 1u; }
        }
        uint32_t total; (void)block_excl_scan_u32<1024>(u, sm, &total);
        __syncthreads();
        if (total >= (uint32_t)s.maxsearch) { A = cnt - T; break; }
    }
    uint32_t out = A;
    for (uint32_t pos = A; pos < cnt; pos += 1024) {
        const bool valid = pos + t < cnt;
        uint32_t rid = 0; bool un = false;
        if (valid) { rid = ids[pos + t]; un = !((s.claimed[rid >> 6] >> (rid & 63)) & 1ULL); }
        uint32_t total; const uint32_t off = block_excl_scan_u32<1024>(un ? 1u : 0u, sm, &total);
        // (round 6, as k_compact_bins: words fetched and entries written only where something moves -- uniform for the workgroup: the pass starts behind
        // `out`, or drops an entry)
        const uint32_t nvalid = cnt - pos < 1024u ? cnt - pos : 1024u;
        if (out != pos || total != nvalid) {
            const uint32_t at = out + off;                        // out <= pos: never ahead of the entries of this pass
            const bool moves = un && at != pos + (uint32_t)t;
            uint64_t rw[W];
            if (moves) {
#pragma unroll
                for (int w = 0; w < W; w++) rw[w] = mir[(size_t)(pos + t) * W + w];
            }
            __builtin_amdgcn_s_waitcnt(0);                        // every entry that moves has arrived ...
            __syncthreads();                                      // ... in every wave, before the first one of the pass is overwritten
            if (moves) {
                ids[at] = rid;
#pragma unroll
                for (int w = 0; w < W; w++) mir[(size_t)at * W + w] = rw[w];
            }
        }
        out += total;
    }
    if (t == 0 && out < cnt) {
        if (out == 0) atomicOr(&slot->count, SLOT_DEAD);
        else slot->count = out | (cw & ~SLOT_CNT_MASK);
    }
}
template <int W> __global__ __launch_bounds__(64) void k_compact_bins(S1Args s, const unsigned long long *list, uint32_t nlist, const uint32_t *sz0)
{
    const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;      // (rows of at most 2^25 bins: wave_grid)
    if (b >= nlist) return;
    if (sz0 && sz0[b] > HARC_HUGEBIN) return;                     // k_compact_huge's
    const int lane = threadIdx.x;
    const unsigned long long e = list[b];
    const int l = (int)(e & 1); const uint64_t si = e >> 1;
    HashSlot *slot = &s.slots[l][si];
    const uint32_t cw = slot->count;
    if (cw & SLOT_DEAD) return;
    const uint32_t cnt = cw & SLOT_CNT_MASK;
    const uint2 lt = s.largetab[b];                               // the slot's `start` is b (k_large_fill)
    uint32_t *ids = const_cast<uint32_t *>(s.ids[l]) + lt.x;
    uint64_t *mir = s.mirror + (size_t)lt.y * W;
    // (round 6: the words of an entry are fetched -- and id and words written -- only where the entry MOVES.  Most bins lose nothing in a super-round: every
    // round read 36 bytes and wrote 36 per entry of every live large bin, 108 us per round at configs[2] with repeats, 1312 rounds a step)
    uint32_t out = 0;
    for (uint32_t pos = 0; pos < cnt; pos += 64) {
        const bool valid = pos + lane < cnt;
        uint32_t rid = 0; bool un = false;
        if (valid) { rid = ids[pos + lane]; un = !((s.claimed[rid >> 6] >> (rid & 63)) & 1ULL); }
        const unsigned long long um = __ballot(un);
        const uint32_t at = out + (uint32_t)__popcll(um & ((1ULL << lane) - 1ULL));     // out <= pos: never ahead of the entries of this pass
        const bool moves = un && at != pos + (uint32_t)lane;
        if (__ballot(moves)) {
            uint64_t rw[W];
            if (moves) {
#pragma unroll
                for (int w = 0; w < W; w++) rw[w] = mir[(size_t)(pos + lane) * W + w];
            }
            if (moves) {                                           // (every lane's words are in its registers before the first store of the wave is issued)
                ids[at] = rid;
#pragma unroll
                for (int w = 0; w < W; w++) mir[(size_t)at * W + w] = rw[w];
            }
        }
        out += (uint32_t)__popcll(um);
    }
    if (lane == 0 && out < cnt) {
        if (out == 0) atomicOr(&slot->count, SLOT_DEAD);
        else slot->count = out | (cw & ~SLOT_CNT_MASK);
    }
}

// ------------------------------------------------------------------------------------------------ finalisation
__global__ void k_chain_counts(const ChainHdr *hdr, const uint4 *cstat, const uint4 *cstat_coop, const uint2 *csteps, const uint2 *cst2, uint32_t K, uint32_t *nmain, uint32_t *nsing, unsigned long long *stats)
{
    const uint32_t c = harc_gid32();
    uint32_t np = 0, nc = 0, nu = 0, um = 0, cf = 0, ncs = 0, cnc = 0, cnu = 0, cncs = 0, sd = 0, sc = 0;
    if (c < K) {
        nmain[c] = hdr[c].n_main; nsing[c] = hdr[c].n_sing;
        const uint4 st = cstat[c], sx = cstat_coop[c]; const uint2 q = cst2[c], w = csteps[c];
        np = st.x + sx.x; nc = st.y + sx.y; nu = st.z + sx.z; ncs = st.w + sx.w; um = q.x; cf = q.y;
        cnc = sx.y; cnu = sx.z; cncs = sx.w; sd = w.x; sc = w.y;
    }
    np = wave_sum_u32(np); nc = wave_sum_u32(nc); nu = wave_sum_u32(nu); um = wave_sum_u32(um); cf = wave_sum_u32(cf); ncs = wave_sum_u32(ncs);
    cnc = wave_sum_u32(cnc); cnu = wave_sum_u32(cnu); cncs = wave_sum_u32(cncs); sd = wave_sum_u32(sd); sc = wave_sum_u32(sc);
    if ((threadIdx.x & 63) == 0) {
        if (um) atomicAdd(&stats[ST_UNMATCHED], (unsigned long long)um);
        if (cf) atomicAdd(&stats[ST_CONFLICTS], (unsigned long long)cf);
        if (np) atomicAdd(&stats[ST_PROBES], (unsigned long long)np);
        if (nc) atomicAdd(&stats[ST_CANDS], (unsigned long long)nc);
        if (nu) atomicAdd(&stats[ST_USEFUL], (unsigned long long)nu);
        if (ncs) atomicAdd(&stats[ST_CANDS_SEQ], (unsigned long long)ncs);
        if (cnc) atomicAdd(&stats[ST_COOP_CANDS], (unsigned long long)cnc);
        if (cnu) atomicAdd(&stats[ST_COOP_USEFUL], (unsigned long long)cnu);
        if (cncs) atomicAdd(&stats[ST_COOP_CANDS_SEQ], (unsigned long long)cncs);
        if (sd) atomicAdd(&stats[ST_DENSE_STEPS], (unsigned long long)sd);
        if (sc) atomicAdd(&stats[ST_COOP_STEPS], (unsigned long long)sc);
    }
}
// per-chain streams concatenated in chain order (reorder.cpp:778-821)
// Two passes: the record of read i goes, as ONE 8-byte store, to its place in chain-major order (a random 32-byte sector per read instead
// of four: the scatter runs at the random-access ceiling of the memory system); a streaming pass then splits the records into the four
// files of reorder.cpp:722-830.
__global__ __launch_bounds__(256) void k_s1_singles(const uint2 *slog, unsigned long long nlog, uint32_t K, const uint32_t *base_sing, uint32_t *order_s, unsigned long long *found)
{
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nlog; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint2 r = slog[i];
        if (r.x < K) { order_s[base_sing[r.x] + r.y] = (uint32_t)i; mine++; }
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(found, (unsigned long long)cnt);
}
// one wave per page of the main stream: 64 records -> their place in the chain-major output, split into the stage-I arrays on the way
__global__ __launch_bounds__(256) void k_pages_out(const uint2 *pg_rec, const uint2 *pg_hdr, uint32_t npages, const uint32_t *n_main, const uint32_t *base_main,
                                                   uint32_t *order, uint8_t *flag, uint8_t *pos, uint8_t *rc)
{
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= npages) return;
    const uint2 hd = pg_hdr[p];
    const uint32_t nm = n_main[hd.x], first = hd.y * 64u;
    if (first + lane >= nm) return;
    const uint2 r = pg_rec[(size_t)p * 64 + lane];
    const size_t at = (size_t)base_main[hd.x] + first + lane;
    order[at] = r.x; pos[at] = (uint8_t)(r.y & 0xFF);
    flag[at] = (r.y >> 8) & 1 ? '1' : '0'; rc[at] = (r.y >> 9) & 1 ? 'r' : 'd';
}
// temp.dna in HBM: read `order[i]`, reverse-complemented where rc[i]=='r' (reorder.cpp:743-752)
// W lanes per read, one 64-bit word each: the W words of a read are ONE request of W x 8 bytes to one sector (profiles/r05/random_access_ceiling.txt:
// a random sector fetched by narrow loads of neighbouring lanes runs at 48.5 G requests/s, the same 32 bytes as two dwordx4 loads of ONE lane -- 64
// distinct lines per instruction -- at 26.3: this kernel was 10.4 ms at configs[2] with a thread per read)
template <int W> __global__ void k_orient(const uint64_t *reads, const uint32_t *order, const uint8_t *rc, uint32_t m, int L, uint64_t *out)
{
    constexpr int RPW = 64 / W;                                   // reads per wave
    const int lane = threadIdx.x & 63, rl = lane / W, w = lane % W;
    const uint64_t wave = harc_gid() >> 6;                       // (64-bit: W x m work-items pass 2^32 from a billion reads of 128 bases on; the grid is folded into rows)
    const uint64_t i64 = wave * RPW + (uint64_t)rl;
    const uint32_t i = (uint32_t)i64;
    const bool on = rl < RPW && i64 < m;
    uint64_t mine = 0; bool rev = false;
    if (on) {
        const uint32_t rid = order[i];
        mine = reads[(size_t)rid * W + w];
        rev = rc != nullptr && rc[i] == 'r';
    }
    uint64_t r[W], o[W];
#pragma unroll
    for (int k = 0; k < W; k++) r[k] = shfl_u64(mine, rl * W + k);
    if (!on) return;
    uint64_t ow = mine;
    if (rev) { rc_words<W>(r, L, o); ow = sel0<W>(o, w); }
    out[(size_t)i * W + w] = ow;
}

// ------------------------------------------------------------------------------------------------ host side
int harc_dict_alloc(harc_amd_ctx *c, DictDev *d, uint32_t n, uint64_t cap_like)
{
    d->cap = 0; d->slots = nullptr; d->ids = nullptr; d->d_nbins = nullptr;
    if (n == 0) return HARC_AMD_OK;
    if (cap_like) d->cap = cap_like;                              // same geometry as a sibling table (k_steps addresses both dictionaries alike)
    else {   // load factor 1/4 when HBM allows (fewer dependent re-probes: the chain kernel is latency-bound), else 1/3, else 1/2: a table takes at most
             // 15 % of the device's memory.  Measured at configs[3] (810 M reads, profiles/r04/bench_c4_cap*.json): 4 / 3 / 2 slots per read =
             // 594 / 591 / 569 Mreads/s at 52 / 39 / 26 GB per table -- there the rule picks 3; configs[2] and configs[4]'s share keep 4
        unsigned long long m = 4;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) { fr += c->pool_total; while (m > 2 && ((double)m * n * sizeof(HashSlot) > 0.2 * (double)fr || (double)m * n * sizeof(HashSlot) > 0.15 * (double)tot)) m--; }
        if (c->P.table_slots_per_read >= 2 && c->P.table_slots_per_read <= 4) m = (unsigned long long)c->P.table_slots_per_read;      // the caller's choice (harc_amd_params)
        if (const char *e = getenv("HARC_AMD_CAPMULT")) m = strtoull(e, nullptr, 10);      // (tests force a fuller table on small inputs)
        d->cap = (((m < 2 ? 2 : m) * n + 4) + 3) & ~3ull;           // whole 64-B buckets of 4 slots
    }
    RC_TRY(dalloc(c, &d->slots, d->cap)); RC_TRY(dalloc(c, &d->ids, n)); RC_TRY(dalloc(c, &d->d_nbins, 2));
    return HARC_AMD_OK;
}
int harc_dict_build(harc_amd_ctx *c, DictDev *d, uint64_t *keys, uint32_t *ids, uint32_t n, unsigned kbits)
{
    (void)kbits;                                                  // the scrambled keys use all 64 bits
    if (n == 0) return HARC_AMD_OK;
    PoolScope scope(c);                                           // temporaries go on every way out
    uint64_t *k1 = nullptr; uint32_t *h0 = nullptr, *b0 = nullptr, *bs = nullptr;
    RC_TRY(dalloc(c, &k1, n)); RC_TRY(dalloc(c, &h0, n)); RC_TRY(dalloc(c, &b0, n)); RC_TRY(dalloc(c, &bs, n));
    // top bits the sort looks at: log2(n) + 8, in whole radix digits, all 64 from 2^36 keys on (HARC_AMD_SORT_BITS forces a count: tests use 8,
    // where nearly every stretch is mixed, and 64)
    unsigned sbits = 64;
    { unsigned lg = 1; while (((uint64_t)1 << lg) < n) lg++; sbits = ((lg + 8 + 7) / 8) * 8; if (sbits > 64) sbits = 64; }
    if (const char *e = getenv("HARC_AMD_SORT_BITS")) { const int x = atoi(e); if (x >= 1 && x <= 64) sbits = (unsigned)x; }
    uint32_t *mixed = nullptr; unsigned int *mmeta = nullptr; uint64_t *k2 = nullptr;
    RC_TRY(dalloc(c, &mixed, MIXED_MAX)); RC_TRY(dalloc(c, &mmeta, 4));
    RC_TRY(dalloc(c, &k2, n));                                    // the sort's output on the top bits; then (k_bin_starts) what the placement's max-scan starts from
    uint64_t *const k2v = k2;
    hipLaunchKernelGGL(k_scramble_keys, harc_grid256(n), dim3(256), 0, c->stream, keys, n, sbits < 64 ? sbits : 0u);
    uint32_t nbins = 0;
    for (;;) {
        HIP_TRY(hipMemsetAsync(mmeta, 0, 16, c->stream));
        HIP_TRY(hipMemsetAsync(d->d_nbins, 0, 8, c->stream));
        if (sbits < 64) {
            RC_TRY(prim_sort_pairs_u64_u32(c, keys, k2, ids, d->ids, n, sbits));       // stable: ids ascending inside a bin (reorder.cpp:372-384)
            hipLaunchKernelGGL(k_mixed_find, harc_grid256(n), dim3(256), 0, c->stream, (const uint64_t *)k2, n, sbits, k1, mixed, mmeta);
            hipLaunchKernelGGL(k_mixed_own, dim3(MIXED_MAX / 256), dim3(256), 0, c->stream, (const uint64_t *)k1, (int)(64 - sbits), mixed, mmeta);
            hipLaunchKernelGGL(k_mixed_fix, dim3(MIXED_MAX / 256), dim3(256), 0, c->stream, k1, d->ids, n, (int)(64 - sbits), (const uint32_t *)mixed, mmeta);
        } else RC_TRY(prim_sort_pairs_u64_u32(c, keys, k1, ids, d->ids, n, 64));
        hipLaunchKernelGGL(k_mark_heads, harc_grid256(n), dim3(256), 0, c->stream, k1, n, h0);
        RC_TRY(prim_excl_scan_u32(c, h0, b0, n));
        hipLaunchKernelGGL(k_bin_starts, harc_grid256(n), dim3(256), 0, c->stream, h0, b0, n, bs, d->d_nbins, (const uint64_t *)k1, d->cap, k2v);
        unsigned int mm[2] = { 0, 0 };
        HIP_TRY(hipMemcpyAsync(&nbins, d->d_nbins, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mm, mmeta, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[index] sort on the top %u bits: %u places where different keys share them%s\n", sbits, mm[0], mm[1] ? " -- too many or too long, sorting on all bits" : "");
        if (!mm[1] || sbits == 64) break;
        hipLaunchKernelGGL(k_rotate_keys, harc_grid256(n), dim3(256), 0, c->stream, keys, n, 64u - sbits);      // back to the scrambled keys as they are
        sbits = 64;
    }
    // the placement clears the table on its way when the gaps between bins are short (fill); a table of few bins is cleared here
    const int fill = getenv("HARC_AMD_TABLE_FILL") ? atoi(getenv("HARC_AMD_TABLE_FILL")) != 0 : (uint64_t)nbins * 16 >= d->cap;
    if (!fill) HIP_TRY(hipMemsetAsync(d->slots, 0, d->cap * sizeof(HashSlot), c->stream));
    d->nbins = nbins;
    {
        uint64_t *v = k2v, *q = keys;                              // the unsorted keys are not needed any more
        const unsigned gb = (nbins + 255) / 256;
        RC_TRY(prim_incl_max_u64(c, v, q, nbins));
        for (int pass = 0; pass < 2; pass++)
            hipLaunchKernelGGL(k_table_place, dim3(gb), dim3(256), 0, c->stream, (const uint64_t *)k1, (const uint32_t *)d->ids, (const uint32_t *)bs, nbins, n, (const uint64_t *)q,
                               d->slots, d->cap, d->bigthresh, d->large_list, d->large_n, d->large_max, d->large_tag, d->d_nbins, pass, pass == 0 ? 0u : (gb > 64 ? gb - 64 : 0u), fill);
    }
    HIP_TRY(hipGetLastError());
    uint32_t nb2[2] = { 0, 0 };
    HIP_TRY(hipMemcpyAsync(nb2, d->d_nbins, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                                    // temporaries are reused after this point
    scope.release_now();
    if (nb2[1]) { harc_set_error("dictionary: %u bins hold more than %u reads with the same k-mer (count field of the slot), or lie more than 16384 bins beyond the end of the table", nb2[1], SLOT_CNT_MASK); return HARC_AMD_EINVAL; }
    return HARC_AMD_OK;
}
void harc_dict_free(harc_amd_ctx *c, DictDev *d)
{
    if (d->slots) harc_dev_free(c, d->slots);
    if (d->ids) harc_dev_free(c, d->ids);
    if (d->d_nbins) harc_dev_free(c, d->d_nbins);
    d->slots = nullptr; d->ids = nullptr; d->d_nbins = nullptr; d->cap = 0;
}

static std::vector<uint16_t> make_probe_table(const harc_amd_params &P)
{
    // reorder.cpp:517-649: for j: forward l=0,1 (skip if dict_end[l]+j >= readlen), reverse l=0,1 (skip if dict_start[l] <= j)
    std::vector<uint16_t> t;
    for (int j = 0; j < P.maxmatch; j++) {
        for (int l = 0; l < 2; l++) if (P.dict_end[l] + j < P.readlen) t.push_back((uint16_t)(j | (0 << 8) | (l << 9)));
        for (int l = 0; l < 2; l++) if (P.dict_start[l] > j) t.push_back((uint16_t)(j | (1 << 8) | (l << 9)));
    }
    return t;
}

static uint32_t auto_chains(uint32_t N, int reads_per_chain)
{
    // one chain per ~2048 reads keeps chains sparse on the genome (every chain costs about one extra contig);
    // 65536 waves is several full waves of occupancy on 256 CUs.  Small inputs: up to 2048 chains (two waves per SIMD) while a chain
    // still has 1024 reads to walk -- with fewer chains the kernel is a handful of waves waiting on HBM round trips; with shorter
    // chains the streams grow (one chain per 256 reads: +10 % on the 26x test set of test_P5_stream_sizes_vs_reference_t8).
    uint32_t k = N / (uint32_t)(reads_per_chain > 0 ? reads_per_chain : 2048);
    const uint32_t floor_k = N / 1024 < 2048 ? N / 1024 : 2048;
    if (k < floor_k) k = floor_k;
    if (k > 65536) k = 65536;
    if (k < 1) k = 1;
    return k;
}

// Everything stage1_run_w owns besides pool memory: released on every way out.  The launches of the dominant kernel are timed with a
// fixed ring of event pairs (params.profile = 1), not with two new events per launch.
// HIP-event pairs around the launches of ONE kernel (params.profile): a ring of 64 pairs, a pair's previous use read before it is recorded again
struct EventRing {
    static constexpr int RING = 64;
    hipEvent_t ring[RING][2];
    int ring_used = 0; uint64_t ring_next = 0; double ms = 0; uint64_t launches = 0;
    EventRing() { for (int i = 0; i < RING; i++) ring[i][0] = ring[i][1] = nullptr; }
    ~EventRing() { for (int i = 0; i < ring_used; i++) { (void)hipEventDestroy(ring[i][0]); (void)hipEventDestroy(ring[i][1]); } }
    // the pair for the next launch; its previous use (RING launches ago) is read first
    int next_pair(hipEvent_t **pair)
    {
        const int k = (int)(ring_next % RING);
        if (k >= ring_used) { HIP_TRY(hipEventCreate(&ring[k][0])); HIP_TRY(hipEventCreate(&ring[k][1])); ring_used = k + 1; }
        else RC_TRY(collect(k));
        ring_next++; launches++;
        *pair = ring[k];
        return HARC_AMD_OK;
    }
    int collect(int k) { float x = 0; HIP_TRY(hipEventSynchronize(ring[k][1])); HIP_TRY(hipEventElapsedTime(&x, ring[k][0], ring[k][1])); ms += x; return HARC_AMD_OK; }
    int collect_all() { const uint64_t n = ring_next < (uint64_t)RING ? ring_next : (uint64_t)RING; for (uint64_t i = 0; i < n; i++) RC_TRY(collect((int)i)); ring_next = 0; return HARC_AMD_OK; }
};
struct S1Resources {
    hipEvent_t e[3] = { nullptr, nullptr, nullptr };
    hipEvent_t eb[2] = { nullptr, nullptr };
    EventRing dense, coop;                                        // the main kernel's launches; the cooperative kernel's (walks that reach a bin of more than HARC_LARGEBIN reads)
    unsigned long long *h_stats = nullptr;
    ~S1Resources()
    {
        for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x);
        for (hipEvent_t x : eb) if (x) (void)hipEventDestroy(x);
        if (h_stats) (void)hipHostFree(h_stats);
    }
    int init()
    {
        for (auto &x : e) HIP_TRY(hipEventCreate(&x));
        for (auto &x : eb) HIP_TRY(hipEventCreate(&x));
        HIP_TRY(hipHostMalloc((void **)&h_stats, (ST_N + HARC_COOPCNT + 8 + 1) * 8));      // statistics, cooperative-walk counters, replica digest, k_reseed_mg's timeout flag
        return HARC_AMD_OK;
    }
};

template <int W> static int stage1_run_w(harc_amd_ctx *c)
{
    const harc_amd_params &P = c->P;
    const uint32_t N = c->N;
    uint32_t K = P.num_chains > 0 ? (uint32_t)P.num_chains : auto_chains(N, P.reads_per_chain);
    if (K > HARC_MAXK) K = HARC_MAXK;
    if (N == 0 || K > N) K = 1;                                  // floor(N/K)=0: only chain 0 ever runs (reorder.cpp:484-490)
    c->C.chains = K;
    int nsteps = P.num_steps > 0 ? P.num_steps : 16;
    // one chain cannot lose a bid: its output does not depend on S, and every super-round is three launches for S steps --
    // exact mode takes the longest walks there are (a launch per 64 reads instead of per 16)
    if (P.num_steps <= 0 && K == 1) nsteps = 64;
    // ... and with tens of thousands of chains -- every BASELINE-sized input -- 32: half the super-rounds, i.e. half of the ~100 us a round costs beside its walks
    // (arbitration at the random-access rate, new seeds, three launches), for walks that lose their later steps a little more often.  Measured (round 5, S = 16 /
    // 32 / 64): configs[2] 613 / 626 / 604 Mreads/s with 6.50 / 6.48 / 6.44 M contigs and the same xz size; configs[3] 656 / 667; a 50 M-read repeat-rich set 222 /
    // 234; configs[4]'s share 490 / 487.  Below 16 384 chains 16 stays (3.3 M reads: +7 % clean, -18 % with repeats; 1 M reads: slower).  The two-chains-per-wave
    // kernel walks at most 16 steps: asking for it keeps 16.
    // ... where walks rarely meet: on a repeat-rich input at 26x (configs[3] with a human-like repeat content) the longer walks took 3216 super-rounds instead of
    // 2008 and the chain phase 7.5 s instead of 4.2 -- chains of one repeat family walk into each other, and a walk that loses a bid gives up everything behind
    // it.  No count taken DURING the run tells such an input early enough (the share of walks that end at a lost bid is 0.4 % after 16 rounds there as on
    // clean data, 20 % only after a thousand: profiles/r05/s_choice_trace.txt), so the choice is made from the index, below: 32 only where the bins of more
    // than HARC_LARGEBIN reads hold less than 2 % of N entries (clean inputs: none), 16 otherwise.  A function of the input alone, like the chain count.
    // Round 6: also from 2048 chains on where the input is NOT a low-coverage one (the index again: at most 98 % of the reads alone in their first-dictionary bin) -- configs[1]'s
    // stand-in (3.3 M reads at 52x, 2048 chains) 203-206 -> 216-220 Mreads/s with fewer contigs; configs[0]'s (2.9x, where a walk needs a new seed every few reads) would lose 7 %.
    // Decided below, once the index is there.  (A function of the input alone: no environment variable changes S -- k_steps_grp walks at most 16 steps and leaves S = 32 to k_steps.)
    if (nsteps > 64) nsteps = 64;
    // few chains -> every launch is a chain of dependent HBM round trips: fetch whole buckets; many chains -> request-rate bound: single slots
    bool quad = K <= 16384;
    if (const char *e = getenv("HARC_AMD_QUAD")) quad = atoi(e) != 0;
    S1Resources R;
    RC_TRY(R.init());
    hipEvent_t &e0 = R.e[0], &e1 = R.e[1], &e2 = R.e[2];
    unsigned long long *const h_stats = R.h_stats;
    HIP_TRY(hipEventRecord(e0, c->stream));

    // ---- pool layout: [stage-I results, worst case][dictionaries][index scratch -> released][chain state] ; all but the results
    //      are released at the end so that stage II starts right above them.  An error on the way releases everything (run_scope).
    PoolScope run_scope(c);
    RC_TRY(dalloc(c, &c->d_order, (size_t)N + 1)); RC_TRY(dalloc(c, &c->d_flag, (size_t)N + 1)); RC_TRY(dalloc(c, &c->d_pos, (size_t)N + 1));
    RC_TRY(dalloc(c, &c->d_rc, (size_t)N + 1)); RC_TRY(dalloc(c, &c->d_order_s, (size_t)N + 1));
    const harc_mark_t mark_results = harc_pool_mark(c);
    // ---- dictionaries (constructdictionary, reorder.cpp:277-394)
    const bool itrace = getenv("HARC_AMD_TRACE") != nullptr;
    auto inow = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double it0 = inow();
    auto ilap = [&](const char *what) { if (itrace) { (void)hipStreamSynchronize(c->stream); const double t = inow(); fprintf(stderr, "[index] %s: %.1f ms\n", what, 1e3 * (t - it0)); it0 = t; } };
    DictDev dict[2];
    // a bitmap of bloom_bits bits per read in front of each table, two bits set per key (HARC_AMD_S1BLOOM=0: none)
    uint32_t *d_bloom[2] = { nullptr, nullptr }; uint32_t bloom_lines = 0; int bloom_nwin[2] = { 0, 0 };
    // lines by minimizer (bloom_pos) once a bitmap is larger than half the Infinity Cache; m = minimizer length in bases
    int bloom_bits = 16, bloom_m = (P.dict_end[0] - P.dict_start[0] + 1) / 2; size_t bloom_mz_bytes = (size_t)128 << 20;
    if (bloom_m > 16) bloom_m = 16;
    if (const char *e = getenv("HARC_AMD_S1BLOOM_M")) { bloom_m = atoi(e); if (bloom_m < 0) bloom_m = 0; if (bloom_m > 16) bloom_m = 16; }
    if (const char *e = getenv("HARC_AMD_S1BLOOM_MZMB")) bloom_mz_bytes = (size_t)strtoull(e, nullptr, 10) << 20;
    const uint32_t bloom_mmask = bloom_m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * bloom_m)) - 1u);
    if (const char *e = getenv("HARC_AMD_S1BLOOM")) { bloom_bits = atoi(e); if (bloom_bits < 0) bloom_bits = 0; if (bloom_bits > 64) bloom_bits = 64; }
    bool bloom_tiled = false;
    unsigned long long *d_large = nullptr; unsigned int *d_nlarge = nullptr;
    const uint32_t maxlarge = 2 * (N / HARC_LARGEBIN) + 16;
    if (N) {
        RC_TRY(harc_dict_alloc(c, &dict[0], N, 0)); RC_TRY(harc_dict_alloc(c, &dict[1], N, dict[0].cap));
        ilap("tables allocated");
        RC_TRY(dalloc(c, &d_large, maxlarge)); RC_TRY(dalloc(c, &d_nlarge, 4));
        HIP_TRY(hipMemsetAsync(d_nlarge, 0, 16, c->stream));
        for (int l = 0; l < 2; l++) { dict[l].large_list = d_large; dict[l].large_n = d_nlarge; dict[l].large_max = maxlarge; dict[l].large_tag = (uint32_t)l; }
        dict[0].bigthresh = dict[1].bigthresh = HARC_LARGEBIN;     // SLOT_BIG: the bin gets a row of largetab and its reads in `mirror`
        if (bloom_bits) {
            uint64_t nl = ((uint64_t)N * (uint64_t)bloom_bits + 511) / 512 + 1;       // lines of 64 bytes
            if (nl > 0x0FFFFFFFull) nl = 0x0FFFFFFFull;
            bloom_lines = (uint32_t)nl;
            // a bitmap beyond the caches is built tile by tile from sorted keys (k_s1_bloom_tile); a small one takes the atomics in L2 (HARC_AMD_S1BLOOM_TILED=0/1 forces either)
            bloom_tiled = getenv("HARC_AMD_S1BLOOM_TILED") ? atoi(getenv("HARC_AMD_S1BLOOM_TILED")) != 0 : (size_t)bloom_lines * 64 >= ((size_t)64 << 20);
            if (((uint64_t)bloom_lines * 16 + BL_TILE_WORDS - 1) / BL_TILE_WORDS + 1 >= ((uint64_t)1 << BL_PART_BITS)) bloom_tiled = false;
            for (int l = 0; l < 2; l++) {
                RC_TRY(dalloc(c, &d_bloom[l], (size_t)bloom_lines * 16)); if (!bloom_tiled) HIP_TRY(hipMemsetAsync(d_bloom[l], 0, (size_t)bloom_lines * 64, c->stream));
                const int nb = P.dict_end[l] - P.dict_start[l] + 1;                   // bases per key
                const bool same = P.dict_end[0] - P.dict_start[0] == P.dict_end[1] - P.dict_start[1];     // k_steps takes one window count for both
                bloom_nwin[l] = (same && bloom_m > 0 && nb > bloom_m && (size_t)bloom_lines * 64 >= bloom_mz_bytes) ? nb - bloom_m + 1 : 0;
            }
        }
        PoolScope kscope(c);
        uint64_t *kboth[2] = { nullptr, nullptr }; uint32_t *i0 = nullptr;
        RC_TRY(dalloc(c, &kboth[0], N)); RC_TRY(dalloc(c, &kboth[1], N)); RC_TRY(dalloc(c, &i0, N));
        hipLaunchKernelGGL((k_keygen2<W>), harc_grid256(N), dim3(256), 0, c->stream, c->d_reads, N, 2 * P.dict_start[0], 2 * (P.dict_end[0] - P.dict_start[0] + 1),
                           2 * P.dict_start[1], 2 * (P.dict_end[1] - P.dict_start[1] + 1), kboth[0], kboth[1], i0);
        ilap("bitmaps allocated, keys made");
        for (int l = 0; l < 2; l++) {
            const int kbits = 2 * (P.dict_end[l] - P.dict_start[l] + 1);
            uint64_t *const k0 = kboth[l];
            if (bloom_lines && bloom_tiled) {
                PoolScope bscope(c);
                uint64_t *pa = nullptr, *pb = nullptr; RC_TRY(dalloc(c, &pa, (size_t)N + 1)); RC_TRY(dalloc(c, &pb, (size_t)N + 1));
                const uint64_t nwords = (uint64_t)bloom_lines * 16;
                hipLaunchKernelGGL(k_s1_bloom_keys, harc_grid256(N), dim3(256), 0, c->stream, (const uint64_t *)k0, N, bloom_lines, bloom_nwin[l], bloom_mmask, pa);
                RC_TRY(harc_bitmap_from_items(c, pa, pb, N, nwords, d_bloom[l]));
                if (getenv("HARC_AMD_S1BLOOM_VERIFY")) {            // tests: word for word what the atomics build
                    uint32_t *ref = nullptr; unsigned long long *nd = nullptr, hnd = 0;
                    RC_TRY(dalloc(c, &ref, (size_t)nwords)); RC_TRY(dalloc(c, &nd, 1));
                    HIP_TRY(hipMemsetAsync(ref, 0, (size_t)nwords * 4, c->stream)); HIP_TRY(hipMemsetAsync(nd, 0, 8, c->stream));
                    hipLaunchKernelGGL(k_s1_bloom_set, harc_grid256(N), dim3(256), 0, c->stream, (const uint64_t *)k0, N, ref, bloom_lines, bloom_nwin[l], bloom_mmask);
                    hipLaunchKernelGGL(k_s1_bloom_diff, harc_grid256(nwords), dim3(256), 0, c->stream, (const uint32_t *)ref, (const uint32_t *)d_bloom[l], nwords, nd);
                    HIP_TRY(hipMemcpyAsync(&hnd, nd, 8, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    if (hnd) { harc_set_error("stage I bitmap built by tiles differs from the one built with atomics in %llu words", hnd); return HARC_AMD_EINTERNAL; }
                }
            } else if (bloom_lines) hipLaunchKernelGGL(k_s1_bloom_set, harc_grid256(N), dim3(256), 0, c->stream, (const uint64_t *)k0, N, d_bloom[l], bloom_lines, bloom_nwin[l], bloom_mmask);
            ilap("bitmap built");
            RC_TRY(harc_dict_build(c, &dict[l], k0, i0, N, (unsigned)kbits));
            ilap("table built");
        }
    }
    // Low coverage: almost every read has its k-mer to itself (distinct k-mers / reads = (1 - e^-x) / x with x reads per genome position:
    // 0.986 at 2.9x, 0.95 at 11x, 0.77 at 52x).  There chains cost next to nothing in compressed size (configs[0] stand-in: 1.003 of the
    // reference's -t 8 with twice the chains) and a small input leaves the chip idle: up to 4096 chains of at least 256 reads (round 6; 512 before: configs[0]'s
    // stand-in 96 -> 119 Mreads/s with 3906 instead of 1953 chains -- 56 -> 32 super-rounds of a kernel whose launch lasts as long as one wave's 16 steps --
    // for 1.0050 instead of 1.0026 of the reference's -t 8 in xz bytes).  At 52x the same costs 6 % (configs[1]'s stand-in with 4096 chains: 274 instead of
    // 202 Mreads/s, 1.030 instead of 0.968 of the reference's size: not taken), so the rule asks the index.
    if (N && P.num_chains <= 0 && P.reads_per_chain <= 0 && (double)dict[0].nbins > 0.98 * (double)N) {
        const uint32_t k2 = N / 256 < 4096 ? N / 256 : 4096;
        if (k2 > K) { K = k2; c->C.chains = K; if (!getenv("HARC_AMD_QUAD")) quad = K <= 16384; }
    }
    // bins large enough to be worth compacting between super-rounds were listed by k_table_insert (none on ordinary data)
    uint32_t nlarge = 0;
    if (N) {
        HIP_TRY(hipMemcpyAsync(&nlarge, d_nlarge, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (nlarge > maxlarge) { harc_set_error("stage I: %u bins above %u reads, list of %u", nlarge, HARC_LARGEBIN, maxlarge); return HARC_AMD_EINTERNAL; }   // cannot happen: 2 N / HARC_LARGEBIN bound
    }
    // their reads once more, in bin order (k_large_fill); nothing on ordinary data
    uint2 *d_largetab = nullptr; uint64_t *d_mirror = nullptr;
    uint32_t *d_sz0 = nullptr, *d_huge = nullptr; unsigned int *d_nhuge = nullptr; uint32_t nhuge = 0;     // the bins k_compact_huge takes
    uint64_t large_entries = 0;                                   // reads in bins of more than HARC_LARGEBIN reads, both dictionaries
    if (nlarge) {
        uint32_t *sz = nullptr; uint64_t *moff = nullptr;
        RC_TRY(dalloc(c, &d_largetab, nlarge)); RC_TRY(dalloc(c, &sz, (size_t)nlarge + 1)); RC_TRY(dalloc(c, &moff, (size_t)nlarge + 1));
        HIP_TRY(hipMemsetAsync(sz + nlarge, 0, 4, c->stream));
        hipLaunchKernelGGL(k_large_sizes, harc_grid256(nlarge), dim3(256), 0, c->stream, (const unsigned long long *)d_large, nlarge, dict[0].slots, dict[1].slots, sz);
        d_sz0 = sz;
        RC_TRY(dalloc(c, &d_huge, (size_t)nlarge + 1)); RC_TRY(dalloc(c, &d_nhuge, 4));
        HIP_TRY(hipMemsetAsync(d_nhuge, 0, 16, c->stream));
        hipLaunchKernelGGL(k_huge_list, harc_grid256(nlarge), dim3(256), 0, c->stream, (const uint32_t *)sz, nlarge, d_huge, d_nhuge);
        HIP_TRY(hipMemcpyAsync(&nhuge, d_nhuge, 4, hipMemcpyDeviceToHost, c->stream));
        RC_TRY(prim_excl_scan_u32_to_u64(c, sz, moff, (size_t)nlarge + 1));
        uint64_t mtotal = 0;
        HIP_TRY(hipMemcpyAsync(&mtotal, moff + nlarge, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (mtotal > 0xFFFFFFFFull) { harc_set_error("stage I: more than 2^32 reads in large bins"); return HARC_AMD_EINVAL; }
        large_entries = mtotal;
        RC_TRY(dalloc(c, &d_mirror, (size_t)mtotal * W + 1));
        hipLaunchKernelGGL((k_large_fill<W>), wave_grid(nlarge), dim3(64), 0, c->stream, (const unsigned long long *)d_large, nlarge, dict[0].slots, dict[1].slots,
                           (const uint32_t *)dict[0].ids, (const uint32_t *)dict[1].ids, (const uint64_t *)moff, (const uint64_t *)c->d_reads, d_largetab, d_mirror);
        HIP_TRY(hipGetLastError());
    }
    if (P.num_steps <= 0 && K > 1) {                              // (see above)
        const bool lowcov = N && (double)dict[0].nbins > 0.98 * (double)N;
        nsteps = ((K > 16384 || (K >= 2048 && !lowcov)) && large_entries * 50 <= (uint64_t)N) ? 32 : 16;
    }
    const bool backoff = K > 16384 && large_entries * 50 > (uint64_t)N;      // the same inputs: chains that keep losing bids sit rounds out (HARC_BO_FREE)
    if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[stage I] %u bins of more than %u reads hold %llu entries (%.2f %% of the reads): %d steps per super-round\n", nlarge, HARC_LARGEBIN,
                                          (unsigned long long)large_entries, N ? 100.0 * (double)large_entries / (double)N : 0.0, nsteps);
    HIP_TRY(hipEventRecord(e1, c->stream));

    // ---- chain state
    S1Args a; memset(&a, 0, sizeof a);
    a.L = P.readlen; a.maxmatch = P.maxmatch; a.thresh = P.thresh; a.maxsearch = P.maxsearch;
    a.Lp = ((W + 1) / 2) * 64; a.S = nsteps;
    if (backoff) { RC_TRY(dalloc(c, &a.bo, (size_t)K)); HIP_TRY(hipMemsetAsync(a.bo, 0, (size_t)K * 4, c->stream)); }
    a.nsugg_per_seed = HARC_NSUGG;
    // ONE chain (exact mode) takes its seeds from the same descending cursor whatever the look-ahead holds (nobody else claims anything between the
    // hand-out and the use, and the walk skips what it took itself): 64 look-ahead seeds instead of 8 are the same bytes and a third of the
    // super-rounds on low-coverage input, where a walk needs a new seed every few reads
    if (K == 1) a.nsugg_per_seed = 64;
#ifdef HARC_AMD_EXPERIMENTS    // schedule knobs change the archive bytes: never read from the environment by a product build (make EXPERIMENTS=1)
    if (const char *e = getenv("HARC_AMD_NSUGG")) { a.nsugg_per_seed = atoi(e); if (a.nsugg_per_seed < 0) a.nsugg_per_seed = 0; if (a.nsugg_per_seed > HARC_NSUGG) a.nsugg_per_seed = HARC_NSUGG; }
#endif
    for (int l = 0; l < 2; l++) { a.ds[l] = P.dict_start[l]; a.de[l] = P.dict_end[l]; a.kbits[l] = 2 * (P.dict_end[l] - P.dict_start[l] + 1); }
    a.N = N; a.K = K; a.reads = c->d_reads;
    // design (R): after harc_amd_replicate_exchange every rank holds all reads and builds the whole index; the chains are dealt to the ranks
    // four at a time (a workgroup of the main kernel), and every super-round ends with one all-gather of the walked steps
    HarcComm *const cm = (c->replicated && c->comm) ? c->comm : nullptr;
    a.own_mod = cm ? (uint32_t)cm->world : 1u; a.own_rem = cm ? (uint32_t)cm->rank : 0u;
    if (a.own_mod > 64) { harc_set_error("design (R): at most 64 ranks (world %u)", a.own_mod); return HARC_AMD_EINVAL; }     // before any work: the digest compare below holds 64 x 8 words
    for (int l = 0; l < 2; l++) { a.slots[l] = dict[l].slots; a.cap[l] = dict[l].cap; a.ids[l] = dict[l].ids; }
    a.bloom[0] = d_bloom[0]; a.bloom[1] = d_bloom[1]; a.bloom_lines = bloom_lines; a.bloom_nwin[0] = bloom_nwin[0]; a.bloom_nwin[1] = bloom_nwin[1]; a.bloom_mmask = bloom_mmask;
    a.largetab = d_largetab; a.mirror = d_mirror;
    const size_t nwords = (size_t)N / 64 + 2;
    const uint32_t nblk = (K + 255) / 256;
    RC_TRY(dalloc(c, &a.claimed, nwords)); RC_TRY(dalloc(c, &a.bid, (size_t)N + 1)); RC_TRY(dalloc(c, &a.hdr, K));
    RC_TRY(dalloc(c, &a.cnt, (size_t)2 * K * a.Lp)); RC_TRY(dalloc(c, &a.steps, (size_t)K * 64)); RC_TRY(dalloc(c, &a.need, (size_t)K + 8192 + 1024));
    RC_TRY(dalloc(c, &a.reseed_g, 4 + 4 * RESEED_G)); HIP_TRY(hipMemsetAsync(a.reseed_g, 0, (4 + 4 * RESEED_G) * 4, c->stream));
    RC_TRY(dalloc(c, &a.grp_ticket, 4)); HIP_TRY(hipMemsetAsync(a.grp_ticket, 0, 16, c->stream));
    a.nsugg_stride = a.nsugg_per_seed > HARC_NSUGG ? a.nsugg_per_seed : HARC_NSUGG;
    RC_TRY(dalloc(c, &a.seedbuf, (size_t)K * (1 + a.nsugg_stride))); RC_TRY(dalloc(c, &a.needrank, (size_t)K + 16)); RC_TRY(dalloc(c, &a.rmeta, 4)); RC_TRY(dalloc(c, &a.cst2, (size_t)K + 1)); RC_TRY(dalloc(c, &a.sugg, (size_t)K * a.nsugg_stride));
    RC_TRY(dalloc(c, &a.slog, (size_t)N + 1));
    const size_t pg_max = (size_t)N / 64 + ((size_t)K + 1) * PG_CHUNK;        // full pages + one open chunk per chain
    RC_TRY(dalloc(c, &a.pg_rec, pg_max * 64)); RC_TRY(dalloc(c, &a.pg_hdr, pg_max)); RC_TRY(dalloc(c, &a.pg_cur, (size_t)K + 1)); RC_TRY(dalloc(c, &a.pg_count, 4));
    RC_TRY(dalloc(c, &a.cursor, 1)); RC_TRY(dalloc(c, &a.stats, ST_N)); RC_TRY(dalloc(c, &a.coopcnt, HARC_COOPCNT)); HIP_TRY(hipMemsetAsync(a.coopcnt, 0, HARC_COOPCNT * 8, c->stream));
    RC_TRY(dalloc(c, &a.cstat, K)); HIP_TRY(hipMemsetAsync(a.cstat, 0, (size_t)K * 16, c->stream));
    RC_TRY(dalloc(c, &a.cstat_coop, K)); HIP_TRY(hipMemsetAsync(a.cstat_coop, 0, (size_t)K * 16, c->stream)); RC_TRY(dalloc(c, &a.csteps, K)); HIP_TRY(hipMemsetAsync(a.csteps, 0, (size_t)K * 8, c->stream));
    RC_TRY(dalloc(c, &a.dbg, 48)); HIP_TRY(hipMemsetAsync(a.dbg, 0, 48 * 8, c->stream));
    unsigned long long *const dbg_ptr = a.dbg;
    if (!getenv("HARC_AMD_TRACE")) a.dbg = nullptr;
    std::vector<uint16_t> tab = make_probe_table(P);
    uint16_t *d_tab = nullptr; RC_TRY(dalloc(c, &d_tab, tab.size() + 1));
    HIP_TRY(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, c->stream));
    a.probe_tab = d_tab; a.nprobe = (int)tab.size();
    {
        uint32_t *lt = nullptr;
        const size_t nm = (size_t)2 * P.maxmatch * StepsLds<W>::MROW;
        RC_TRY(dalloc(c, &lt, nm + 2 * tab.size() + 2));
        hipLaunchKernelGGL((k_steps_tables<W>), dim3(4), dim3(256), 0, c->stream, a, lt);
        a.lds_tab = lt;
    }
    a.stepcap = HARC_STEP_CAP; a.budget = HARC_SCAN_BUDGET;      // part of the schedule (the oracle has the same constants): fixed in a product build
#ifdef HARC_AMD_EXPERIMENTS
    if (getenv("HARC_AMD_STEPCAP")) a.stepcap = atoi(getenv("HARC_AMD_STEPCAP"));
    if (getenv("HARC_AMD_BUDGET")) a.budget = atoi(getenv("HARC_AMD_BUDGET"));
#endif
    if (a.stepcap < 1) a.stepcap = 1;
    a.weedmin = getenv("HARC_AMD_WEEDMIN") ? atoi(getenv("HARC_AMD_WEEDMIN")) : 3;      // (k_steps_grp never weeds by claim bit unless told to: set below)
    a.lazy = getenv("HARC_AMD_LAZY") ? (atoi(getenv("HARC_AMD_LAZY")) != 0 ? 1 : 0) : 1;
    a.firstmax = (bloom_nwin[0] > 0 && bloom_nwin[1] > 0) ? 64 : 48;     // lines by minimizer: speculative probes share their lines
    if (const char *e = getenv("HARC_AMD_FIRSTMAX")) { const int x = atoi(e); a.firstmax = x < 1 ? 1 : x > 64 ? 64 : x; }
    // successor lists (k_succ): by default where the walk is ONE wave (exact mode, num_chains = 1) -- there the precomputation, twice the lookups of a
    // whole run, is done by a chip that has nothing else to do.  HARC_AMD_SUCC=1 asks for them with any number of chains of the few-chains kernel
    // (tests: the same bytes), =0 never.  They need the lazy counts (a step by list does not fetch its read), every small bin scanned to its end
    // (maxsearch above HARC_LARGEBIN) and 128 bytes per read.
    {
        bool want = getenv("HARC_AMD_SUCC") ? atoi(getenv("HARC_AMD_SUCC")) != 0 : K == 1;
        const size_t sbytes = (size_t)N * 2 * HARC_SUCC_N * sizeof(uint2);
        size_t fr = 0, tot = 0;
        if (want) HIP_TRY(hipMemGetInfo(&fr, &tot));
        if (want && N && quad && !cm && a.lazy && P.maxsearch >= (int)HARC_LARGEBIN && P.thresh <= 127 && a.nprobe <= 4095 && P.maxmatch <= 255 && sbytes < tot / 8) {
            uint2 *sp = nullptr;
            RC_TRY(dalloc(c, &sp, (size_t)N * 2 * HARC_SUCC_N));
            hipLaunchKernelGGL((k_succ<W>), harc_fold256(((uint64_t)N * 2 + 3) / 4), dim3(256), succ_lds_bytes(W, P.maxmatch, a.nprobe), c->stream, a, sp, N);
            HIP_TRY(hipGetLastError());
            a.succ = sp;
            if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[stage I] successor lists of %u reads x 2 orientations (%.1f MB)\n", N, (double)sbytes / 1e6);
        }
    }
    HIP_TRY(hipMemsetAsync(a.claimed, 0, nwords * 8, c->stream));
    HIP_TRY(hipMemsetAsync(a.bid, 0xFF, ((size_t)N + 1) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(a.slog, 0xFF, ((size_t)N + 1) * sizeof(uint2), c->stream));
    HIP_TRY(hipMemsetAsync(a.rmeta, 0, 16, c->stream));
    HIP_TRY(hipMemsetAsync(a.stats, 0, ST_N * 8, c->stream));
    HIP_TRY(hipMemsetAsync(a.need, 0, (size_t)K + 8192 + 1024, c->stream));
    const long long cur0 = (long long)N - 1;
    HIP_TRY(hipMemcpyAsync(a.cursor, &cur0, 8, hipMemcpyHostToDevice, c->stream));
    {   // chains alive at the start: all of them when floor(N/K) > 0, else only chain 0 (reorder.cpp:484-490)
        const unsigned long long act0 = N == 0 ? 0ULL : (N / K > 0 ? (unsigned long long)K : 1ULL);
        HIP_TRY(hipMemcpyAsync(a.stats + ST_ACTIVE, &act0, 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_init_chains, dim3(nblk), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());

    // SLOT_DEAD of the large bins must say "no unclaimed read" from the first super-round on (the seeds of k_init_chains are claimed)
    if (nlarge) hipLaunchKernelGGL((k_compact_bins<W>), wave_grid(nlarge), dim3(64), 0, c->stream, a, (const unsigned long long *)d_large, nlarge, (const uint32_t *)d_sz0);
    if (nhuge) hipLaunchKernelGGL((k_compact_huge<W>), dim3(nhuge), dim3(1024), 0, c->stream, a, (const unsigned long long *)d_large, (const uint32_t *)d_huge, nhuge);
    // ---- rounds
    uint32_t *x_send = nullptr, *x_recv = nullptr; uint32_t x_nper = 0; size_t x_bytes = 0; unsigned long long *x_dig = nullptr;
    std::vector<size_t> x_so, x_sb, x_ro, x_rb;
    if (cm) {
        x_nper = ((((K + 3) / 4) + a.own_mod - 1) / a.own_mod) * 4;                 // chains of one rank's block
        x_bytes = (size_t)x_nper * (8 + 2 * (size_t)nsteps) * 4;
        RC_TRY(dalloc(c, &x_send, x_bytes / 4 + 4)); RC_TRY(dalloc(c, &x_recv, (x_bytes / 4) * a.own_mod + 4));
        x_so.assign(a.own_mod, 0); x_sb.assign(a.own_mod, x_bytes); x_ro.resize(a.own_mod); x_rb.assign(a.own_mod, x_bytes);
        for (uint32_t p = 0; p < a.own_mod; p++) x_ro[p] = (size_t)p * x_bytes;
        RC_TRY(dalloc(c, &x_dig, 8));
    }
    const size_t lds_bytes = steps_lds_bytes(W, P.maxmatch, a.nprobe), lds_bytes_seq = steps_lds_bytes(W, P.maxmatch, a.nprobe, true), lds_bytes_coop = steps_lds_bytes_coop(W, P.maxmatch, a.nprobe);
    const bool prof = P.profile != 0;
    // dense kernels (7 / 8 waves per SIMD) for every launch that is not QUAD (round 2: from 49 152 chains on; with the counts in LDS they pay from
    // 16 385 on: 24 k chains of configs[2] at 1/7 scale +5 %, the 21 k chains of c3sd +3.6 %)
    const bool dense = getenv("HARC_AMD_DENSE") ? atoi(getenv("HARC_AMD_DENSE")) != 0 : K > 16384;     // see HARC_STEPS_WAVES
    // dense launches over mostly single-read bins (more than 88 % distinct first-dictionary k-mers: configs[2] 0.95, configs[3] 0.96): the small bins
    // of a batch one after the other by the whole wave, 8 waves per SIMD; at 190x (configs[4]: 0.81, a quarter of it k-mers with a sequencing
    // error) every other true bin holds several reads, most of them claimed, and scanning them one after the other costs more round trips
    // than the lanes' own scans cost instructions (measured there: 370 against 385 Mreads/s)
    // ... and repeat families put reads of different copies into small bins of several reads as well (c3sd is 18 % slower with the wave-uniform
    // scan, and no count the index has tells it from configs[2]).  Both variants compute the same thing, so a dense run MEASURES: the first
    // eight super-rounds with the wave-uniform scan, the next eight without, the faster one from there on (HARC_AMD_SEQ=0/1 forces one).
    bool seq = getenv("HARC_AMD_SEQ") ? atoi(getenv("HARC_AMD_SEQ")) != 0 : (HARC_SEQ_SCAN && N && (double)dict[0].nbins > 0.88 * (double)N);
    // the specialised form of the dense wave-uniform kernel (k_steps' SPEC): its constants must be this run's parameters
    const bool spec = (getenv("HARC_AMD_SPEC") ? atoi(getenv("HARC_AMD_SPEC")) != 0 : true) && W >= 4 && a.kbits[0] == 64 && a.kbits[1] == 64 && bloom_lines > 0 && bloom_nwin[0] == 17 && bloom_nwin[1] == 17 &&
                      bloom_mmask == 0xFFFFFFFFu && (dict[0].cap >> 34) == 0 && P.maxsearch >= (int)HARC_LARGEBIN && a.firstmax == 64;
    // two chains per wave (k_steps_grp, steps_group.h): the dense SPEC conditions, at most 32 steps and look-ahead seeds per group of 32 lanes.
    // HARC_AMD_GRP=1 asks for it wherever it can run, =0 never (same bytes either way: tests)
#ifdef HARC_AMD_WITH_GRP
    const bool grp_ok = (W == 4 || W == 5) && dense && !quad && spec && !backoff && nsteps <= 16 && a.nsugg_per_seed <= 16 && a.nsugg_stride <= 16 && (dict[0].cap >> 32) == 0;
    bool grp = grp_ok && (getenv("HARC_AMD_GRP") ? atoi(getenv("HARC_AMD_GRP")) != 0 : false);
    if (getenv("HARC_AMD_GRP") && atoi(getenv("HARC_AMD_GRP")) == 2 && !grp) {      // tests: the kernel asked for must be the kernel that runs
        harc_set_error("HARC_AMD_GRP=2: k_steps_grp cannot run here (W %d dense %d quad %d spec %d steps %d look-ahead seeds %d)", W, (int)dense, (int)quad, (int)spec, nsteps, a.nsugg_per_seed);
        return HARC_AMD_EINVAL;
    }
    // lanes per chain: 16 (four chains per wave) for reads of up to 128 bases, 32 beyond (HARC_AMD_GRP_G=32 forces two chains per wave)
    int grp_g = W <= 4 ? 16 : 32;
    if (const char *e = getenv("HARC_AMD_GRP_G")) { if (atoi(e) == 32) grp_g = 32; }
    const int grp_u = 2;                                         // probes per lane and batch (measured with 1, 2 and 4: steps_group.h)
    const size_t lds_bytes_grp = steps_grp_lds_bytes(W, grp_g, true, P.maxmatch, a.nprobe), lds_bytes_grp32 = steps_grp_lds_bytes(W, 32, false, P.maxmatch, a.nprobe);
    // the u32 form behind it (steps_group.h: chains whose counts outgrow 16 bits): during the first batch of the run, and from the batch on in which
    // the first form reports a chain beyond GRP_WIDE_WARN; HARC_AMD_GRP_WIDE=1 launches it always (tests)
    a.grp_wide_warn = GRP_WIDE_WARN; a.grp_wide_limit = GRP_WIDE_LIMIT;
    if (const char *e = getenv("HARC_AMD_GRP_WIDE_LIMIT")) { const int x = atoi(e); if (x >= 1 && x < (int)GRP_WIDE_LIMIT) { a.grp_wide_limit = (uint32_t)x; a.grp_wide_warn = (uint32_t)x / 2; } }      // tests (with HARC_AMD_GRP_WIDE=1): chains change hands at small counts
    // workgroups of it the chip holds at once (HARC_AMD_GRP_GRID: that many per compute unit instead of what the occupancy calculator says; a large
    // number = one workgroup per 8 / 16 chains, no tickets)
    uint32_t grp_resident = 1u << 30;
    if (grp) {
        if constexpr (W == 4 || W == 5) {
            int per_cu = 0;
            hipError_t oe = hipSuccess;
            if (W == 4 && grp_g == 16) { if constexpr (W == 4) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_steps_grp<W, 16, true, 2>, 256, lds_bytes_grp); }
            else oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_steps_grp<W, 32, true, 2>, 256, lds_bytes_grp);
            if (oe != hipSuccess || per_cu < 1) per_cu = 4;
            if (const char *e = getenv("HARC_AMD_GRP_GRID")) { const int x = atoi(e); if (x >= 1) per_cu = x; }
            hipDeviceProp_t pr;
            const int ncu = (hipGetDeviceProperties(&pr, P.device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
            grp_resident = (uint32_t)per_cu * (uint32_t)ncu;
            if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[stage I] k_steps_grp: %d lanes per chain, %d probes per lane and batch, %d workgroups per compute unit x %d\n", grp_g, grp_u, per_cu, ncu);
        }
    }
#else
    const bool grp = false;
    if (getenv("HARC_AMD_GRP") && atoi(getenv("HARC_AMD_GRP")) == 2) { harc_set_error("HARC_AMD_GRP=2: k_steps_grp is not compiled into this library (make GRP=1)"); return HARC_AMD_EINVAL; }
#endif
    if (grp && !getenv("HARC_AMD_WEEDMIN")) a.weedmin = 99;     // with several chains per wave the look at the claim bitmap ahead of the tests is one more trip for everybody: 1090 -> 1066 us per wave
#ifdef HARC_AMD_WITH_GRP
    bool grp_wide_seen = getenv("HARC_AMD_GRP_WIDE") && atoi(getenv("HARC_AMD_GRP_WIDE")) != 0; uint64_t grp_rounds = 0;
#endif
    int seq_probe = (!getenv("HARC_AMD_SEQ") && HARC_SEQ_SCAN && dense && !quad && seq && !grp) ? 0 : 2;      // 0 / 1: the batch being timed, 2: decided
    float seq_ms[2] = { 0, 0 };
    // ... once per context and input shape: a later run over as many reads of the same length with the same schedule starts with the scan the earlier one
    // measured (configs[2]: the eight super-rounds with the slower scan were 3.5 ms of every step; HARC_AMD_SEQ_REMEASURE=1 measures every run)
    const uint64_t seq_key = ((uint64_t)N << 32) ^ ((uint64_t)K << 8) ^ ((uint64_t)nsteps << 1) ^ ((uint64_t)P.readlen << 52) ^ 1u;
    bool seq_measured = false;
    if (seq_probe == 0 && c->s1_seq_key == seq_key && c->s1_seq_choice >= 0 && !(getenv("HARC_AMD_SEQ_REMEASURE") && atoi(getenv("HARC_AMD_SEQ_REMEASURE")) != 0)) { seq = c->s1_seq_choice != 0; seq_probe = 2; }
    uint64_t rounds = 0, launches = 0;
    int coop_forced = getenv("HARC_AMD_COOP_WAVES") ? atoi(getenv("HARC_AMD_COOP_WAVES")) : 0;     // tests: 1, 2 or 4 waves per cooperative workgroup
    if (coop_forced != 1 && coop_forced != 2 && coop_forced != 4) coop_forced = 0;
    int coop_waves = coop_forced ? coop_forced : 4;
    unsigned long long coop_seen = 0;
    double coop_slots = 1024.0;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, P.device) == hipSuccess && pr.multiProcessorCount > 0) coop_slots = 4.0 * pr.multiProcessorCount; }
    // super-rounds between two looks of the host at the counters: 8; 32 where a round is one wave's walk (a look costs as much as two such rounds)
    const int batch = getenv("HARC_AMD_BATCHSYNC") ? atoi(getenv("HARC_AMD_BATCHSYNC")) : (K <= 4 ? 32 : 8);
    // k_reseed by 64 workgroups once a single one has thousands of seeds to hand out per round (HARC_AMD_RESEED_MG=0/1 forces either; same result)
    const bool reseed_mg = getenv("HARC_AMD_RESEED_MG") ? atoi(getenv("HARC_AMD_RESEED_MG")) != 0 : K > 4096;
    // one launch for both while a single workgroup of k_resolve holds all chains (HARC_AMD_FUSED_RR=0: two, as with more chains; same result)
    const bool fused_rr = !reseed_mg && K <= 4u * (64u / (nsteps <= 16 ? 16u : nsteps <= 32 ? 32u : 64u)) && !(getenv("HARC_AMD_FUSED_RR") && atoi(getenv("HARC_AMD_FUSED_RR")) == 0);
    a.reseed_win = RESEED_G * RESEED_NT;
    if (const char *e = getenv("HARC_AMD_RESEED_WIN")) { const int x = atoi(e); if (x >= 1 && x <= RESEED_G * RESEED_NT) a.reseed_win = (uint32_t)x; }     // same seeds, more passes
    if (const char *e = getenv("HARC_AMD_RESEED_STRESS")) { const int x = atoi(e); a.reseed_stress = x < 0 ? 0u : x > 1000 ? 1000u : (uint32_t)x; }     // delays only
    for (;;) {
        if (seq_probe < 2) { seq = seq_probe == 0; HIP_TRY(hipEventRecord(R.eb[0], c->stream)); }
        for (int r = 0; r < batch; r++) {
            hipEvent_t *pair = nullptr;
            if (cm) hipLaunchKernelGGL(k_apply_seed, dim3(nblk), dim3(256), 0, c->stream, a);          // the seeds of the chains other ranks walk
            if (prof) { RC_TRY(R.dense.next_pair(&pair)); HIP_TRY(hipEventRecord(pair[0], c->stream)); }
            if (quad) hipLaunchKernelGGL((k_steps<W, true, false>), dim3((K + 3) / 4), dim3(256), lds_bytes_seq, c->stream, a);
#ifdef HARC_AMD_WITH_GRP
            else if (grp) {
                if constexpr (W == 4 || W == 5) {
                    // a persistent grid: as many workgroups as the chip holds at once (its groups take further chains by ticket), fewer when there are fewer chains
                    const uint32_t cpb = grp_g == 16 ? 16u : 8u, gmax = (K + cpb - 1) / cpb, ggrid = gmax < grp_resident ? gmax : grp_resident;
                    if (W == 4 && grp_g == 16) { if constexpr (W == 4) hipLaunchKernelGGL((k_steps_grp<W, 16, true, 2>), dim3(ggrid), dim3(256), lds_bytes_grp, c->stream, a); }
                    else hipLaunchKernelGGL((k_steps_grp<W, 32, true, 2>), dim3(ggrid), dim3(256), lds_bytes_grp, c->stream, a);
                    if (grp_wide_seen || grp_rounds < (uint64_t)batch) hipLaunchKernelGGL((k_steps_grp<W, 32, false, 1>), dim3((K + 7) / 8), dim3(256), lds_bytes_grp32, c->stream, a);
                    grp_rounds++;
                }
            }
#endif
            else if (dense && seq && spec) hipLaunchKernelGGL((k_steps<W, false, false, true, 4, true, true>), dim3((K + 3) / 4), dim3(256), lds_bytes_seq, c->stream, a);
            else if (dense && seq) hipLaunchKernelGGL((k_steps<W, false, false, true, 4, true>), dim3((K + 3) / 4), dim3(256), lds_bytes_seq, c->stream, a);
            else if (dense) hipLaunchKernelGGL((k_steps<W, false, false, true>), dim3((K + 3) / 4), dim3(256), lds_bytes, c->stream, a);
            else hipLaunchKernelGGL((k_steps<W, false, false>), dim3((K + 3) / 4), dim3(256), lds_bytes, c->stream, a);
            if (prof) HIP_TRY(hipEventRecord(pair[1], c->stream));
            // the steps that have to scan a large bin (none without such bins: the launch is skipped)
            if (nlarge) {
                if (prof) { RC_TRY(R.coop.next_pair(&pair)); HIP_TRY(hipEventRecord(pair[0], c->stream)); }      // timed apart from the main kernel: its scans stream mirrors from L2, another ceiling
                if (coop_waves == 1) hipLaunchKernelGGL((k_steps<W, true, true, false, 1>), dim3(K), dim3(64), lds_bytes_coop, c->stream, a);
                else if (coop_waves == 2) hipLaunchKernelGGL((k_steps<W, true, true, false, 2>), dim3(K), dim3(128), lds_bytes_coop, c->stream, a);
                else hipLaunchKernelGGL((k_steps<W, true, true>), dim3(K), dim3(256), lds_bytes_coop, c->stream, a);
                if (prof) HIP_TRY(hipEventRecord(pair[1], c->stream));
            }
            if (cm) {   // ONE all-gather per super-round: header + walked steps of every chain, from the rank that walked it
                const uint64_t tot = (uint64_t)x_nper * (8 + 2 * (uint64_t)nsteps);
                hipLaunchKernelGGL(k_pack_chains, harc_grid256(tot), dim3(256), 0, c->stream, a, x_send, x_nper);
                const void *sp[1] = { x_send }; void *rp[1] = { x_recv };
                const size_t *sop[1] = { x_so.data() }, *sbp[1] = { x_sb.data() }, *rop[1] = { x_ro.data() }, *rbp[1] = { x_rb.data() };
                RC_TRY(cm->alltoallv(c, 1, sp, sop, sbp, rp, rop, rbp));
                hipLaunchKernelGGL(k_unpack_chains, dim3((unsigned)((tot + 255) / 256), a.own_mod), dim3(256), 0, c->stream, a, (const uint32_t *)x_recv, x_nper);
            }
            if (fused_rr) {
                if (nsteps <= 16) hipLaunchKernelGGL((k_resolve_reseed<16>), dim3(1), dim3(256), 0, c->stream, a);
                else if (nsteps <= 32) hipLaunchKernelGGL((k_resolve_reseed<32>), dim3(1), dim3(256), 0, c->stream, a);
                else hipLaunchKernelGGL((k_resolve_reseed<64>), dim3(1), dim3(256), 0, c->stream, a);
            }
            else if (nsteps <= 16) hipLaunchKernelGGL((k_resolve<16>), dim3((K + 15) / 16), dim3(256), 0, c->stream, a);
            else if (nsteps <= 32) hipLaunchKernelGGL((k_resolve<32>), dim3((K + 7) / 8), dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL((k_resolve<64>), dim3((K + 3) / 4), dim3(256), 0, c->stream, a);
            if (fused_rr) { }
            else if (reseed_mg) hipLaunchKernelGGL(k_reseed_mg, dim3(RESEED_G), dim3(RESEED_NT), 0, c->stream, a, a.reseed_g);
            else if (K <= 4096) hipLaunchKernelGGL((k_reseed<256>), dim3(1), dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL((k_reseed<1024>), dim3(1), dim3(1024), 0, c->stream, a);
            if (nlarge) hipLaunchKernelGGL((k_compact_bins<W>), wave_grid(nlarge), dim3(64), 0, c->stream, a, (const unsigned long long *)d_large, nlarge, (const uint32_t *)d_sz0);
    if (nhuge) hipLaunchKernelGGL((k_compact_huge<W>), dim3(nhuge), dim3(1024), 0, c->stream, a, (const unsigned long long *)d_large, (const uint32_t *)d_huge, nhuge);
            launches++;
        }
        rounds += batch;
        if (seq_probe < 2) HIP_TRY(hipEventRecord(R.eb[1], c->stream));
        HIP_TRY(hipMemcpyAsync(h_stats, a.stats, ST_N * 8, hipMemcpyDeviceToHost, c->stream));
        if (nlarge) HIP_TRY(hipMemcpyAsync(h_stats + ST_N, a.coopcnt, HARC_COOPCNT * 8, hipMemcpyDeviceToHost, c->stream));
        // (every result of a batch lands in the PINNED block h_stats: a copy into pageable memory would hold this thread inside hipMemcpyAsync
        // until the stream has drained -- behind a collective whose peer died, for ever -- and cm->wait() below would never get to poll)
        unsigned int *const h_reseed_timeout = reinterpret_cast<unsigned int *>(h_stats + ST_N + HARC_COOPCNT + 8);
        *h_reseed_timeout = 0;
        if (reseed_mg) HIP_TRY(hipMemcpyAsync(h_reseed_timeout, a.reseed_g + 2, 4, hipMemcpyDeviceToHost, c->stream));
        if (cm) {
            HIP_TRY(hipMemsetAsync(x_dig, 0, 8 * 8, c->stream));
            hipLaunchKernelGGL(k_replica_digest, dim3(1024), dim3(256), 0, c->stream, a, (unsigned long long)nwords, x_dig);
            HIP_TRY(hipMemcpyAsync(h_stats + ST_N + HARC_COOPCNT, x_dig, 8 * 8, hipMemcpyDeviceToHost, c->stream));
            RC_TRY(cm->wait(c, "all-gather of the walked steps"));          // a peer that died shows as a timeout, not as a hang
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipGetLastError());
#ifdef HARC_AMD_WITH_GRP
        if (grp) {
            if (h_stats[ST_WIDE]) grp_wide_seen = true;
            if (h_stats[ST_WIDE + 1]) { harc_set_error("stage I: %llu chains with wide counts were left unwalked by k_steps_grp (super-round %llu)", h_stats[ST_WIDE + 1], (unsigned long long)rounds); return HARC_AMD_EINTERNAL; }
        }
#endif
        if (*h_reseed_timeout) { harc_set_error("stage I: the workgroups of k_reseed_mg did not meet (super-round %llu); HARC_AMD_RESEED_MG=0 uses the single workgroup", (unsigned long long)rounds); return HARC_AMD_EINTERNAL; }
        // what this rank would run next, from its own clock and counters: which scan of the small bins (measured over the first two batches) and how
        // many waves per cooperative workgroup.  Every variant computes the same bytes; design (R) still runs rank 0's choice on all ranks (below), so
        // that a variant that ever differed would differ between RUNS, where the parity tests see it, and not between the replicas of one run
        if (seq_probe < 2) {
            HIP_TRY(hipEventElapsedTime(&seq_ms[seq_probe], R.eb[0], R.eb[1]));
            if (++seq_probe == 2) { seq = seq_ms[0] < seq_ms[1]; seq_measured = true; }
        }
        if (nlarge && coop_forced == 0) {
            // cooperative walks per super-round over the last rounds against the workgroups of four waves the chip holds (4 per CU): well beyond
            // that the kernel is bound by wave slots, and helpers hold most of them idle
            unsigned long long tot = 0;
            for (int k = 0; k < HARC_COOPCNT; k++) tot += h_stats[ST_N + k];
            const double per_round = (double)(tot - coop_seen) / (double)batch;
            coop_seen = tot;
            // measured: 6800 walks per round (c3sd) 417 / 373 / 392 ms with 4 / 2 / 1 waves; 1300 walks (c2d, c2r) 112 / 133 / 194 ms
            // (with the step cap: c3sd 343 / 280 / 268 ms with 4 / 2 / 1 waves, c2r 69 / 79 / 106 ms)
            coop_waves = per_round > 5.0 * coop_slots ? 1 : per_round > 3.0 * coop_slots ? 2 : 4;
        }
        if (cm) {   // the replicas must agree (k_replica_digest): a rank that drifted would hang the others in the next all-gather, or worse
            uint64_t mine[10], all[10 * 64];
            for (int k = 0; k < 7; k++) mine[k] = h_stats[ST_N + HARC_COOPCNT + k];
            mine[7] = h_stats[ST_ACTIVE];
            mine[8] = seq_probe == 2 ? (seq ? 1u : 0u) : 2u; mine[9] = (uint64_t)coop_waves;
            RC_TRY(cm->allgather_u64(c, mine, 10, all));
            static const char *const what[8] = { "claim bitmap", "chain headers", "chains asking for a seed", "cursor", "seeds wanted", "seeds handed out", "look-ahead seeds", "chains alive" };
            for (uint32_t p = 0; p < a.own_mod; p++) for (int k = 0; k < 8; k++) if (all[(size_t)p * 10 + k] != all[k]) {
                harc_set_error("design (R): the replicas of rank 0 and rank %u differ after super-round %llu (%s: %llx / %llx)", p, (unsigned long long)rounds, what[k],
                               (unsigned long long)all[k], (unsigned long long)all[(size_t)p * 10 + k]);
                return HARC_AMD_EINTERNAL;
            }
            if (all[8] < 2u) seq = all[8] != 0;                    // rank 0's choices
            if (coop_forced == 0 && (all[9] == 1u || all[9] == 2u || all[9] == 4u)) coop_waves = (int)all[9];
        }
        if (seq_measured) { c->s1_seq_key = seq_key; c->s1_seq_choice = seq ? 1 : 0; seq_measured = false; }      // (design (R): rank 0's measurement, on every rank)
        if (getenv("HARC_AMD_TRACE")) {
            uint32_t rm[4] = { 0, 0, 0, 0 };
            HIP_TRY(hipMemcpy(rm, a.rmeta, 16, hipMemcpyDeviceToHost));
            fprintf(stderr, "[stage I] round %llu: %llu chains alive%s; extra passes of k_reseed_mg so far %u\n", (unsigned long long)rounds, h_stats[ST_ACTIVE], seq ? " (wave-uniform scan)" : "", rm[3]);
        }
        if (h_stats[ST_ACTIVE] == 0) break;
        if (rounds > (uint64_t)N * 2 + 1024) { harc_set_error("stage I did not converge after %llu rounds", (unsigned long long)rounds); return HARC_AMD_EINTERNAL; }
    }
    HIP_TRY(hipEventRecord(e2, c->stream));

    // ---- per-chain streams -> chain-major output
    uint32_t *nmain = nullptr, *nsing = nullptr, *bmain = nullptr, *bsing = nullptr;
    RC_TRY(dalloc(c, &nmain, (size_t)K + 1)); RC_TRY(dalloc(c, &nsing, (size_t)K + 1)); RC_TRY(dalloc(c, &bmain, (size_t)K + 1)); RC_TRY(dalloc(c, &bsing, (size_t)K + 1));
    HIP_TRY(hipMemsetAsync(nmain, 0, ((size_t)K + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(nsing, 0, ((size_t)K + 1) * 4, c->stream));
    hipLaunchKernelGGL(k_chain_counts, dim3(nblk), dim3(256), 0, c->stream, a.hdr, a.cstat, (const uint4 *)a.cstat_coop, (const uint2 *)a.csteps, (const uint2 *)a.cst2, K, nmain, nsing, a.stats);
    RC_TRY(prim_excl_scan_u32(c, nmain, bmain, (size_t)K + 1));
    RC_TRY(prim_excl_scan_u32(c, nsing, bsing, (size_t)K + 1));
    uint32_t M = 0, S = 0; const unsigned long long nlog = N;
    HIP_TRY(hipMemcpyAsync(&M, bmain + K, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&S, bsing + K, 4, hipMemcpyDeviceToHost, c->stream));
    unsigned int npages = 0;
    HIP_TRY(hipMemcpyAsync(&npages, a.pg_count, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (getenv("HARC_AMD_DIAG")) {     // bookkeeping post-mortem: what the bitmap, the bids and the chain headers say at the end
        std::vector<unsigned long long> hc(nwords); std::vector<uint32_t> hb((size_t)N + 1); std::vector<ChainHdr> hh(K);
        HIP_TRY(hipMemcpy(hc.data(), a.claimed, nwords * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hb.data(), a.bid, ((size_t)N + 1) * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(hh.data(), a.hdr, (size_t)K * sizeof(ChainHdr), hipMemcpyDeviceToHost));
        unsigned long long nunclaimed = 0, nact = 0, npend = 0;
        for (uint32_t k = 0; k < K; k++) { if (hh[k].flags & CH_ACTIVE) nact++; if (hh[k].flags & CH_PREVUNM) npend++; }
        for (unsigned long long i = 0; i < nlog; i++) {
            if ((hc[i >> 6] >> (i & 63)) & 1ULL) continue;
            if (nunclaimed++ < 24) fprintf(stderr, "[diag rank %u] read %llu was never claimed; bid %x\n", a.own_rem, i, hb[i]);
        }
        fprintf(stderr, "[diag rank %u] %llu reads unclaimed; M %u S %u N %u; chains active %llu, with a pending seed %llu; rounds %llu\n", a.own_rem, nunclaimed, M, S, N, nact, npend, (unsigned long long)rounds);
    }
    if ((unsigned long long)M + S != nlog) { harc_set_error("stage I bookkeeping: M=%u S=%u N=%u", M, S, N); return HARC_AMD_EINTERNAL; }
    c->M = M; c->S = S;
    // every read has exactly one record: M + S = N above, S singleton entries found below, and the pages handed out are those the counts ask for
    unsigned long long *d_found = a.stats + ST_N - 1;             // last statistics word: entries of the singleton log
    if (nlog) hipLaunchKernelGGL(k_s1_singles, dim3((unsigned)(nlog / 256 + 1 < 8192 ? nlog / 256 + 1 : 8192)), dim3(256), 0, c->stream, (const uint2 *)a.slog, nlog, K, bsing, c->d_order_s, d_found);
    if ((size_t)npages > pg_max) { harc_set_error("stage I bookkeeping: %u pages handed out, %zu reserved", npages, pg_max); return HARC_AMD_EINTERNAL; }
    if (npages) hipLaunchKernelGGL(k_pages_out, dim3((npages + 3) / 4), dim3(256), 0, c->stream, (const uint2 *)a.pg_rec, (const uint2 *)a.pg_hdr, (uint32_t)npages, (const uint32_t *)nmain, (const uint32_t *)bmain,
                                   c->d_order, c->d_flag, c->d_pos, c->d_rc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_stats, a.stats, ST_N * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_stats[ST_N - 1] != S) { harc_set_error("stage I bookkeeping: %llu singleton records for %u singletons", h_stats[ST_N - 1], S); return HARC_AMD_EINTERNAL; }

    c->C.n_main = M; c->C.n_singleton = S; c->C.unmatched = h_stats[ST_UNMATCHED]; c->C.conflicts = h_stats[ST_CONFLICTS];
    c->C.probes = h_stats[ST_PROBES]; c->C.candidates = h_stats[ST_CANDS]; c->C.useful_probes = h_stats[ST_USEFUL]; c->C.candidates_seq = h_stats[ST_CANDS_SEQ]; c->C.rounds = rounds; c->C.propose_launches = launches;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1)); c->C.index_ms = ms;
    HIP_TRY(hipEventElapsedTime(&ms, e1, e2)); c->C.chain_ms = ms;
    RC_TRY(R.dense.collect_all()); RC_TRY(R.coop.collect_all());
    c->C.propose_ms = R.dense.ms;
    c->C.coop_ms = R.coop.ms; c->C.coop_launches = R.coop.launches;
    c->C.coop_useful_probes = h_stats[ST_COOP_USEFUL]; c->C.coop_candidates_seq = h_stats[ST_COOP_CANDS_SEQ]; c->C.coop_candidates = h_stats[ST_COOP_CANDS];
    c->C.coop_steps = h_stats[ST_COOP_STEPS]; c->C.dense_steps = h_stats[ST_DENSE_STEPS];
#ifdef HARC_TIMING
    {
        unsigned long long d[16];
        HIP_TRY(hipMemcpy(d, a.dbg, sizeof d, hipMemcpyDeviceToHost));
        const double st = (double)(d[6] ? d[6] : 1);
        fprintf(stderr, "[k_steps timing] steps %llu waves-launched %llu; cycles/step: top %.0f pack+rc %.0f probe+smallscan %.0f coop %.0f bid %.0f update %.0f\n",
                d[6], d[7], d[0] / st, d[1] / st, d[2] / st, d[3] / st, d[4] / st, d[5] / st);
        fprintf(stderr, "[k_steps timing] per step: batches %.2f  max-lane slot iterations %.2f  max-lane scan iterations %.2f  total scan iterations %.2f  lanes probing %.1f\n",
                d[10] / st, d[8] / st, d[9] / st, d[11] / st, d[12] / st);
    }
#endif
#ifdef HARC_GRP_STATS
    if (a.dbg && grp) {
        unsigned long long d[24];
        HIP_TRY(hipMemcpy(d, dbg_ptr, sizeof d, hipMemcpyDeviceToHost));
        const double w = (double)(d[0] ? d[0] : 1);
        fprintf(stderr, "[k_steps_grp] us per wave by phase: take %.1f, rows %.1f, keys + bitmap issue %.1f, bitmap wait + table %.1f, candidates %.1f, step end + seeds %.1f, finish %.1f\n", d[16] / w / 100.0, d[17] / w / 100.0, d[18] / w / 100.0, d[19] / w / 100.0, d[20] / w / 100.0, d[21] / w / 100.0, d[22] / w / 100.0);
        fprintf(stderr, "[k_steps_grp] waves %llu over %llu launches; per wave: slots %.1f, live groups per slot %.2f, slots with a take %.1f, with a finish %.1f, with a hit %.1f, candidate turns %.1f; wave life %.1f us mean, %.1f us longest (100 MHz clock)\n",
                d[0], (unsigned long long)launches, d[1] / w, (double)d[2] / (double)(d[1] ? d[1] : 1), d[3] / w, d[4] / w, d[6] / w, d[5] / w, d[7] / w / 100.0, d[8] / 100.0);
    } else
#endif
    if (a.dbg) {
        unsigned long long d[48];
        HIP_TRY(hipMemcpy(d, dbg_ptr, sizeof d, hipMemcpyDeviceToHost));
#ifdef HARC_SKETCH_STATS
        fprintf(stderr, "[sketch] Hamming tests of the cooperative scans: %llu; passed %llu (%.3f %%); a 16-base sketch would have rejected %llu (%.1f %%), a 32-base sketch %llu (%.1f %%)\n", d[44], d[47], 100.0 * (double)d[47] / (double)(d[44] ? d[44] : 1),
                d[45], 100.0 * (double)d[45] / (double)(d[44] ? d[44] : 1), d[46], 100.0 * (double)d[46] / (double)(d[44] ? d[44] : 1));
#endif
        const double stp = (double)(d[4] ? d[4] : 1);
        fprintf(stderr, "[k_steps] steps walked %llu (of them kept %u): per step batches %.2f, cooperative bin scans %.2f, their 64-entry chunks %.2f, chunk x probe tests %.2f; steps without a hit %.3f\n", d[4], N, d[5] / stp, d[0] / stp, d[1] / stp, d[3] / stp, d[2] / stp);
        fprintf(stderr, "[k_steps] worst walk of the run: %llu chunks, %llu chunk x probe tests, %llu bin scans\n", d[6], d[7], d[8]);
        fprintf(stderr, "[k_steps] walks of the main kernel: %llu (%.0f per super-round); %llu of them end in front of a large bin (%.1f %%), %llu of those without a step walked (%.1f %% of all walks)\n",
                d[13], (double)d[13] / (double)(rounds ? rounds : 1), d[15], 100.0 * (double)d[15] / (double)(d[13] ? d[13] : 1), d[14], 100.0 * (double)d[14] / (double)(d[13] ? d[13] : 1));
        if (d[9]) fprintf(stderr, "[k_steps] cooperative walks: %llu (%.1f per super-round), %.2f steps each, mean %.1f us, longest %.1f us (100 MHz wall clock)\n",
                          d[9], (double)d[9] / (double)(rounds ? rounds : 1), (double)d[12] / (double)d[9], (double)d[10] / (double)d[9] / 100.0, (double)d[11] / 100.0);
        if (d[9] && d[16]) {                                         // -DHARC_COOP_TRACE builds
            fprintf(stderr, "[k_steps] cooperative walks, mean us by phase: setup %.1f, consensus rows %.1f, probes + small bins %.1f, large-bin scans %.1f, bid + update %.1f, tail %.1f\n",
                    d[16] / (double)d[9] / 100.0, d[17] / (double)d[9] / 100.0, d[18] / (double)d[9] / 100.0, d[19] / (double)d[9] / 100.0, d[20] / (double)d[9] / 100.0, d[21] / (double)d[9] / 100.0);
            fprintf(stderr, "[k_steps] large-bin scans by live entries of the bin: <= 64: %llu (%.2f us each), <= 256: %llu (%.2f), <= 1024: %llu (%.2f), more: %llu (%.2f)\n",
                    d[36], d[40] / (double)(d[36] ? d[36] : 1) / 100.0, d[37], d[41] / (double)(d[37] ? d[37] : 1) / 100.0, d[38], d[42] / (double)(d[38] ? d[38] : 1) / 100.0, d[39], d[43] / (double)(d[39] ? d[39] : 1) / 100.0);
            fprintf(stderr, "[k_steps] cooperative walks by duration (25 us bins, last = longer):");
            for (int k = 0; k < 12; k++) fprintf(stderr, " %llu", d[24 + k]);
            fprintf(stderr, "\n");
        }
    }

    harc_pool_release(c, mark_results);                          // stage II starts right above the results
    run_scope.keep();
    c->have_s1 = true;
    return HARC_AMD_OK;
}

template <int W> static int orient_w(harc_amd_ctx *c, const uint64_t *reads, const uint32_t *order, const uint8_t *rc, uint32_t m, uint64_t *out)
{
    if (!m) return HARC_AMD_OK;
    { const uint32_t per_block = 4u * (64u / W); hipLaunchKernelGGL((k_orient<W>), harc_fold256(((uint64_t)m + per_block - 1) / per_block), dim3(256), 0, c->stream, reads, order, rc, m, c->P.readlen, out); }
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}
int s1_orient(harc_amd_ctx *c, const uint64_t *reads, const uint32_t *order, const uint8_t *rc, uint32_t m, uint64_t *out)
{
    switch (c->W) {
    case 1: return orient_w<1>(c, reads, order, rc, m, out);
    case 2: return orient_w<2>(c, reads, order, rc, m, out);
    case 3: return orient_w<3>(c, reads, order, rc, m, out);
    case 4: return orient_w<4>(c, reads, order, rc, m, out);
    case 5: return orient_w<5>(c, reads, order, rc, m, out);
    case 6: return orient_w<6>(c, reads, order, rc, m, out);
    case 7: return orient_w<7>(c, reads, order, rc, m, out);
    default: return orient_w<8>(c, reads, order, rc, m, out);
    }
}

int stage1_run(harc_amd_ctx *c)
{
    switch (c->W) {
    case 1: return stage1_run_w<1>(c);
    case 2: return stage1_run_w<2>(c);
    case 3: return stage1_run_w<3>(c);
    case 4: return stage1_run_w<4>(c);
    case 5: return stage1_run_w<5>(c);
    case 6: return stage1_run_w<6>(c);
    case 7: return stage1_run_w<7>(c);
    default: return stage1_run_w<8>(c);
    }
}

int stage1_make_oriented(harc_amd_ctx *c, uint32_t i0, uint32_t i1)
{
    if (!c->d_oreads) RC_TRY(dalloc(c, &c->d_oreads, (size_t)c->M * c->W + 1));
    if (i1 > c->M) i1 = c->M;
    if (i0 >= i1) return HARC_AMD_OK;
    return s1_orient(c, c->d_reads, c->d_order + i0, c->d_rc + i0, i1 - i0, c->d_oreads + (size_t)i0 * c->W);
}
