// stage2.hip -- HARC stage II (reference-delta encoding) for gfx950.
//
// Reference: src/encoder.cpp.  The per-thread sequential contig loop (:219-441) becomes a fully data-parallel pipeline over
// ONE global column coordinate: every reordered read i gets gstart[i] = the column of its first base in the concatenation
// of all contig consensi (a prefix sum of the shift bytes, a contig head advancing by readlen).  Then
//   buildcontig  :619-652  -> k_consensus   (one thread per column, majority over the reads covering it, ties A<C<G<T)
//   realignment  :231-418  -> k_realign_propose (one thread per window start, 4 dictionary probes, 3-bit XOR+popcount) +
//                             atomicMin of the (column, direction, dictionary) tuple per candidate read: the sequential
//                             first-come claim of the reference at num_thr=1 is exactly the minimum tuple
//   list.insert  :310-313  -> a merge by rank (two binary searches) of the reads with the accepted candidates
//   writecontig  :654-717  -> k_noise<W, false/true> (sizes, then noise, noisepos, pos, order routing, rc) on packed words
//   packbits     :512-616  -> k_pack2_bytes / k_pack1_bytes
// Shard e of num_thr owns reordered reads [e*q, (e+1)*q) (:171-180); its streams are slices of the global arrays.
#include "devutil.h"
#include <time.h>
#include <algorithm>
#include <stdlib.h>

#define TUPLE_NONE 0xFFFFFFFFFFFFFFFFULL

struct S2Args {
    int L, W, W3, thresh_s, maxsearch;
    int ds[2], de[2], kbits[2];
    uint32_t M, S, T, q, nC;
    uint64_t total;                    // total consensus columns
    const uint64_t *oreads;            // M x W oriented reads
    const uint8_t *flag, *pos, *rc;
    const uint32_t *order;
    const uint64_t *cand3;             // T x W3 : singletons then N reads, 3-bit
    const uint32_t *cand_order;        // T
    uint8_t *head;                     // M
    uint64_t *gstart;                  // M
    uint8_t *cons;                     // total bytes, A0 C1 G2 T3
    const uint32_t *chead;             // nC: read index of each contig head
    const HashSlot *slots[2]; uint64_t cap[2]; const uint32_t *ids[2];
    unsigned long long *best;          // T
    unsigned long long *bestbin[2];    // k_realign_big: best[] once more in the order of ids[l] (refreshed before every window pass): a look reads 64 claims in one line instead of 64 lines
    const uint32_t *bloom[2]; int bloom_shift[2];   // one-hash bitmap over the keys of each dictionary (16 bits per key): most windows match nothing
    int bloom_lbits, bloom_nwin;                    // combined bitmap (k_bloom4_set): log2 of its 64-byte lines; minimizer windows per key (0: lines hashed from the key)
    const uint64_t *cons2;             // consensus, 2-bit code A0 G1 C2 T3, 32 columns per word (k_pack_cons2)
    const uint64_t *cand2, *candN;     // T x W: the candidates in the same 2-bit code (N -> 0) and their N masks (both bits of the field set)
    uint4 *events;                     // probes that hit a bin larger than maxsearch: {tuple lo, tuple hi, dict, slot index lo} (+ slot hi in w>>?)
    unsigned int *nevents; uint32_t maxevents;
    int trace;                         // HARC_AMD_TRACE: count the events that look in a window pass
    // stage II partitioned over the ranks of a design-(R) run (stage2_run): this rank owns the consensus columns [col0, col1) -- whole encoder shards,
    // so whole contigs -- and launches its tile kernels from tile `tile_base` on; on one GPU: [0, total), 0
    uint64_t col0, col1; uint32_t tile_base;
    const uint64_t *evwin;             // k_realign_big: the 3-bit window words of every event (k_ev_windows), W3 per event
    // k_realign_block, passes of a million events and more: the bin-ordered claims COMPACTED (k_bestbin_refresh_bins) -- of every bin that has an event looking in
    // this pass, only the entries not claimed before the EARLIEST such event (nobody who looks can see the others): cbb = their claims, cpos = their positions in
    // the bin, ccnt[first entry of the bin] = how many.  Deep in a drained bin a window of 1000 visible entries spans tens of thousands of claimed ones.
    unsigned long long *cbb[2]; uint32_t *cpos[2], *ccnt[2]; unsigned long long *tminbin[2]; int compact;
    int binmax_on;                     // (HARC_AMD_S2_RANGE=0: off -- every event behind the earliest moved claim looks again, as before round 5; tests)
    unsigned long long *binmax[2];     // window passes: per bin, (pass stamp, the LATEST tuple a claim of the bin was moved away from in that pass) -- see EV_TBITS
};

// ---------------------------------------------------------------------------------------------- small device helpers
__device__ __forceinline__ int base2_at(const uint64_t *r, int j) { return (int)((r[j >> 5] >> (2 * (j & 31))) & 3); }     // packed code A0 G1 C2 T3
__device__ __forceinline__ int pc_to_idx(int pc) { return ((pc & 1) << 1) | (pc >> 1); }                                   // -> A0 C1 G2 T3
__device__ __forceinline__ int c3_at(const uint64_t *r, int W3, int j)
{
    const int off = 3 * j, wi = off >> 6, sh = off & 63;
    uint64_t v = r[wi] >> sh;
    if (sh > 61 && wi + 1 < W3) v |= r[wi + 1] << (64 - sh);
    return (int)(v & 7);
}
__device__ __forceinline__ int c3_to_idx5(int c3) { return c3 == 0 ? 0 : c3 == 4 ? 1 : c3 == 2 ? 2 : c3 == 6 ? 3 : 4; }   // A C G T N
__device__ __forceinline__ int idx_to_c3(int idx) { return idx == 0 ? 0 : idx == 1 ? 4 : idx == 2 ? 2 : 6; }
__device__ __forceinline__ int comp5(int b) { return b == 4 ? 4 : 3 - b; }
// enc_noise (encoder.cpp:751-771), rows = consensus base A C G T, columns = read base A C G T N
__device__ __forceinline__ char enc_noise(int ref, int rd)
{
    const unsigned tabA = 0x32100, tabC = 0x32100, tabG = 0x30021, tabT = 0x30012;   // nibble rd -> code
    const unsigned t = ref == 0 ? tabA : ref == 1 ? tabC : ref == 2 ? tabG : tabT;
    return (char)('0' + ((t >> (4 * rd)) & 0xF));
}
// largest i in [0,n) with a[i] <= x, or -1
__device__ __forceinline__ long long ub_le(const uint64_t *a, long long n, uint64_t x)
{
    long long lo = -1, hi = n;                                   // a[lo] <= x < a[hi]
    while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// ---------------------------------------------------------------------------------------------- candidates (3-bit store)
// singleton r: 2-bit read -> std::bitset<3L> words (encoder.cpp:815-821). One thread per (read, word).
__global__ void k_cand3_from2(const uint64_t *reads2, const uint32_t *gather, uint32_t n, int L, int W, int W3, uint64_t *out)
{
    const size_t gid = harc_gid();
    if (gid >= (size_t)n * W3) return;
    const uint32_t i = (uint32_t)(gid / W3); const int w = (int)(gid % W3);
    const uint64_t *r = reads2 + (size_t)(gather ? gather[i] : i) * W;
    uint64_t v = 0;
    const int b0 = (64 * w) / 3, b1 = (64 * w + 63) / 3;
    for (int b = b0; b <= b1 && b < L; b++) {
        const uint64_t c3 = (uint64_t)(2 * base2_at(r, b));      // A0 G1 C2 T3 -> A0 G2 C4 T6
        const int sh = 3 * b - 64 * w;
        v |= sh >= 0 ? (c3 << sh) : (c3 >> (-sh));
    }
    out[gid] = v;
}
__global__ void k_cand_order(const uint32_t *order_s, uint32_t S, uint32_t T, uint32_t *out)
{
    const uint32_t i = harc_gid32();
    if (i >= T) return;
    out[i] = i < S ? order_s[i] : i - S;                          // encoder.cpp:865-870
}
__global__ void k_key3(const uint64_t *cand3, uint32_t T, int W3, int off, int nbits, uint64_t *keys, uint32_t *ids)
{
    const uint32_t i = harc_gid32();
    if (i >= T) return;
    const uint64_t *r = cand3 + (size_t)i * W3;
    const int wi = off >> 6, sh = off & 63;
    uint64_t v = r[wi] >> sh;
    if (sh && wi + 1 < W3) v |= r[wi + 1] << (64 - sh);
    if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
    keys[i] = v; ids[i] = i;
}
__global__ void k_count_big_bins(const HashSlot *slots, uint64_t cap, uint32_t maxsearch, unsigned long long *out)
{
    const uint64_t i = harc_gid();
    if (i >= cap) return;
    if ((slots[i].count & SLOT_CNT_MASK) > maxsearch) atomicAdd(out, 1ULL);
}

// ---------------------------------------------------------------------------------------------- contig structure
// contig heads: flag '0', shard starts (encoder.cpp:171-180) -- step 1
__global__ void k_heads1(const uint8_t *flag, uint32_t M, uint32_t q, uint8_t *head, uint32_t *hidx)
{
    const uint32_t i = harc_gid32();
    if (i >= M) return;
    const uint8_t h = (flag[i] == '0' || (i % q) == 0) ? 1 : 0;
    head[i] = h; hidx[i] = h ? i : 0;
}
// step 2: a contig is cut once it holds 10,000,001 reads (encoder.cpp:226 `list_size>10000000`)
__global__ void k_heads2(uint32_t M, uint8_t *head, const uint32_t *hmax, uint32_t *hidx)
{
    const uint32_t i = harc_gid32();
    if (i >= M) return;
    const uint32_t r = i - hmax[i];
    if (r > 0 && (r % 10000001u) == 0) head[i] = 1;
    hidx[i] = head[i] ? 1u : 0u;                                  // reused as the flag array for the contig-id scan
}
__global__ void k_col_steps(const uint8_t *head, const uint8_t *pos, uint32_t M, int L, uint64_t *d)
{
    const uint32_t i = harc_gid32();
    if (i >= M) return;
    d[i] = head[i] ? (i == 0 ? 0 : (uint64_t)L) : (uint64_t)pos[i];
}
__global__ void k_contig_heads(const uint8_t *head, const uint32_t *cid, uint32_t M, uint32_t *chead)
{
    const uint32_t i = harc_gid32();
    if (i >= M) return;
    if (head[i]) chead[cid[i]] = i;
}

// per contig: end column, and bit 63 = "no realignment" (the last contig of a shard, encoder.cpp:438-441)
__global__ void k_contig_info(S2Args s, unsigned long long *cinfo)
{
    const uint32_t k = harc_gid32();
    if (k >= s.nC) return;
    const bool lastc = (k + 1 == s.nC) || (s.chead[k + 1] % s.q) == 0;
    const unsigned long long cend = (k + 1 == s.nC) ? s.total : s.gstart[s.chead[k + 1]];
    cinfo[k] = cend | ((unsigned long long)lastc << 63);
}

// buildcontig (encoder.cpp:619-652): column x <- first strict maximum over A,C,G,T of the reads covering it.
// A workgroup owns a tile of 2048 columns, a thread a strip of 8.  The reads are sorted by their first column, so the ones that touch
// the tile are one contiguous range: they go through LDS in pieces of CCHUNK (one coalesced pass over the reads instead of every strip
// fetching its ~13 reads through L2: 10x the bytes), and a strip finds its own reads in a piece by binary search over their
// tile-relative starts.  Byte written: base | 4 when a realignment window may start at x (fits in its contig, encoder.cpp:252, and the
// contig is not the last of its shard).
#define CSTRIP 8
#define CTILE (256 * CSTRIP)
#define CCHUNK 512
// first and last read that touch tile t (a binary search per tile here, not two dependent ones at the top of every workgroup)
__global__ void k_consensus_tiles(S2Args s, uint32_t ntiles, uint32_t *tlo, uint32_t *thi)
{
    const uint32_t t = harc_gid32();
    if (t >= ntiles) return;
    const uint64_t X0 = (uint64_t)(t + s.tile_base) * CTILE, X1 = X0 + CTILE < s.total ? X0 + CTILE : s.total;
    thi[t] = (uint32_t)ub_le(s.gstart, (long long)s.M, X1 - 1);             // last read starting inside or before the tile (every column is covered: >= 0)
    tlo[t] = X0 >= (uint64_t)s.L ? (uint32_t)(ub_le(s.gstart, (long long)s.M, X0 - (uint64_t)s.L) + 1) : 0u;   // first read that reaches column X0
}
__global__ __launch_bounds__(256) void k_consensus(S2Args s, const uint32_t *cid, const unsigned long long *cinfo, int want_windows, const uint32_t *tlo, const uint32_t *thi)
{
    extern __shared__ uint64_t cl_words[];                        // [CCHUNK][W] read words, then CCHUNK tile-relative starts
    int *const cl_g = reinterpret_cast<int *>(cl_words + (size_t)CCHUNK * s.W);
    const uint64_t X0 = (uint64_t)(blockIdx.x + s.tile_base) * CTILE;
    const int L = s.L, W = s.W;
    const long long ilo = tlo[blockIdx.x], ihi = thi[blockIdx.x];
    const int x0r = (int)threadIdx.x * CSTRIP;                   // tile-relative first column of the strip
    const uint64_t x0 = X0 + (uint64_t)x0r;
    const bool mine = x0 < s.total;
    const int xlr = (x0 + CSTRIP - 1 < s.total ? x0r + CSTRIP - 1 : (int)(s.total - 1 - X0));   // last column of the strip
    // Column counts: four 32-bit counters per column (rows A C G T, reorder.cpp order of ties) fed from four 8-BIT counters packed into one
    // register per column (in packed-code order A G C T: the 2-bit code is the byte index, no recoding per base), which are emptied into the
    // wide ones before any of them can reach 256.  A read adds to all columns of the strip from ONE 16-bit window of its words (round 2:
    // per base a word select, a shift, a recode and four compare-adds: 22.6 G vector instructions per launch at configs[2], 34 ms).
    uint32_t cnt[CSTRIP][4], pk[CSTRIP];
#pragma unroll
    for (int c = 0; c < CSTRIP; c++) { cnt[c][0] = cnt[c][1] = cnt[c][2] = cnt[c][3] = 0; pk[c] = 0; }
    int since = 0;                                                // reads added to pk since it was last emptied
    auto flush = [&]() {
#pragma unroll
        for (int c = 0; c < CSTRIP; c++) {
            cnt[c][0] += pk[c] & 0xFFu; cnt[c][2] += (pk[c] >> 8) & 0xFFu; cnt[c][1] += (pk[c] >> 16) & 0xFFu; cnt[c][3] += pk[c] >> 24;   // code A0 G1 C2 T3 -> row A0 C1 G2 T3
            pk[c] = 0;
        }
        since = 0;
    };
    long long ilast = ilo - 1;                                    // last read starting at or before x0
    for (long long base = ilo; base <= ihi; base += CCHUNK) {
        const int nch = (int)(ihi + 1 - base < CCHUNK ? ihi + 1 - base : CCHUNK);
        __syncthreads();
        for (int k = threadIdx.x; k < nch * W; k += 256) cl_words[k] = s.oreads[(size_t)base * W + k];
        for (int k = threadIdx.x; k < nch; k += 256) cl_g[k] = (int)((long long)s.gstart[base + k] - (long long)X0);
        __syncthreads();
        if (!mine) continue;
        // reads of the piece that cover a column of the strip: start in (x0r - L, xlr]
        int lo = -1, hi = nch;                                    // cl_g[lo] <= x0r - L < cl_g[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cl_g[mid] <= x0r - L) lo = mid; else hi = mid; }
        const int first = hi;
        lo = first - 1; hi = nch;                                 // cl_g[lo] <= xlr < cl_g[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cl_g[mid] <= xlr) lo = mid; else hi = mid; }
        const int last = lo;
        for (int k = first; k <= last; k++) {
            const int g = cl_g[k];
            if (g <= x0r) ilast = base + k;
            const uint64_t *r = cl_words + (size_t)k * W;
            const int o0 = x0r - g;                               // offset of column x0 inside the read (may be negative)
            // the 2 CSTRIP bits of the read at the strip's columns: bases o0 .. o0 + CSTRIP - 1 (those before the read's start come out of the shift as zeros and are masked)
            const int j0 = o0 < 0 ? 0 : o0;
            const int w0 = j0 >> 5, sh = 2 * (j0 & 31);
            const uint64_t wa = r[w0], wb = (w0 + 1 < W) ? r[w0 + 1] : 0;
            uint32_t bits = (uint32_t)(sh ? ((wa >> sh) | (wb << (64 - sh))) : wa) & 0xFFFFu;
            if (o0 < 0) bits <<= 2 * (-o0);                       // column c holds base o0 + c: the read starts inside the strip
            // columns of the strip the read covers: c >= -o0, o0 + c < L, x0r + c <= xlr
            const int c_lo = o0 < 0 ? -o0 : 0, c_hi = (L - o0 < xlr - x0r + 1 ? L - o0 : xlr - x0r + 1);      // [c_lo, c_hi)
            uint32_t vm = c_hi >= CSTRIP ? 0xFFu : ((1u << (c_hi > 0 ? c_hi : 0)) - 1u);
            vm &= ~((1u << c_lo) - 1u);
#pragma unroll
            for (int c = 0; c < CSTRIP; c++) pk[c] += ((vm >> c) & 1u) << (8u * ((bits >> (2 * c)) & 3u));
            if (++since == 255) flush();
        }
    }
    flush();
    if (!mine) return;
    uint32_t kprev = HARC_NONE; unsigned long long cend = 0; bool lastc = true;
    long long i = ilast;
    uint8_t outb[CSTRIP];
#pragma unroll
    for (int c = 0; c < CSTRIP; c++) {
        const uint64_t x = x0 + c;
        outb[c] = 0;
        if (x >= s.total) continue;
        uint32_t mx = 0; int ind = 0;
        if (cnt[c][0] > mx) { mx = cnt[c][0]; ind = 0; }
        if (cnt[c][1] > mx) { mx = cnt[c][1]; ind = 1; }
        if (cnt[c][2] > mx) { mx = cnt[c][2]; ind = 2; }
        if (cnt[c][3] > mx) { mx = cnt[c][3]; ind = 3; }
        int valid = 0;
        if (want_windows) {
            while (i + 1 < (long long)s.M && s.gstart[i + 1] <= x) i++;
            const uint32_t k = cid[i] + (uint32_t)s.head[i] - 1u;    // cid = exclusive scan of the head flags: a head's own contig index
            if (k != kprev) { const unsigned long long ci = cinfo[k]; kprev = k; cend = ci & ~(1ULL << 63); lastc = (ci >> 63) != 0; }
            valid = (!lastc && x + (uint64_t)L <= cend) ? 1 : 0;
        }
        outb[c] = (uint8_t)(ind | (valid << 2));
    }
    if (x0 + CSTRIP <= s.total) {
        uint2 w;
        w.x = (uint32_t)outb[0] | ((uint32_t)outb[1] << 8) | ((uint32_t)outb[2] << 16) | ((uint32_t)outb[3] << 24);
        w.y = (uint32_t)outb[4] | ((uint32_t)outb[5] << 8) | ((uint32_t)outb[6] << 16) | ((uint32_t)outb[7] << 24);
        *reinterpret_cast<uint2 *>(s.cons + x0) = w;
    } else {
        for (int c = 0; c < CSTRIP && x0 + c < s.total; c++) s.cons[x0 + c] = outb[c];
    }
}

// exact key -> (start,count) lookup in a bucketed table (64 B = 4 slots, overflow flag in slot 0: see k_table_insert); two slots at a time
__device__ __forceinline__ bool dict_lookup_b(const HashSlot *tab, uint64_t cap, uint64_t key, uint32_t *start, uint32_t *count)
{
    key = key_scramble(key);                                      // what the table stores (harc_dict_build)
    uint64_t sl = bucket_slot(key, cap);
    for (;;) {
        const uint4 r0 = *reinterpret_cast<const uint4 *>(&tab[sl]), r1 = *reinterpret_cast<const uint4 *>(&tab[sl + 1]);
        if (r0.w == 0) return false;
        if (((uint64_t)r0.x | ((uint64_t)r0.y << 32)) == key) { *start = r0.z; *count = r0.w; return true; }
        if (r1.w == 0) return false;
        if (((uint64_t)r1.x | ((uint64_t)r1.y << 32)) == key) { *start = r1.z; *count = r1.w; return true; }
        const uint4 r2 = *reinterpret_cast<const uint4 *>(&tab[sl + 2]), r3 = *reinterpret_cast<const uint4 *>(&tab[sl + 3]);
        if (r2.w == 0) return false;
        if (((uint64_t)r2.x | ((uint64_t)r2.y << 32)) == key) { *start = r2.z; *count = r2.w; return true; }
        if (r3.w == 0) return false;
        if (((uint64_t)r3.x | ((uint64_t)r3.y << 32)) == key) { *start = r3.z; *count = r3.w; return true; }
        if (!(r0.w & SLOT_OVF)) return false;
        sl += 4; if (sl >= cap) sl = 0;
    }
}
__global__ void k_bloom_set(const uint64_t *keys, uint32_t n, uint32_t *bloom, int shift)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t b = mix64(keys[i] ^ 0x9E3779B97F4A7C15ULL) >> shift;
    atomicOr(&bloom[b >> 5], 1u << (b & 31));
}
// Combined bitmap of both dictionaries and both orientations, one 4-bit entry per hash position (plane l: a key of dictionary l hashes
// here; plane 2 + l: the reverse complement of such a key does), two entries of ONE 32-bit word per key.  When both windows have the same
// width, the four probes of encoder.cpp:262-410 at a window start look up four k-mers of the consensus -- and over all window starts
// each k-mer of the consensus is looked up four times.  With this bitmap it is hashed and looked up once.
// The 64-byte line of a key: by its minimizer (the 15-mer inside it with the smallest hash) when nwin > 0 -- consecutive k-mers of the
// consensus share it, and k_realign_propose1 looks up 3.6 G consecutive k-mers at configs[2] -- else hashed from the key.  (A 10-mer
// minimizer compared as it is: every value occurs hundreds of times in a genome, the lines of a 26x data set were saturated.)  Word and
// the two 4-bit entries inside the line come from a hash of the whole key.
#ifndef BLOOM4_M
#define BLOOM4_M 15
#endif
// (The bitmap is addressed by the k-mer in a 2-BIT code of its own -- A0 C1 G2 T3, the consensus bytes & 3, base b at bits 2b: the consensus holds
// no N, so a dictionary key with an N in it can match nothing and is left out, and a 21-mer is 42 bits instead of 63: the m-mer hash is one
// 32-bit multiplication, the roll from one window to the next a 64-bit shift.  k_realign_propose1 hashes 3.6 G of them at configs[2].)
__device__ __forceinline__ uint32_t bloom4_mmer(uint64_t fk)
{
    const uint32_t h = ((uint32_t)fk & ((1u << (2 * BLOOM4_M)) - 1u)) * 0x9E3779B1u;      // the low 2 M bits: M bases
    return h ^ (h >> 15);
}
__device__ __forceinline__ uint32_t bloom4_minimizer(uint64_t fk, int nwin)
{
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < nwin; i++) { const uint32_t x = bloom4_mmer(fk >> (2 * i)); best = x < best ? x : best; }
    return best;
}
__device__ __forceinline__ void bloom4_pos(uint64_t fk, uint32_t minz, int nwin, int lbits, uint32_t *word, int *sh0, int *sh1)
{
    uint32_t kh = ((uint32_t)fk * 0x9E3779B1u) ^ ((uint32_t)(fk >> 32) * 0x85EBCA77u);
    kh ^= kh >> 15; kh *= 0xC2B2AE3Du; kh ^= kh >> 13;
    uint32_t lh = nwin > 0 ? minz * 0x9E3779B1u : kh * 0x165667B1u;
    lh ^= lh >> 15; lh *= 0x85EBCA77u; lh ^= lh >> 13;
    *word = ((lh >> (32 - lbits)) << 4) | ((kh >> 6) & 15u);
    *sh0 = 4 * (int)(kh & 7u); *sh1 = 4 * (int)((kh >> 3) & 7u);
}
__global__ void k_bloom4_set(const uint64_t *keys, uint32_t n, uint32_t *bloom, int lbits, int nwin, int l, int nb)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t k = keys[i];
    uint64_t f = 0, r = 0;                                        // the k-mer and its reverse complement in the 2-bit code
    for (int b = 0; b < nb; b++) {
        const uint32_t c3 = (uint32_t)(k >> (3 * b)) & 7u;
        if (c3 & 1u) return;                                      // an N: no window of the consensus equals this key
        const uint64_t idx = ((c3 >> 2) & 1u) | (c3 & 2u);       // A 000 -> 0, C 100 -> 1, G 010 -> 2, T 110 -> 3
        f |= idx << (2 * b); r |= (3u - idx) << (2 * (nb - 1 - b));
    }
    uint32_t w; int a, b;
    bloom4_pos(f, bloom4_minimizer(f, nwin), nwin, lbits, &w, &a, &b);
    atomicOr(&bloom[w], ((1u << l) << a) | ((1u << l) << b));
    bloom4_pos(r, bloom4_minimizer(r, nwin), nwin, lbits, &w, &a, &b);
    atomicOr(&bloom[w], ((4u << l) << a) | ((4u << l) << b));
}
// the same two entries per key as (tile, word, bits) items for harc_bitmap_from_items: a combined bitmap beyond the caches (tens of millions of
// candidates: reads with N on a 150-bp set) is built from sorted items instead of with random atomics; keys with an N become items that set nothing
__global__ void k_bloom4_items(const uint64_t *keys, uint32_t n, int lbits, int nwin, int l, int nb, uint32_t skip_tile, uint64_t *items)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t k = keys[i];
    uint64_t f = 0, r = 0; bool hasn = false;
    for (int b = 0; b < nb; b++) {
        const uint32_t c3 = (uint32_t)(k >> (3 * b)) & 7u;
        hasn |= (c3 & 1u) != 0;
        const uint64_t idx = ((c3 >> 2) & 1u) | (c3 & 2u);
        f |= idx << (2 * b); r |= (3u - idx) << (2 * (nb - 1 - b));
    }
    if (hasn) { items[2 * (size_t)i] = items[2 * (size_t)i + 1] = (uint64_t)skip_tile; return; }
    uint32_t w; int a, b;
    bloom4_pos(f, bloom4_minimizer(f, nwin), nwin, lbits, &w, &a, &b);
    items[2 * (size_t)i] = harc_bitmap_item(w, a + l, b + l);
    bloom4_pos(r, bloom4_minimizer(r, nwin), nwin, lbits, &w, &a, &b);
    items[2 * (size_t)i + 1] = harc_bitmap_item(w, a + 2 + l, b + 2 + l);
}
__global__ void k_words_differ(const uint32_t *a, const uint32_t *b, uint64_t nwords, unsigned long long *ndiff)
{
    const uint64_t i = harc_gid();
    const unsigned long long m = __ballot(i < nwords && a[i] != b[i]);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(ndiff, (unsigned long long)__popcll(m));
}
// candidates in the reads' 2-bit code + N mask (3-bit code A0 N1 G2 C4 T6: code = c3 >> 1, N = c3 & 1); one thread per (read, word)
__global__ void k_cand2_from3(const uint64_t *cand3, uint32_t T, int L, int W, int W3, uint64_t *cand2, uint64_t *candN)
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)T * W) return;
    const uint32_t i = (uint32_t)(gid / W); const int w = (int)(gid % W);
    const uint64_t *r = cand3 + (size_t)i * W3;
    // the 32 bases of word w are the 96 bits from bit 96 w on: 64 + 32 of them out of two (three) words, every shift below a constant
    // (a call of c3_at per base was 20 ms for the 220 M candidates of configs[3]); the store holds zeros behind base L, so nothing is cut
    (void)L;
    const int wi = (3 * w) >> 1;
    const uint64_t x0 = r[wi], x1 = wi + 1 < W3 ? r[wi + 1] : 0;
    uint64_t lo, hi;
    if (w & 1) { lo = (x0 >> 32) | (x1 << 32); hi = x1 >> 32; } else { lo = x0; hi = x1 & 0xFFFFFFFFull; }
    uint64_t a = 0, n = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint64_t c3 = (k <= 20 ? (lo >> (3 * k)) : k == 21 ? ((lo >> 63) | (hi << 1)) : (hi >> (3 * k - 64))) & 7;
        a |= (c3 >> 1) << (2 * k);
        n |= ((c3 & 1) * 3) << (2 * k);
    }
    cand2[gid] = a; candN[gid] = n;
}
template <int W> __device__ __forceinline__ void cons_words(const uint64_t *cons2, uint64_t g, uint64_t (&cw)[W])
{
    const uint64_t bit = 2 * g, wi = bit >> 6; const int sh = (int)(bit & 63);
    uint64_t lo = cons2[wi];
#pragma unroll
    for (int w = 0; w < W; w++) { const uint64_t hi = cons2[wi + w + 1]; cw[w] = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo; lo = hi; }
}

// singleton / N-read realignment, phase 1 (encoder.cpp:252-410).  A workgroup stages 2048 + L consensus bytes in LDS; each thread
// owns 8 consecutive window starts and rolls its four dictionary keys (forward / reverse x 2 dictionaries) from one window to the next.
// A key first meets a one-hash bitmap (cache-resident, most windows end there), then the bucketed table; a candidate is compared on
// packed words: the 3-bit Hamming distance of encoder.cpp:300-312 equals, position by position, popcount of the 2-bit XOR where the
// read has a base, and 1 + popcount(window code) where it has an N (N = 001 against A 000, G 010, C 100, T 110).
#define RSTRIP 8
#define RTILE (256 * RSTRIP)
// one probe that passed the bitmap: window start x, direction, dictionary, key
template <int W> __device__ __forceinline__ void realign_probe(const S2Args &s, uint64_t x, int dir, int l, uint64_t key)
{
    uint32_t st = 0, cnt = 0;
    if (!dict_lookup_b(s.slots[l], s.cap[l], key, &st, &cnt)) return;
    const int L = s.L;
    const unsigned long long tp = (x << 2) | ((uint64_t)dir << 1) | (uint64_t)l;
    if (cnt & SLOT_BIG) {                                     // > maxsearch reads: the visible window slides as reads get claimed
        const unsigned int at = atomicAdd(s.nevents, 1u);     // (encoder.cpp:293) -> exact sequential pass k_realign_big
        if (at < s.maxevents) s.events[at] = make_uint4((uint32_t)tp, (uint32_t)(tp >> 32), st, cnt & SLOT_CNT_MASK);
        return;
    }
    uint64_t wv[W];                                           // window words, forward or reverse complement
    {
        uint64_t cw[W];
        cons_words<W>(s.cons2, x, cw);
#pragma unroll
        for (int w = 0; w < W; w++) cw[w] &= lowmask_word(2 * L, w);
        if (dir) rc_words<W>(cw, L, wv);
        else {
#pragma unroll
            for (int w = 0; w < W; w++) wv[w] = cw[w];
        }
    }
    const bool emb = (cnt & SLOT_EMB) != 0;
    cnt &= SLOT_CNT_MASK;                                     // <= maxsearch: the whole bin is always visible
    for (uint32_t t = 0; t < cnt; t++) {
        const uint32_t rid = emb ? st : s.ids[l][st + cnt - 1 - t];
        const uint64_t *a2 = s.cand2 + (size_t)rid * W, *n2 = s.candN + (size_t)rid * W;
        int hd = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            const uint64_t av = a2[w], nv = n2[w];
            hd += __popcll((av ^ wv[w]) & ~nv) + __popcll(nv & 0x5555555555555555ULL) + __popcll(wv[w] & nv);
        }
        if (hd <= s.thresh_s) atomicMin(&s.best[rid], tp);
    }
}
// Both windows of the same width n (every read length >= 50): one thread per 8 consecutive k-mer starts c, ONE rolled key, one
// bitmap word per k-mer (k_bloom4_set).  Plane l says the k-mer is window l of the forward read starting at x = c - ds[l]; plane 2 + l
// that it is the reverse complement of window l of the read starting at x = c - (L - 1 - de[l]).  The claim order of the reference
// (window start, direction, dictionary) is carried by the tuple and atomicMin, so it does not matter which thread finds a probe.
#define RQCAP 4096
template <int W> __device__ __forceinline__ void realign_hit1(const S2Args &s, const uint8_t *tile, uint64_t X0, int tc, int p, int n)
{
    const int dir = p >> 1, l = p & 1;
    const uint64_t cpos = X0 + tc;
    const uint64_t off = dir ? (uint64_t)(s.L - 1 - s.de[l]) : (uint64_t)s.ds[l];
    if (cpos < off) return;
    const uint64_t x = cpos - off;
    if (x < s.col0 || x >= s.col1 || !(s.cons[x] & 4)) return;    // another rank's column, or no read may start there (k_consensus)
    uint64_t key = 0;
    if (dir) for (int b = 0; b < n; b++) key |= (uint64_t)idx_to_c3(3 - (tile[tc + n - 1 - b] & 3)) << (3 * b);
    else for (int b = 0; b < n; b++) key |= (uint64_t)idx_to_c3(tile[tc + b] & 3) << (3 * b);
    realign_probe<W>(s, x, dir, l, key);
}
template <int W, int NWIN> __global__ __launch_bounds__(256) void k_realign_propose1(S2Args s)
{
    __shared__ uint32_t tile32[(RTILE + 64 + 8) / 4];
    __shared__ uint16_t queue[RQCAP];
    __shared__ uint2 lookup[RTILE];                               // per column: word of the bitmap, the two 4-bit fields inside it
    __shared__ unsigned int qn;
    uint8_t *tile = reinterpret_cast<uint8_t *>(tile32);
    const uint64_t X0 = (uint64_t)(blockIdx.x + s.tile_base) * RTILE;
    const int n = s.de[0] - s.ds[0] + 1;
    const int ntile = RTILE + n + NWIN + 1;
    for (int d = threadIdx.x; d < (ntile + 3) / 4; d += 256) {
        const uint64_t g = X0 + 4ull * d;
        uint32_t v = 0;
        if (g + 4 <= s.total) v = *reinterpret_cast<const uint32_t *>(s.cons + g);
        else for (int k = 0; k < 4; k++) if (g + k < s.total) v |= (uint32_t)s.cons[g + k] << (8 * k);
        tile32[d] = v;
    }
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const int t0 = threadIdx.x * RSTRIP;
    uint64_t k = 0;                                               // the window in the bitmap's 2-bit code (bloom4_mmer)
    for (int b = 0; b < n; b++) k |= (uint64_t)(tile[t0 + b] & 3) << (2 * b);
    const uint32_t *bloom = s.bloom[0]; const int lbits = s.bloom_lbits;
    // the keys of the strip; with minimizers also the m-mers at the NWIN - 1 positions behind it (the m-mer at p is the low end of key p)
    constexpr int NK = RSTRIP + (NWIN > 0 ? NWIN - 1 : 0);
    uint64_t keys[RSTRIP]; uint32_t mm[NK];
#pragma unroll
    for (int c = 0; c < NK; c++) {
        if (c < RSTRIP) keys[c] = k;
        mm[c] = bloom4_mmer(k);
        k = (k >> 2) | ((uint64_t)(tile[t0 + c + n] & 3) << (2 * (n - 1)));
    }
    // where the key of every column of the strip sits in the bitmap ...
#pragma unroll
    for (int c = 0; c < RSTRIP; c++) {
        uint32_t mz = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NWIN; i++) mz = mm[c + i] < mz ? mm[c + i] : mz;
        uint32_t w; int a, b;
        bloom4_pos(keys[c], mz, NWIN, lbits, &w, &a, &b);
        lookup[t0 + c] = make_uint2(w, (uint32_t)a | ((uint32_t)b << 8));
    }
    __syncthreads();
    // ... and the lookups TRANSPOSED: lane i of a wave takes column 64 q + i, so that the ~4 consecutive columns that share a minimizer -- and
    // with it a 64-byte line of the bitmap -- are neighbouring lanes of ONE load instead of four loads of one lane (a quarter of the requests
    // to L2: the kernel made 3.9 G of them per launch at configs[2], 1.07 G missing).  The rare hits go through a queue in LDS and are worked
    // off one per lane (inline, a wave would wait for its unluckiest lane's chain of dependent loads while the other 63 idle)
#pragma unroll
    for (int c = 0; c < RSTRIP; c++) {
        const int col = c * 256 + (int)threadIdx.x;
        if (X0 + col < s.total) {
            const uint2 lk = lookup[col];
            const uint32_t v = bloom[lk.x];
            uint32_t m = (v >> (lk.y & 0xFFu)) & (v >> (lk.y >> 8)) & 15u;
            while (m) {
                const int p = __builtin_ctz(m); m &= m - 1;
                const unsigned int at = atomicAdd(&qn, 1u);
                if (at < RQCAP) queue[at] = (uint16_t)(col | (p << 12));
                else realign_hit1<W>(s, tile, X0, col, p, n);
            }
        }
    }
    __syncthreads();
    const unsigned int nq = qn < RQCAP ? qn : RQCAP;
    for (unsigned int i = threadIdx.x; i < nq; i += 256) { const uint16_t e = queue[i]; realign_hit1<W>(s, tile, X0, e & 0xFFF, e >> 12, n); }
}

template <int W> __global__ __launch_bounds__(256) void k_realign_propose(S2Args s)
{
    __shared__ uint32_t tile32[(RTILE + 256 + 8) / 4];
    uint8_t *tile = reinterpret_cast<uint8_t *>(tile32);
    const uint64_t X0 = (uint64_t)(blockIdx.x + s.tile_base) * RTILE;
    const int L = s.L;
    const int ntile = RTILE + L + 1;
    for (int d = threadIdx.x; d < (ntile + 3) / 4; d += 256) {
        const uint64_t g = X0 + 4ull * d;
        uint32_t v = 0;
        if (g + 4 <= s.total) v = *reinterpret_cast<const uint32_t *>(s.cons + g);
        else for (int k = 0; k < 4; k++) if (g + k < s.total) v |= (uint32_t)s.cons[g + k] << (8 * k);
        tile32[d] = v;
    }
    __syncthreads();
    const int t0 = threadIdx.x * RSTRIP;
    if (X0 + t0 >= s.total) return;
    const int n0 = s.de[0] - s.ds[0] + 1, n1 = s.de[1] - s.ds[1] + 1;
    const uint64_t m0 = (n0 * 3 < 64) ? ((1ULL << (3 * n0)) - 1) : ~0ULL, m1 = (n1 * 3 < 64) ? ((1ULL << (3 * n1)) - 1) : ~0ULL;
    uint64_t kf0 = 0, kf1 = 0, kr0 = 0, kr1 = 0;
    for (int b = 0; b < n0; b++) {
        kf0 |= (uint64_t)idx_to_c3(tile[t0 + s.ds[0] + b] & 3) << (3 * b);
        kr0 |= (uint64_t)idx_to_c3(3 - (tile[t0 + L - 1 - s.ds[0] - b] & 3)) << (3 * b);
    }
    for (int b = 0; b < n1; b++) {
        kf1 |= (uint64_t)idx_to_c3(tile[t0 + s.ds[1] + b] & 3) << (3 * b);
        kr1 |= (uint64_t)idx_to_c3(3 - (tile[t0 + L - 1 - s.ds[1] - b] & 3)) << (3 * b);
    }
    for (int c = 0; c < RSTRIP; c++) {
        const int tx = t0 + c;
        const uint64_t x = X0 + tx;
        if (x >= s.total) break;
        if (x >= s.col0 && x < s.col1 && (tile[tx] & 4)) {
            // the four probes of the window: bitmap first (all four loads in flight together)
            const uint64_t keys[4] = { kf0, kf1, kr0, kr1 };               // order of encoder.cpp: direction, then dictionary
            bool pass[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l = k & 1;
                const uint64_t b = mix64(keys[k] ^ 0x9E3779B97F4A7C15ULL) >> s.bloom_shift[l];
                pass[k] = (s.bloom[l][b >> 5] >> (b & 31)) & 1u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) if (pass[k]) realign_probe<W>(s, x, k >> 1, k & 1, keys[k]);
        }
        // roll the keys to window x+1
        kf0 = (kf0 >> 3) | ((uint64_t)idx_to_c3(tile[tx + 1 + s.de[0]] & 3) << (3 * (n0 - 1)));
        kf1 = (kf1 >> 3) | ((uint64_t)idx_to_c3(tile[tx + 1 + s.de[1]] & 3) << (3 * (n1 - 1)));
        kr0 = ((kr0 << 3) & m0) | (uint64_t)idx_to_c3(3 - (tile[tx + L - s.ds[0]] & 3));
        kr1 = ((kr1 << 3) & m1) | (uint64_t)idx_to_c3(3 - (tile[tx + L - s.ds[1]] & 3));
    }
}

// Bins larger than maxsearch, exact: the reference scans, at every probe, the maxsearch highest ids of the bin that are STILL
// unclaimed at that moment (encoder.cpp:293 after the removals of :321-336), so what a probe sees depends on the probes before it.
// The probes that hit such a bin were recorded by k_realign_propose as events (tuple, bin).  With c(r) = the tuple at which read r is
// claimed (best[r]), the sequential result is the least fixed point of
//     c(r) = min { t(e) : event e accepts r and fewer than maxsearch reads r' > r of its bin have c(r') >= t(e) }
// (together with the claims of the ordinary bins, which never depend on a window).  Iterating from c = "unclaimed" only ever lowers c
// and never below the sequential value (a read that looks visible with too-late claims above it is visible in the sequential run as
// well), and a state that no event changes any more IS the sequential one.  So: one wave per event, all events at once, 64
// candidates per round trip, repeated until a pass changes nothing -- as many passes as the deepest bin has windows of maxsearch.
// estart[e]: the reads of the bin at positions >= estart[e] (the highest ids) were claimed before event e's tuple in an earlier pass;
// claims only move to earlier tuples, so they stay claimed for e and the next pass starts below them (a bin of n identical reads would
// otherwise cost every one of its events n loads per pass).  (One wave per BIN walking its events in tuple order settles a deep bin in
// one pass, but serialises the hundreds of thousands of probes a low-complexity bin attracts: tried, 100x slower.)
// binver[l][start of the bin in ids[l]]: bumped whenever the claim of one of the bin's reads moves (both bins of the read: it sits in one
// bin of either dictionary).  What an event does is a function of the claims of its bin's reads, so an event whose bin has the version
// it had when the event last looked is skipped: after the first passes only the events of the bins that are still settling -- the
// deepest ones -- are scanned again (a 50 M-read repeat-rich set: 18 passes over ALL events before).
// Which events run in a pass is free (any order of event executions keeps c above the sequential value, see above); only the END needs
// passes over all events that change nothing.  Running all events in every pass costs (depth of the deepest bin) x (all events):
// a poly-A bin of 17 000 reads attracts half a million probes, the first 17 of which -- in tuple order -- take 1000 reads each while
// all the others rescan the same window seventeen times to lose every bid (50 M-read repeat-rich set: 17 passes x 23 ms).  So the
// events are ranked by tuple inside their bin (two stable radix sorts) and the passes go over rank ranges [0,64), [64,256), ... x4, each
// repeated until it is quiet: the early events of every bin settle first, at the price of a few dozen events per bin and pass, and the
// later ones then run once against a settled bin.  perm = events in (bin, tuple) order, rank = position inside the bin.
#define EV_DONE 0xFFFFFFFEu     // lastver of an event that never needs to look again (bins within maxsearch: the window never closes)
__global__ void k_ev_key_tuple(const uint4 *ev, uint32_t nev, uint64_t *key, uint32_t *idx)
{
    const uint32_t i = harc_gid32();
    if (i >= nev) return;
    const uint4 e = ev[i];
    key[i] = (uint64_t)e.x | ((uint64_t)e.y << 32); idx[i] = i;
}
__global__ void k_ev_key_bin(const uint4 *ev, const uint32_t *idx, uint32_t nev, uint64_t *key)
{
    const uint32_t i = harc_gid32();
    if (i >= nev) return;
    const uint4 e = ev[idx[i]];
    key[i] = ((uint64_t)(e.x & 1u) << 32) | e.z;                  // (dictionary, first id index of the bin)
}
// the events once more in (bin, tuple) order: a look starts with its event, and through perm[] that was two dependent round trips
__global__ void k_ev_gather(const uint4 *ev, const uint32_t *perm, uint32_t nev, uint4 *out)
{
    const uint32_t i = harc_gid32();
    if (i < nev) out[i] = ev[perm[i]];
}
// one wave per event: the L columns of its window (forward, or reverse-complemented) as W3 words of the candidates' 3-bit code.  One lane per base
// fetches its code from the packed consensus (A0 G1 C2 T3 there, twice that in the 3-bit code; complement = 3 - code) into LDS, then lanes
// 0..W3-1 put one word together each
__global__ __launch_bounds__(256) void k_ev_windows(S2Args s, const uint4 *ev, uint32_t nev, uint64_t *win, uint32_t kbase)
{
    __shared__ uint8_t sc3[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t e = kbase + blockIdx.x * 4 + wv;
    if (e >= nev) return;
    const uint4 v = ev[e];
    const unsigned long long tp = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    const uint64_t x = tp >> 2; const int dir = (int)((tp >> 1) & 1), L = s.L;
    for (int b = lane; b < L; b += 64) {
        const uint64_t g = x + (uint64_t)(dir ? L - 1 - b : b);
        const uint32_t c2 = (uint32_t)(s.cons2[g >> 5] >> (2 * (g & 31))) & 3u;
        sc3[wv][b] = (uint8_t)((dir ? 3u - c2 : c2) << 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < s.W3) {
        unsigned long long w = 0;
        const int b0 = (64 * lane) / 3, b1 = (64 * lane + 63) / 3;
        for (int b = b0; b <= b1 && b < L; b++) {
            const unsigned long long c3 = (unsigned long long)sc3[wv][b];
            const int sh = 3 * b - 64 * lane;
            w |= sh >= 0 ? (c3 << sh) : (c3 >> (-sh));
        }
        win[(size_t)e * s.W3 + lane] = w;
    }
}
__global__ void k_ev_gather_win(const uint64_t *win, const uint32_t *perm, uint32_t nev, int W3, uint64_t *out)
{
    const uint64_t t = harc_gid();
    if (t >= (uint64_t)nev * W3) return;
    const uint32_t i = (uint32_t)(t / W3); const int w = (int)(t % W3);
    out[t] = win[(size_t)perm[i] * W3 + w];
}
__global__ void k_ev_heads(const uint64_t *key, uint32_t nev, uint32_t *head)
{
    const uint32_t i = harc_gid32();
    if (i >= nev) return;
    head[i] = (i > 0 && key[i] != key[i - 1]) ? i : 0u;          // max-scan -> first position of the bin's events
}
__global__ void k_ev_rank(uint32_t *seg, uint32_t nev, unsigned int *maxrank)
{
    const uint32_t i = harc_gid32();
    uint32_t r = 0;
    if (i < nev) { r = i - seg[i]; seg[i] = r; }
    for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(r, o, 64); r = x > r ? x : r; }
    if ((threadIdx.x & 63) == 0 && r) atomicMax(maxrank, r);
}
// Which events have to look again: what an event does is a function of WHICH reads of its bin are claimed before its tuple, so only a
// claim with an EARLIER tuple on a read of its bin can change it.  binmin[l][first id index of the bin] = (pass stamp, smallest tuple
// claimed on a read of the bin during that pass), kept with atomicMin on ((~pass) << 40 | tuple): a newer pass always replaces an older
// entry, inside a pass the smallest tuple stays -- no clearing between passes; two copies by the parity of the pass, so that the claims
// of the running pass do not replace what the previous one left.  An event that looked (or was validated) in the previous
// pass and finds no entry of that pass below its own tuple is validated again without touching the bin.  (A version counter per bin made
// every event of a bin look again whenever ANY claim of the bin moved: once the ranges had settled, 16 more passes of 13 ms over all
// events of the deep bins while a handful of claims moved between the two dictionaries -- c3sd with another stage-I schedule: 220 ms.)
// (round 5) ... and only an event BETWEEN the two tuples of a moved claim: a claim that moves from tuple t_old to t_new < t_old changes "claimed before my
// tuple" for the events with t_new < tuple <= t_old and for nobody else (behind t_old the read was claimed before and still is).  binmax[l][bin] = (pass
// stamp, the largest t_old of the pass; all ones in the tuple field for a read that was unclaimed), kept with atomicMax on (pass << 40 | t_old).  Once
// the claims only shift among neighbouring probes -- the last ten passes over everything at configs[3] with repeats, 20 M probes looking again in each --
// the probes behind the place where it happens stay validated.
#define EV_TBITS 40
// The claims in bin order.  A look used to gather best[ids[l][..]] -- 64 random 8-byte loads per chunk, ~2 G of them per step on a repeat-rich
// 50 M-read set; a deep bin is looked at by hundreds of thousands of events that all want the same entries.  Before every pass one thread per
// (dictionary, entry) copies the claim of that entry's read next to its neighbours in the bin, and the looks read claims as they read ids: one
// line per chunk.  A claim made during a pass goes to best[] (the truth: atomicMin decides) and to the claimed entry of the bin being looked
// at; the read's entry in the OTHER dictionary's bin, and anything read a little late, is staler than best[] -- values that were true at some
// time, which is all the fixed point needs (above: stale claims are later claims, they can only keep an event from claiming, and the bin is
// marked for another look); the pass that ends it changes nothing, so it saw the truth.
__global__ void k_bestbin_refresh(const unsigned long long *best, const uint32_t *ids0, const uint32_t *ids1, uint32_t T, unsigned long long *bb0, unsigned long long *bb1)
{
    const uint32_t i = harc_gid32();
    if (i >= T) return;
    bb0[i] = best[ids0[i]]; bb1[i] = best[ids1[i]];
}
// ... and only where somebody reads them: the bins that have events (one workgroup per bin through the list of the bins' first events; the copy over all
// 2 T entries was 9.3 ms per pass at configs[3] with repeats -- a second per step over its hundred passes -- for 1210 bins that hold a tenth of the entries)
__global__ __launch_bounds__(256) void k_bestbin_refresh_bins(const unsigned long long *best, const uint32_t *ids0, const uint32_t *ids1, const uint4 *events, const uint32_t *firsts, uint32_t nfirsts,
                                                              unsigned long long *bb0, unsigned long long *bb1, S2Args s, int compact)
{
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long sh_tmin;
    const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
    if (b >= nfirsts) return;
    const uint4 ev = events[firsts[b]];
    const int l = (int)(ev.x & 1u);
    const uint32_t *const ids = (l ? ids1 : ids0) + ev.z;
    unsigned long long *const bb = (l ? bb1 : bb0) + ev.z;
    if (!compact) {
        for (uint32_t i = threadIdx.x; i < ev.w; i += 256) bb[i] = best[ids[i]];
        return;
    }
    // compact (s.compact passes): the copy as before, and beside it the entries somebody who looks in this pass can see -- not claimed before the earliest
    // event of the bin that looks (k_ev_validate_tmin) --, in order, with their positions.  A bin nobody looks at keeps what it had (nobody reads it).
    if (threadIdx.x == 0) { sh_tmin = s.tminbin[l][ev.z]; s.tminbin[l][ev.z] = ~0ULL; }
    __syncthreads();
    const unsigned long long tmin = sh_tmin;
    const bool any = tmin != ~0ULL;
    unsigned long long *const cb = s.cbb[l] + ev.z; uint32_t *const cp = s.cpos[l] + ev.z;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < ev.w; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        unsigned long long v = 0; bool keep = false;
        if (i < ev.w) { v = best[ids[i]]; bb[i] = v; keep = any && v >= tmin; }
        if (!any) continue;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = base;
        for (int w = 0; w < wv; w++) off += wcnt[w];
        if (keep) { const uint32_t at = off + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL)); cb[at] = v; cp[at] = i; }
        base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    if (any && threadIdx.x == 0) s.ccnt[l][ev.z] = base;
}
// one event (probe e, at position ei of the pass order), one wave.  validate: an event that looked in the previous pass and finds no earlier
// claim on its bin since is validated without looking.  Returns whether a claim moved (wave-uniform).
__device__ __forceinline__ bool realign_event(const S2Args &s, uint32_t ei, uint32_t e, uint32_t *estart, unsigned int *changed, unsigned long long *binmin0, unsigned long long *binmin1,
                                              uint32_t *lastpass, uint32_t pass, uint32_t T1, unsigned long long *swin, bool validate)
{
    const int lane = threadIdx.x & 63;
    const uint32_t lp = lastpass[ei];
    if (lp == EV_DONE) return false;
    const int W3 = s.W3;
    const uint4 ev = s.events[e];
    const unsigned long long tp = (unsigned long long)ev.x | ((unsigned long long)ev.y << 32);
    const int l = (int)(tp & 1);
    const uint32_t st = ev.z, cnt = ev.w;
    const size_t cur = (size_t)(pass & 1u) * T1, prv = (size_t)((pass - 1u) & 1u) * T1;      // T1 = entries per copy
    unsigned long long *const mymin = (l ? binmin1 : binmin0) + st;
    const unsigned long long tmask = (1ULL << EV_TBITS) - 1ULL, stamp = ((unsigned long long)(~pass & 0xFFFFFFu)) << EV_TBITS;
    if (validate && lp == pass - 1) {                             // looked, or was validated, in the previous pass: anything earlier than this event since?
        const unsigned long long m = __hip_atomic_load(mymin + prv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool moved = (m >> EV_TBITS) == (unsigned long long)(~(pass - 1) & 0xFFFFFFu) && (m & tmask) < tp;
        if (moved && s.binmax_on) { const unsigned long long x = __hip_atomic_load(s.binmax[l] + prv + st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((x >> EV_TBITS) == (unsigned long long)((pass - 1) & 0xFFFFFFu)) moved = tp <= (x & tmask); }
        if (!moved) { if (lane == 0) lastpass[ei] = pass; return false; }
    }
    if (estart[ei] == 0) { if (lane == 0) lastpass[ei] = EV_DONE; return false; }      // every read of the bin was claimed before this event: claims only move to earlier tuples
    if (s.trace && lane == 0) atomicAdd(changed + 2, 1u);         // trace only: events that look (one word: it serialises them)
    // the 3-bit window words of the event (forward or reverse complement of the consensus at its column): made once per event by k_ev_windows
    // (every look used to put them together from the packed consensus again; and the events of a multi-GPU run carry them to the other ranks)
    if (lane < W3) swin[lane] = s.evwin[(size_t)e * W3 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t top = estart[ei]; if (top > cnt) top = cnt;
    uint32_t seen = 0, pos = top; bool ch = false, leading = true;   // wave-uniform
    unsigned long long oldmax = 0;                                 // the latest tuple a claim of this look was taken from (per lane)
    const unsigned long long pstamp = ((unsigned long long)(pass & 0xFFFFFFu)) << EV_TBITS;
    // claims from the bin-ordered copy (k_bestbin_refresh), the next chunk's requested before this one is worked on (more in flight -- two ahead with
    // the gathered claims, eight, or four at once with the copy -- cost the many short looks more than they saved the long ones); ids only for
    // the candidates tested
    const uint32_t *const idl = s.ids[l] + st;
    unsigned long long *const bbl = s.bestbin[l] + st;
    uint32_t pos1 = pos > 64 ? pos - 64 : 0;
    unsigned long long b = (uint32_t)lane < pos ? __hip_atomic_load(&bbl[pos - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
    while (pos > 0 && seen < (uint32_t)s.maxsearch) {             // highest id first
        const uint32_t pos2 = pos1 > 64 ? pos1 - 64 : 0;
        const unsigned long long b1 = (uint32_t)lane < pos1 ? __hip_atomic_load(&bbl[pos1 - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
        const bool valid = (uint32_t)lane < pos;
        const bool un = valid && b >= tp;                          // not claimed before this probe (claimed BY this probe in an earlier pass counts as visible)
        const unsigned long long um = __ballot(un);
        if (leading) {                                             // the claimed reads on top stay claimed for this event
            if (um == 0) top -= (uint32_t)__popcll(__ballot(valid));
            else { top -= (uint32_t)(__ffsll((long long)um) - 1); leading = false; }
        }
        const uint32_t rank = (uint32_t)__popcll(um & ((1ULL << lane) - 1ULL));
        if (un && seen + rank < (uint32_t)s.maxsearch && b > tp) {
            const uint32_t rid = idl[pos - 1 - lane];
            const uint64_t *r = s.cand3 + (size_t)rid * W3;
            int hd = 0;
            for (int w = 0; w < W3; w++) { hd += __popcll(swin[w] ^ r[w]); if (hd > s.thresh_s) break; }
            unsigned long long old = 0;
            if (hd <= s.thresh_s && (old = atomicMin(&s.best[rid], tp)) > tp) {      // every passing candidate of the window is taken (encoder.cpp:296-317)
                atomicMin(&bbl[pos - 1 - lane], tp);
                ch = true;
                old = old > tmask ? tmask : old; oldmax = old > oldmax ? old : oldmax;
                // the read's bin in the other dictionary sees a claim at this tuple too
                const int ol = 1 - l, off = 3 * s.ds[ol], wi = off >> 6, sh = off & 63;
                uint64_t okey = r[wi] >> sh;
                if (sh && wi + 1 < W3) okey |= r[wi + 1] << (64 - sh);
                if (s.kbits[ol] < 64) okey &= ((uint64_t)1 << s.kbits[ol]) - 1;
                uint32_t ost = 0, ocnt = 0;
                if (dict_lookup_b(s.slots[ol], s.cap[ol], okey, &ost, &ocnt) && !(ocnt & SLOT_EMB)) { atomicMin((ol ? binmin1 : binmin0) + cur + ost, stamp | tp); atomicMax(s.binmax[ol] + cur + ost, pstamp | old); }
            }
        }
        seen += (uint32_t)__popcll(um);
        pos = pos1; pos1 = pos2; b = b1;
    }
    ch = __ballot(ch) != 0;
    if (ch) for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = shfl_u64(oldmax, (lane + o) & 63); oldmax = x > oldmax ? x : oldmax; }
    if (lane == 0) { estart[ei] = top; lastpass[ei] = cnt <= (uint32_t)s.maxsearch ? EV_DONE : pass; if (ch) { atomicMin(mymin + cur, stamp | tp); atomicMax(s.binmax[l] + cur + st, pstamp | oldmax); atomicOr(changed, 1u); } }
    return ch;
}
__global__ __launch_bounds__(256) void k_realign_big(S2Args s, uint32_t nev, uint32_t *estart, unsigned int *changed, unsigned long long *binmin0, unsigned long long *binmin1,
                                                     uint32_t *lastpass, uint32_t pass, uint32_t T1, const uint32_t *perm, const uint32_t *rank, uint32_t rlo, uint32_t rhi, const uint32_t *order2, uint32_t kbase)
{
    __shared__ unsigned long long swin[4][HARC_MAXW3 + 32];            // the window words of a wave's event + 256 bytes for the codes they are made of
    const int wv = threadIdx.x >> 6;
    const uint32_t k = kbase + blockIdx.x * 4 + wv;
    if (k >= nev) return;                                         // nev: the events of the ranges this pass covers (order2) or all of them
    const uint32_t ei = order2 ? order2[k] : k;                   // estart / lastpass are indexed by the event's position in (bin, tuple) order
    uint32_t e = ei;
    if (perm) { const uint32_t r = rank[ei]; if (r < rlo || r >= rhi) return; }      // (with ranges s.events is in (bin, tuple) order already: k_ev_gather)
    (void)realign_event(s, ei, e, estart, changed, binmin0, binmin1, lastpass, pass, T1, swin[wv], true);
}
// The chaser.  All events of a pass look at the state the previous pass left, so a chain of events of one bin each of which acts on what
// its predecessor just did -- in a deep bin the event next in tuple order inherits the window -- advances ONE link per pass, while every
// later event of the bin looks again for nothing (c3sd under another stage-I schedule: 22 passes x 340 000 events x 12 ms for 22 claims).
// After each pass one wave per bin that saw a claim follows the chain in tuple order from the first event behind the smallest tuple
// claimed, as long as claims keep moving (and a few events beyond).  Which events run when is free (see above): the passes still end
// with one over everything that changes nothing.
#define EV_CHASE_QUIET 16
#define EV_CHASE_MAX 64
__global__ __launch_bounds__(256) void k_realign_chase(S2Args s, uint32_t nev, uint32_t *estart, unsigned int *changed, unsigned long long *binmin0, unsigned long long *binmin1,
                                                       uint32_t *lastpass, uint32_t pass, uint32_t T1, const uint32_t *perm, const uint32_t *rank, const uint32_t *seglen, uint32_t rhi, uint32_t kbase, const uint32_t *firsts)
{
    __shared__ unsigned long long swin[4][HARC_MAXW3 + 32];            // the window words of a wave's event + 256 bytes for the codes they are made of
    const int wv = threadIdx.x >> 6;
    const uint32_t k0 = kbase + blockIdx.x * 4 + wv;
    if (k0 >= nev) return;                                        // nev: the bins (firsts) or the events
    const uint32_t i0 = firsts ? firsts[k0] : k0;
    if (rank[i0] != 0) return;                       // one wave per bin: the first of its events in (bin, tuple) order
    const uint4 ev0 = s.events[i0];
    const int l = (int)(ev0.x & 1u);
    const unsigned long long m = __hip_atomic_load((l ? binmin1 : binmin0) + (size_t)(pass & 1u) * T1 + ev0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((m >> EV_TBITS) != (unsigned long long)(~pass & 0xFFFFFFu)) return;       // no claim on this bin in the pass that just ran
    const unsigned long long mt = m & ((1ULL << EV_TBITS) - 1ULL);
    uint32_t len = seglen[i0]; if (len > rhi) len = rhi;          // only the ranks the passes have reached
    uint32_t lo = 0, hi = len;                                     // first event of the bin with a tuple above mt
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint4 evm = s.events[i0 + mid];
        const unsigned long long tm = (unsigned long long)evm.x | ((unsigned long long)evm.y << 32);
        if (tm <= mt) lo = mid + 1; else hi = mid;
    }
    int quiet = 0;
    for (uint32_t j = lo, n = 0; j < len && n < EV_CHASE_MAX && quiet < EV_CHASE_QUIET; j++, n++) {
        const bool ch = realign_event(s, i0 + j, i0 + j, estart, changed, binmin0, binmin1, lastpass, pass, T1, swin[wv], false);
        quiet = ch ? 0 : quiet + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// Large passes in two kernels (round 5).  k_realign_big starts a WAVE per event of the ranges reached so far and lets most of them leave at once --
// validated, or done for good: configs[3] with human-like repeats has 402 M probes into large bins, a hundred passes, and a few thousand to a few
// million events that actually look in most of them: 13 s of waves that read two words and left.  From a million events on a pass first asks the
// question with a THREAD per event (the same tests as the top of realign_event, in the same order: what the previous pass left is final, so asking
// before the pass or inside it is the same) and lists the events that have to look; a wave is started for those only.
__global__ void k_ev_validate(S2Args s, uint32_t nact, const uint32_t *order2, const uint32_t *perm, const uint32_t *rank, uint32_t rhi, uint32_t *estart, const unsigned long long *binmin0,
                              const unsigned long long *binmin1, uint32_t *lastpass, uint32_t pass, uint32_t T1, uint32_t *list, unsigned int *nlist)
{
    const uint32_t k = harc_gid32();
    bool look = false; uint32_t ei = 0;
    if (k < nact) {
        ei = order2 ? order2[k] : k;
        const uint32_t lp = lastpass[ei];
        bool go = lp != EV_DONE;
        if (go && perm && rank[ei] >= rhi) go = false;
        if (go) {
            const uint4 ev = s.events[ei];
            const unsigned long long tp = (unsigned long long)ev.x | ((unsigned long long)ev.y << 32);
            const int l = (int)(tp & 1);
            bool moved = true;
            if (lp == pass - 1) {
                const unsigned long long m = ((l ? binmin1 : binmin0) + ev.z)[(size_t)((pass - 1u) & 1u) * T1];
                moved = (m >> EV_TBITS) == (unsigned long long)(~(pass - 1) & 0xFFFFFFu) && (m & ((1ULL << EV_TBITS) - 1ULL)) < tp;
                if (moved && s.binmax_on) { const unsigned long long x = (s.binmax[l] + ev.z)[(size_t)((pass - 1u) & 1u) * T1]; if ((x >> EV_TBITS) == (unsigned long long)((pass - 1) & 0xFFFFFFu)) moved = tp <= (x & ((1ULL << EV_TBITS) - 1ULL)); }
                if (!moved) lastpass[ei] = pass;
            }
            if (moved) { if (estart[ei] == 0) lastpass[ei] = EV_DONE; else look = true; }
        }
    }
    const unsigned long long m = __ballot(look);
    if (m) {
        unsigned int base = 0;
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(nlist, (unsigned int)__popcll(m));
        base = __shfl(base, leader, 64);
        if (look) list[base + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL))] = ei;
    }
}
__global__ __launch_bounds__(256) void k_realign_list(S2Args s, const uint32_t *list, uint32_t nl, uint32_t *estart, unsigned int *changed, unsigned long long *binmin0, unsigned long long *binmin1,
                                                      uint32_t *lastpass, uint32_t pass, uint32_t T1, uint32_t kbase)
{
    __shared__ unsigned long long swin[4][HARC_MAXW3 + 32];
    const int wv = threadIdx.x >> 6;
    const uint32_t k = kbase + blockIdx.x * 4 + wv;
    if (k >= nl) return;
    const uint32_t ei = list[k];
    (void)realign_event(s, ei, ei, estart, changed, binmin0, binmin1, lastpass, pass, T1, swin[wv], true);
}
// In front of a compacted pass: the validation of k_realign_block by a thread per event (the same tests; an event validated here has lastpass = pass and is
// passed over by the kernel), and for every bin the smallest tuple among its events that LOOK (tminbin, reset by k_bestbin_refresh_bins when it has used it):
// the events of a bin follow each other in tuple order, so that is the first looking lane of a run of lanes of one bin.
__global__ __launch_bounds__(256) void k_ev_validate_tmin(S2Args s, uint32_t nact, const uint32_t *order2, const uint32_t *rank, uint32_t rhi, uint32_t *estart, const unsigned long long *binmin0,
                                                          const unsigned long long *binmin1, uint32_t *lastpass, uint32_t pass, uint32_t T1)
{
    const uint64_t k = harc_gid();
    const int lane = threadIdx.x & 63;
    bool look = false; unsigned long long tp = 0, key = ~0ULL; int l = 0; uint32_t st = 0;
    if (k < nact) {
        const uint32_t ei = order2 ? order2[k] : (uint32_t)k;
        const uint32_t lp = lastpass[ei];
        bool go = lp != EV_DONE && lp != pass;
        if (go && rank && rank[ei] >= rhi) go = false;
        if (go) {
            const uint4 ev = s.events[ei];
            tp = (unsigned long long)ev.x | ((unsigned long long)ev.y << 32);
            const unsigned long long tmask = (1ULL << EV_TBITS) - 1ULL;
            l = (int)(tp & 1); st = ev.z; key = ((unsigned long long)st << 1) | (unsigned long long)l;
            bool moved = true;
            if (lp == pass - 1) {
                const size_t prv = (size_t)((pass - 1u) & 1u) * T1;
                const unsigned long long m = ((l ? binmin1 : binmin0) + st)[prv];
                moved = (m >> EV_TBITS) == (unsigned long long)(~(pass - 1) & 0xFFFFFFu) && (m & tmask) < tp;
                if (moved && s.binmax_on) { const unsigned long long x = (s.binmax[l] + st)[prv]; if ((x >> EV_TBITS) == (unsigned long long)((pass - 1) & 0xFFFFFFu)) moved = tp <= (x & tmask); }
                if (!moved) lastpass[ei] = pass;
            }
            if (moved) { if (estart[ei] == 0) lastpass[ei] = EV_DONE; else look = true; }
        }
    }
    const unsigned long long m = __ballot(look);
    if (!m) return;
    const unsigned long long pk = shfl_u64(key, (lane + 63) & 63);
    const unsigned long long heads = __ballot(lane == 0 || key != pk);      // (a lane that does not look has key = all ones: it ends a run, which is what is wanted: fewer atomics, not exactness)
    if (look) {
        const int runstart = 63 - __clzll((long long)(heads & ((2ULL << lane) - 1ULL)));
        const unsigned long long earlier = m & ((1ULL << lane) - 1ULL) & ~((1ULL << runstart) - 1ULL);
        if (!earlier) atomicMin(s.tminbin[l] + st, tp);
    }
}
// The looks TRANSPOSED (round 5, default): one wave per 64 consecutive events of the pass order -- events of ONE bin in tuple order, except where two
// bins meet -- with an event per LANE.  A wave per event fetched, for every candidate of its window, the candidate's words by read id (64 lanes, 64
// random lines, an early-out loop of dependent loads per lane): configs[3] with human-like repeats made 1.4 G looks of up to 1000 candidates, 10.4 s of
// k_realign_list per step at 1.5 TB/s of such fetches.  The events of a bin look at the SAME candidates: here the wave fetches a chunk of 64 entries of
// the bin once (claim, read id, words: lane j takes entry j) into LDS, and every lane then runs its own event over the chunk, entry by entry from the
// highest id down -- the entry's words are the same address for all lanes (a broadcast read), the window words of the lane's event stay in registers, and
// what a lane does with an entry is what realign_event does with it: skipped above the event's start, counted as visible when not claimed before the
// event's tuple, tested while the window (maxsearch visible entries) is open.  The fetches per look fall by the number of lanes that look (the events
// that have to look again are the ones BEHIND a claim that moved: runs of neighbours), the compare becomes the kernel's cost: ~35 vector instructions
// per entry for 64 events.
// The lanes of a bin are in tuple order, and that is used: when several of them accept an entry in the same step, the first one -- the smallest tuple --
// claims it, and for the lanes behind it the entry is claimed before their tuple from that moment (not visible, not counted): within a wave the events
// see each other's claims as the sequential run would, instead of one pass later.  (Any such value is a claim that holds: the argument above k_realign_big.)
template <int NW> __global__ __launch_bounds__(256) void k_realign_block(S2Args s, uint32_t nact, const uint32_t *order2, const uint32_t *rank, uint32_t rhi, uint32_t *estart, unsigned int *changed,
                                                                         unsigned long long *binmin0, unsigned long long *binmin1, uint32_t *lastpass, uint32_t pass, uint32_t T1, int sorted, uint32_t *ebot)
{
    // ebot[event] (may be null: HARC_AMD_S2_EBOT=0): the lowest entry of its bin the event has had in its open window so far.  An event that looks AGAIN
    // tests only below it: an entry above it that is visible now (not claimed before the event's tuple) was visible at that look as well -- claims only
    // move to earlier tuples -- and was tested then, with the result it would have now; everything else above it is claimed before the tuple and stays so.
    // What remains of a second look is the count of the visible entries from the claims alone (the window still has to be found), and the chunks wholly
    // above every lane's mark are not fetched at all.  (The late passes over everything are such looks: 20 M of them per pass at configs[3] with repeats.)
    __shared__ unsigned long long sw[4][64 * NW];                  // the chunk: words of entry j at [j * NW, j * NW + NW)
    __shared__ unsigned long long sb[4][64];                       // its claims
    __shared__ uint32_t srid[4][64];                               // its read ids
    __shared__ uint32_t spos[4][64];                               // its positions in the bin (s.compact: the entries that are left are not neighbours)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t k = harc_gid();                                 // a thread per event of the pass order (G256: more than 2^32 threads go to a second grid row)
    const int W3 = s.W3;
    const unsigned long long tmask = (1ULL << EV_TBITS) - 1ULL, stamp = ((unsigned long long)(~pass & 0xFFFFFFu)) << EV_TBITS;
    const size_t cur = (size_t)(pass & 1u) * T1, prv = (size_t)((pass - 1u) & 1u) * T1;
    // ---- which lanes look (the tests of k_ev_validate / the top of realign_event, in the same order)
    bool look = false; uint32_t ei = 0, st = 0, cnt = 0; int l = 0; unsigned long long tp = 0;
    if (k < nact) {
        ei = order2 ? order2[k] : (uint32_t)k;
        const uint32_t lp = lastpass[ei];
        bool go = lp != EV_DONE && lp != pass;                     // (lp == pass: validated by k_ev_validate_tmin in front of this launch)
        if (go && rank && rank[ei] >= rhi) go = false;
        if (go) {
            const uint4 ev = s.events[ei];
            tp = (unsigned long long)ev.x | ((unsigned long long)ev.y << 32);
            l = (int)(tp & 1); st = ev.z; cnt = ev.w;
            bool moved = true;
            if (lp == pass - 1) {
                const unsigned long long m = __hip_atomic_load((l ? binmin1 : binmin0) + prv + st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                moved = (m >> EV_TBITS) == (unsigned long long)(~(pass - 1) & 0xFFFFFFu) && (m & tmask) < tp;
                if (moved && s.binmax_on) { const unsigned long long x = __hip_atomic_load(s.binmax[l] + prv + st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((x >> EV_TBITS) == (unsigned long long)((pass - 1) & 0xFFFFFFu)) moved = tp <= (x & tmask); }
                if (!moved) lastpass[ei] = pass;
            }
            if (moved) { if (estart[ei] == 0) lastpass[ei] = EV_DONE; else look = true; }
        }
    }
    unsigned long long todo = __ballot(look);
    if (!todo) return;
    if (s.trace && lane == 0) { atomicAdd(changed + 2, (unsigned int)__popcll(todo)); atomicAdd(changed + 3, 1u); }      // trace only: events that look, waves that hold one
    unsigned long long win[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) win[w] = s.evwin[(size_t)(look ? ei : 0u) * W3 + (w < W3 ? w : W3 - 1)];
#pragma unroll
    for (int w = 0; w < NW; w++) if (!look || w >= W3) win[w] = 0ULL;
    bool anych = false;
    while (todo) {                                                 // bin by bin (one, nearly always)
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t g_st = (uint32_t)__shfl((int)st, first, 64); const int g_l = __shfl(l, first, 64);
        const uint32_t g_cnt = (uint32_t)__shfl((int)cnt, first, 64);
        const bool act = look && st == g_st && l == g_l;
        const unsigned long long grp = __ballot(act);
        todo &= ~grp;
        const uint32_t *const idl = s.ids[g_l] + g_st;
        unsigned long long *const bbl = s.bestbin[g_l] + g_st;
        uint32_t top0 = 0, ebot0 = 0xFFFFFFFFu, plow = 0xFFFFFFFFu;
        if (act) { top0 = estart[ei]; if (top0 > g_cnt) top0 = g_cnt; if (ebot) ebot0 = ebot[ei]; }
        uint32_t top = top0, seen = 0; bool leading = true, done = !act, ch = false;
        unsigned long long oldmax = 0;                             // the latest tuple a claim of this lane's event was taken from
        const unsigned long long pstamp = ((unsigned long long)(pass & 0xFFFFFFu)) << EV_TBITS;
        uint32_t pos0 = top0;                                      // the chunk covers the entries [pos0 - nv, pos0), lane j fetches entry pos0 - 1 - j
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)pos0, o, 64); pos0 = x > pos0 ? x : pos0; }
        // s.compact: the walk is over the entries that are LEFT of the bin -- entry number x is cb[x], at position cp[x] of the bin; it starts at the first
        // of them at or above the highest start of the lanes
        const bool cmp = s.compact != 0;
        unsigned long long *const cb = cmp ? s.cbb[g_l] + g_st : bbl;
        const uint32_t *const cp = cmp ? s.cpos[g_l] + g_st : nullptr;
        if (cmp) {
            uint32_t lo = 0, hi = s.ccnt[g_l][g_st];
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cp[mid] < pos0) lo = mid + 1; else hi = mid; }
            pos0 = lo;
        }
        while (pos0 > 0 && __ballot(!done)) {
            const uint32_t nv = pos0 < 64u ? pos0 : 64u;
            __builtin_amdgcn_wave_barrier();
            if ((uint32_t)lane < nv) {
                const uint32_t x = pos0 - 1u - (uint32_t)lane;
                sb[wv][lane] = __hip_atomic_load(&cb[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spos[wv][lane] = cmp ? cp[x] : x;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool needw = __ballot(!done && ebot0 > spos[wv][nv - 1u]) != 0;      // somebody may test an entry of this chunk
            if (needw && (uint32_t)lane < nv) {
                const uint32_t p = spos[wv][lane];
                const uint32_t rid = idl[p];
                srid[wv][lane] = rid;
                const uint64_t *r = s.cand3 + (size_t)rid * W3;
                unsigned long long cw[NW];                         // all words asked for at once (a word behind the read's last: its last word again, then dropped)
#pragma unroll
                for (int w = 0; w < NW; w++) cw[w] = r[w < W3 ? w : W3 - 1];
#pragma unroll
                for (int w = 0; w < NW; w++) sw[wv][lane * NW + w] = w < W3 ? cw[w] : 0ULL;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = 0; j < nv; j++) {
                const uint32_t p = spos[wv][j];
                const unsigned long long bj = sb[wv][j];
                const bool in = !done && p < top0;
                bool un = in && bj >= tp;                          // not claimed before this event's tuple (claimed BY it in an earlier pass counts as visible)
                if (leading && in) { if (un) leading = false; else top = p; }      // the claimed reads on top stay claimed for this event
                if (in) plow = p;
                const bool test = un && bj > tp && p < ebot0;
                if (__ballot(test)) {
                    int hd = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) hd += __popcll(win[w] ^ sw[wv][j * NW + w]);
                    const unsigned long long pm = __ballot(test && hd <= s.thresh_s);      // every passing candidate of the window is taken (encoder.cpp:296-317)
                    if (pm) {
                        // (sorted: the pass order is (bin, tuple) order.  HARC_AMD_S2_FLATPASSES runs over the events as they were recorded: there every
                        // lane that accepts the entry bids for it, as separate waves would)
                        const int f = sorted ? __ffsll((long long)pm) - 1 : -1;   // the smallest tuple among them
                        if (sorted ? lane == f : (test && hd <= s.thresh_s)) {
                            const uint32_t rid = srid[wv][j];
                            unsigned long long old = atomicMin(&s.best[rid], tp);
                            if (old > tp) {
                                atomicMin(&cb[pos0 - 1u - j], tp);
                                ch = true;
                                old = old > tmask ? tmask : old; oldmax = old > oldmax ? old : oldmax;
                                // the read's bin in the other dictionary sees a claim at this tuple too
                                const unsigned long long *r = &sw[wv][j * NW];
                                const int ol = 1 - g_l, off = 3 * s.ds[ol], wi = off >> 6, sh = off & 63;
                                uint64_t okey = r[wi] >> sh;
                                if (sh && wi + 1 < W3) okey |= r[wi + 1] << (64 - sh);
                                if (s.kbits[ol] < 64) okey &= ((uint64_t)1 << s.kbits[ol]) - 1;
                                uint32_t ost = 0, ocnt = 0;
                                if (dict_lookup_b(s.slots[ol], s.cap[ol], okey, &ost, &ocnt) && !(ocnt & SLOT_EMB)) { atomicMin((ol ? binmin1 : binmin0) + cur + ost, stamp | tp); atomicMax(s.binmax[ol] + cur + ost, pstamp | old); }
                            }
                        } else if (sorted && lane > f) un = false;  // claimed at a tuple below this lane's: not visible to it any more
                    }
                }
                if (un) { seen++; if (seen >= (uint32_t)s.maxsearch) done = true; }
            }
            pos0 -= nv;
        }
        // (compacted, and the walk came to the end with this lane's window open: what was left out below is claimed before its tuple like what it saw)
        if (cmp && act && !done && pos0 == 0) { plow = 0; if (leading) top = 0; }
        if (act) {
            estart[ei] = top; lastpass[ei] = g_cnt <= (uint32_t)s.maxsearch ? EV_DONE : pass;
            if (ebot && plow < ebot0) ebot[ei] = plow;
            if (ch) { atomicMin((g_l ? binmin1 : binmin0) + cur + g_st, stamp | tp); atomicMax(s.binmax[g_l] + cur + g_st, pstamp | oldmax); }
        }
        { const unsigned long long cm = __ballot(ch); anych |= cm != 0; if (s.trace && cm && lane == 0) atomicAdd(changed + 4, (unsigned int)__popcll(cm)); }      // trace only: events that moved a claim
    }
    if (anych && lane == 0) atomicOr(changed, 1u);
}
// the first event of every bin in (bin, tuple) order: the chaser starts a wave per BIN, not one per event that finds out it is not a bin's first
__global__ void k_ev_firsts(const uint32_t *rank, uint32_t nev, uint32_t *list, unsigned int *nlist)
{
    const uint32_t i = harc_gid32();
    const bool first = i < nev && rank[i] == 0;
    const unsigned long long m = __ballot(first);
    if (m) {
        unsigned int base = 0;
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(nlist, (unsigned int)__popcll(m));
        base = __shfl(base, leader, 64);
        if (first) list[base + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL))] = i;
    }
}
// the rank range an event belongs to ([0, r0), [r0, 4 r0), [4 r0, 16 r0) ...) as a sort key, and how many events every range holds: the passes
// over the first ranges then launch a wave per event of THOSE ranges instead of one per event of the whole list (two million workgroups
// that leave at once cost 0.5 ms per pass, and the first ranges take dozens of passes)
__global__ void k_ev_range_keys(const uint32_t *rank, uint32_t nev, uint32_t r0, uint64_t *key, uint32_t *pos, unsigned int *hist)
{
    const uint32_t i = harc_gid32();
    const bool in = i < nev;
    uint32_t k = 0xFFFFFFFFu;
    if (in) {
        const uint32_t r = rank[i];
        k = 0;
        for (unsigned long long b = r0; r >= b && k < 31; b *= 4) k++;
        key[i] = k; pos[i] = i;
    }
    unsigned long long todo = __ballot(in);
    while (todo) {                                                 // one atomic per distinct range per wave
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lk = __shfl(k, leader, 64);
        const unsigned long long same = __ballot(in && k == lk);
        if ((threadIdx.x & 63) == leader) atomicAdd(&hist[lk], (unsigned int)__popcll(same));
        todo &= ~same;
    }
}
__global__ void k_ev_seglen(const uint32_t *rank, uint32_t nev, uint32_t *seglen)
{
    const uint32_t i = harc_gid32();
    if (i >= nev) return;
    if (i + 1 == nev || rank[i + 1] == 0) seglen[i - rank[i]] = rank[i] + 1;      // written at the bin's first position
}

__global__ void k_acc_flags(const unsigned long long *best, uint32_t T, uint32_t *flag)
{
    const uint32_t i = harc_gid32();
    if (i >= T) return;
    flag[i] = best[i] != TUPLE_NONE ? 1u : 0u;
}
// accepted candidates, in DESCENDING rid order (so that the stable sort by tuple leaves equal tuples rid-descending,
// the order in which one bin scan inserts them, encoder.cpp:293-317)
__global__ void k_acc_compact(const unsigned long long *best, const uint32_t *rank, uint32_t T, uint32_t A, uint64_t *tup, uint32_t *rid)
{
    const uint32_t i = harc_gid32();
    if (i >= T) return;
    if (best[i] != TUPLE_NONE) { const uint32_t at = A - 1 - rank[i]; tup[at] = best[i]; rid[at] = i; }
}

// merge by rank: final index of read i = i + #accepted with column < gstart[i]; of accepted k = k + #reads with gstart <= column
struct FinalArrays { uint32_t *ref; uint8_t *kind; uint64_t *g; };   // kind 0 original, 1 candidate forward, 2 candidate reverse
// (the reads [i0, i0 + n) and the accepted candidates [a0, a0 + na) of this rank's shards; the final arrays start at final index fbase)
// (gstart does not decrease with i, so neither does the count: k_merge_cuts looks it up for the first read of every block of 256 -- and for the last read of
// the range --, and a read searches between its block's cut and the next one: ~70 accepted candidates at configs[3], a handful of loads from lines
// its neighbours ask for too, instead of 28 dependent ones through 180 M tuples: encode 327 -> 317 ms there, 88 -> 84 ms at configs[2])
__global__ void k_merge_cuts(const uint64_t *gstart, uint32_t i0, uint32_t n, const uint64_t *tup, uint32_t A, long long *cut)
{
    const uint32_t b = harc_gid32(), nb = (n + 255u) / 256u;
    if (b > nb) return;
    const uint32_t t = b < nb ? b * 256u : n - 1u;
    const uint64_t g = gstart[i0 + t];
    long long lo = -1, hi = A;                                    // #accepted with (tuple>>2) < g, minus one
    while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if ((tup[mid] >> 2) < g) lo = mid; else hi = mid; }
    cut[b] = lo;
}
// (round 5: the candidates between the two cuts go to LDS first, one coalesced read for the workgroup, and the searches run there -- through global
// memory every read made ~6 dependent loads of lines its neighbours had just asked for: 1.7 ms per shard of 44 M reads at configs[2], eight shards a step)
#define MERGE_TILE 2048
__global__ __launch_bounds__(256) void k_merge_orig(const uint64_t *gstart, uint32_t i0, uint32_t n, const uint64_t *tup, uint32_t A, FinalArrays f, uint32_t fbase, const long long *cut)
{
    __shared__ uint64_t cols[MERGE_TILE];
    const uint32_t t = harc_gid32();
    const long long c0 = cut[harc_bid()], c1 = cut[harc_bid() + 1] + 1;      // tup[c0] < g (or c0 = -1); tup[c1] >= the next block's first g >= g (or c1 = A)
    const long long inside = c1 - c0 - 1;                                  // the candidates a search of this block can land on
    const bool tiled = inside <= MERGE_TILE;
    if (tiled) {
        for (long long j = threadIdx.x; j < inside; j += 256) cols[j] = tup[c0 + 1 + j] >> 2;
        __syncthreads();
    }
    if (t >= n) return;
    const uint32_t i = i0 + t;
    const uint64_t g = gstart[i];
    long long lo = c0, hi = c1;
    if (tiled) while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (cols[mid - c0 - 1] < g) lo = mid; else hi = mid; }
    else while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if ((tup[mid] >> 2) < g) lo = mid; else hi = mid; }
    const uint32_t at = i + (uint32_t)(lo + 1) - fbase;
    f.ref[at] = i; f.kind[at] = 0; f.g[at] = g;
}
__global__ void k_merge_acc(const uint64_t *gstart, uint32_t M, const uint64_t *tup, const uint32_t *rid, uint32_t a0, uint32_t na, FinalArrays f, uint32_t fbase)
{
    const uint32_t t = harc_gid32();
    if (t >= na) return;
    const uint32_t k = a0 + t;
    const uint64_t x = tup[k] >> 2;
    const long long n = ub_le(gstart, (long long)M, x) + 1;       // originals with cumulative pos <= j come first (encoder.cpp:254-268)
    const uint32_t at = k + (uint32_t)n - fbase;
    f.ref[at] = rid[k]; f.kind[at] = (tup[k] & 2) ? 2 : 1; f.g[at] = x;
}
// final index of the first read of every encoder shard (F where a shard is empty), and the first accepted candidate at or behind its first column
__global__ void k_shard_bounds(const uint64_t *gstart, uint32_t M, uint32_t q, uint32_t E, const uint64_t *tup, uint32_t A, uint64_t total, uint32_t *sh_f, uint32_t *sh_a, uint64_t *sh_col)
{
    const uint32_t e = harc_gid32();
    if (e > E) return;
    const uint64_t st = (uint64_t)e * q;
    if (e == E || st >= M) { sh_f[e] = M + A; sh_a[e] = A; sh_col[e] = total; return; }
    const uint64_t g = gstart[st];
    long long lo = -1, hi = A;
    while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if ((tup[mid] >> 2) < g) lo = mid; else hi = mid; }
    sh_f[e] = (uint32_t)st + (uint32_t)(lo + 1); sh_a[e] = (uint32_t)(lo + 1); sh_col[e] = g;
}

// The noise pass works on packed words: `cons2` is the consensus on the global column axis in the reads' own 2-bit code (A0 G1 C2 T3,
// 32 columns per word), so a read is compared with W funnel-shifted words and the mismatching columns are the set bits of
// ((x | x>>1) & 0x55..) with x = read ^ consensus; an N of a candidate read is a mismatch wherever it stands.
__global__ void k_pack_cons2(const uint8_t *cons, uint64_t total, uint64_t w0, uint64_t nwords, uint64_t *cons2)
{
    const uint64_t w = w0 + harc_gid();      // the words [w0, w0 + nwords)
    if (w >= w0 + nwords) return;
    uint64_t v = 0;
    const uint64_t c0 = w * 32;
    if (c0 + 32 <= total) {
        const uint4 q0 = *(const uint4 *)(cons + c0), q1 = *(const uint4 *)(cons + c0 + 16);
        const uint32_t qq[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int k = 0; k < 32; k++) { const uint32_t idx = (qq[k >> 2] >> (8 * (k & 3))) & 3; v |= (uint64_t)(((idx & 1) << 1) | (idx >> 1)) << (2 * k); }
    } else {
        for (int k = 0; k < 32 && c0 + k < total; k++) { const uint32_t idx = cons[c0 + k] & 3; v |= (uint64_t)(((idx & 1) << 1) | (idx >> 1)) << (2 * k); }
    }
    cons2[w] = v;
}
// final element -> W words of 2-bit code as written to the contig (candidates reverse-complemented when kind 2) + N mask (both bits of
// the field set); 3-bit code A0 N1 G2 C4 T6 -> 2-bit code = c3 >> 1, N = c3 & 1
template <int W> __device__ __forceinline__ void final_words(const S2Args &s, uint32_t ref, int kind, uint64_t (&rd)[W], uint64_t (&nmk)[W])
{
    if (kind == 0) {
        const uint64_t *r = s.oreads + (size_t)ref * W;
#pragma unroll
        for (int w = 0; w < W; w++) { rd[w] = r[w]; nmk[w] = 0; }
        return;
    }
    constexpr int W3M = (3 * 32 * W + 63) / 64;
    uint64_t x3[W3M + 1];
    const uint64_t *r = s.cand3 + (size_t)ref * s.W3;
#pragma unroll
    for (int w = 0; w <= W3M; w++) x3[w] = w < s.W3 ? r[w] : 0;
    uint64_t a[W], n[W];
#pragma unroll
    for (int w = 0; w < W; w++) { a[w] = 0; n[w] = 0; }
#pragma unroll
    for (int j = 0; j < 32 * W; j++) {
        const int off = 3 * j, wi = off >> 6, sh = off & 63;
        uint64_t v = x3[wi] >> sh;
        if (sh > 61) v |= x3[wi + 1] << (64 - sh);
        const uint64_t c3 = v & 7;
        a[j >> 5] |= (c3 >> 1) << (2 * (j & 31));
        n[j >> 5] |= ((c3 & 1) * 3) << (2 * (j & 31));
    }
#pragma unroll
    for (int w = 0; w < W; w++) { const uint64_t lm = lowmask_word(2 * s.L, w); a[w] &= lm; n[w] &= lm; }
    if (kind == 1) {
#pragma unroll
        for (int w = 0; w < W; w++) { rd[w] = a[w]; nmk[w] = n[w]; }
    } else {
        uint64_t t[W];
        rc_words<W>(a, s.L, rd);
        rc_words<W>(n, s.L, t);                                   // reversed and complemented; undo the complement
#pragma unroll
        for (int w = 0; w < W; w++) nmk[w] = ~t[w] & lowmask_word(2 * s.L, w);
    }
}
// (the final-list entries [fb, F): the emitting pass is launched shard by shard, so that a shard's streams leave for the host while the next is written)
template <int W, bool EMIT> __global__ void k_noise(S2Args s, FinalArrays f, const uint64_t *cons2, uint32_t fb, uint32_t F, uint32_t *nm, uint32_t *nonN,
                                                    const uint64_t *nmoff, const uint32_t *nonNrank,
                                                    uint8_t *noise, uint8_t *noisepos, uint8_t *posb, uint8_t *rcb, uint32_t *order_out, uint32_t *orderN_out)
{
    const uint64_t i64 = (uint64_t)fb + harc_gid();
    if (i64 >= F) return;
    const uint32_t i = (uint32_t)i64;
    const uint32_t ref = f.ref[i]; const int kind = f.kind[i]; const uint64_t g = f.g[i];
    uint64_t rd[W], nk[W], cw[W], mm[W];
    final_words<W>(s, ref, kind, rd, nk);
    cons_words<W>(cons2, g, cw);
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t x = rd[w] ^ cw[w];
        mm[w] = ((x | (x >> 1) | nk[w]) & 0x5555555555555555ULL) & lowmask_word(2 * s.L, w);
        cnt += (uint32_t)__popcll(mm[w]);
    }
    if (!EMIT) {                                                  // writecontig's sizes first (encoder.cpp:654-717)
        nm[i] = cnt;
        nonN[i] = (kind != 0 && ref >= s.S) ? 0u : 1u;            // N reads are exactly the candidates with index >= S
        return;
    }
    uint64_t np = nmoff[i], nz = nmoff[i] + i;                    // one '\n' per earlier read
    int prevj = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        uint64_t m = mm[w];
        while (m) {
            const int b = __ffsll((long long)m) - 1; m &= m - 1;
            const int j = 32 * w + (b >> 1);
            const int rb = ((nk[w] >> b) & 1) ? 4 : pc_to_idx((int)((rd[w] >> b) & 3)), cb = pc_to_idx((int)((cw[w] >> b) & 3));
            noise[nz++] = (uint8_t)enc_noise(cb, rb); noisepos[np++] = (uint8_t)(j - prevj); prevj = j;
        }
    }
    noise[nz] = '\n';
    const bool is_head = (kind == 0) && s.head[ref];
    posb[i] = is_head ? (uint8_t)s.L : (uint8_t)(g - f.g[i - 1]);
    rcb[i] = kind == 0 ? s.rc[ref] : (kind == 1 ? 'd' : 'r');
    const uint32_t ov = kind == 0 ? s.order[ref] : s.cand_order[ref];
    const bool isN = (kind != 0 && ref >= s.S);
    if (isN) orderN_out[i - nonNrank[i]] = ov; else order_out[nonNrank[i]] = ov;
}
template <bool EMIT> static void launch_noise(harc_amd_ctx *c, const S2Args &a, const FinalArrays &f, const uint64_t *cons2, uint32_t fb, uint32_t F, uint32_t *nm, uint32_t *nonN,
                                              const uint64_t *nmoff, const uint32_t *nonNrank, uint8_t *noise, uint8_t *noisepos, uint8_t *posb, uint8_t *rcb,
                                              uint32_t *order_out, uint32_t *orderN_out)
{
#define NOISE_CASE(WW) case WW: hipLaunchKernelGGL((k_noise<WW, EMIT>), harc_grid256(F - fb), dim3(256), 0, c->stream, a, f, cons2, fb, F, nm, nonN, nmoff, nonNrank, noise, noisepos, posb, rcb, order_out, orderN_out); break;
    switch (a.W) { NOISE_CASE(1) NOISE_CASE(2) NOISE_CASE(3) NOISE_CASE(4) NOISE_CASE(5) NOISE_CASE(6) NOISE_CASE(7) NOISE_CASE(8) }
#undef NOISE_CASE
}

// unaligned candidates (encoder.cpp:484-499): singletons -> order + bases; N reads -> order_N + text
// (candidates [t0, t0 + n): this rank's share of the leftovers; fs / fn / rs / rn are indexed from t0)
__global__ void k_left_flags(const unsigned long long *best, uint32_t t0, uint32_t n, uint32_t S, uint32_t *fs, uint32_t *fn)
{
    const uint32_t t = harc_gid32();
    if (t >= n) return;
    const uint32_t i = t0 + t;
    const bool un = best[i] == TUPLE_NONE;
    fs[t] = (un && i < S) ? 1u : 0u; fn[t] = (un && i >= S) ? 1u : 0u;
}
// One thread per (read, group of 16 bases): neighbouring threads write neighbouring 16 bytes.  (A thread per read wrote its L bytes one
// by one, L bytes apart from its neighbour's: 50 ms for the 115 M leftover reads of an 8-way bucket shard of configs[2].)
// their order entries (behind the aligned ones: the bases are known long before -- k_noise's sizes pass decides where the aligned entries end)
__global__ void k_left_orders(S2Args s, uint32_t t0, uint32_t nt, const uint32_t *rs, const uint32_t *rn, uint32_t *order_out, uint32_t order_base, uint32_t *orderN_out, uint32_t orderN_base)
{
    const uint32_t t = harc_gid32();
    if (t >= nt) return;
    const uint32_t i = t0 + t;
    if (s.best[i] != TUPLE_NONE) return;
    if (i < s.S) order_out[order_base + rs[t]] = s.cand_order[i]; else orderN_out[orderN_base + rn[t]] = s.cand_order[i];
}
__global__ void k_left_emit(S2Args s, uint32_t t0, uint32_t nt, const uint32_t *rs, const uint32_t *rn, uint8_t *sing_bases, char *ntext)
{
    const int G = (s.L + 15) / 16;
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)nt * G) return;
    const uint32_t i = t0 + (uint32_t)(gid / G); const int g = (int)(gid % G);
    if (s.best[i] != TUPLE_NONE) return;
    const uint64_t *r = s.cand3 + (size_t)i * s.W3;
    const int j0 = 16 * g, j1 = j0 + 16 < s.L ? j0 + 16 : s.L;
    if (i < s.S) {
        const uint32_t k = rs[i - t0];
        uint8_t *o = sing_bases + (size_t)k * s.L;
        for (int j = j0; j < j1; j++) o[j] = (uint8_t)c3_to_idx5(c3_at(r, s.W3, j));
    } else {
        const uint32_t k = rn[i - t0];
        char *o = ntext + (size_t)k * (s.L + 1);
        for (int j = j0; j < j1; j++) o[j] = "ACGTN"[c3_to_idx5(c3_at(r, s.W3, j))];
        if (j1 == s.L) o[s.L] = '\n';
    }
}

// The same from the LIST of the leftover candidates, singletons first (k_left_list), one wave per read and one lane per base: with a thread group per
// candidate, 1.4 G threads found out that nine in ten of the 200 M candidates of configs[3] are aligned -- 30 ms for 1.4 GB of text.
// (A first version kept k_left_emit's groups of 16 bases over the list: at 50 M reads its text differed between two launches on the same inputs, in
// whole waves and always in the sixth base of every group -- the one iteration in which two lanes per read take the branch for a field that straddles
// two words -- while the same loop over all candidates, with a tenth of the lanes active, gave the same bytes every time; no overlapping buffers, no
// concurrent writer, cause not found.  This form has no loop and no branch on the field's position.)
__global__ void k_left_list(const uint32_t *fs, const uint32_t *fn, const uint32_t *rs, const uint32_t *rn, uint32_t nt, uint32_t US, uint32_t *list)
{
    const uint32_t t = harc_gid32();
    if (t >= nt) return;
    if (fs[t]) list[rs[t]] = t;
    else if (fn[t]) list[US + rn[t]] = t;
}
__global__ __launch_bounds__(256) void k_left_emit_w(S2Args s, uint32_t t0, const uint32_t *list, uint32_t nl, uint32_t US, uint8_t *sing_bases, char *ntext, uint32_t kbase)
{
    const uint32_t q = kbase + blockIdx.x * 4u + (threadIdx.x >> 6);      // entry of the list: singletons [0, US), reads with N behind them
    const int lane = threadIdx.x & 63;
    if (q >= nl) return;
    const uint32_t i = t0 + list[q];
    const uint64_t *r = s.cand3 + (size_t)i * s.W3;
    const bool single = q < US;
    uint8_t *const ob = sing_bases + (size_t)q * s.L;
    char *const ot = ntext + (size_t)(q - US) * (s.L + 1);
    for (int j = lane; j < s.L; j += 64) {
        const int off = 3 * j, wi = off >> 6, sh = off & 63;
        const uint64_t lo = r[wi], hi = wi + 1 < s.W3 ? r[wi + 1] : 0ULL;
        const int c3 = (int)(((lo >> sh) | (sh ? (hi << (64 - sh)) : 0ULL)) & 7ULL);
        const int b = c3_to_idx5(c3);
        if (single) ob[j] = (uint8_t)b; else ot[j] = "ACGTN"[b];
    }
    if (!single && lane == 0) ot[s.L] = '\n';
}

// packbits (encoder.cpp:527-548, :560-578)
__global__ void k_pack2_bytes(const uint8_t *bases, uint64_t nbytes_out, uint8_t *out)
{
    const uint64_t i = harc_gid();
    if (i >= nbytes_out) return;
    const uint8_t *b = bases + 4 * i;
    out[i] = (uint8_t)((b[0] & 3) | ((b[1] & 3) << 2) | ((b[2] & 3) << 4) | ((b[3] & 3) << 6));
}
__global__ void k_pack1_bytes(const uint8_t *rc, uint64_t nbytes_out, uint8_t *out)
{
    const uint64_t i = harc_gid();
    if (i >= nbytes_out) return;
    const uint8_t *b = rc + 8 * i;
    uint8_t v = 0;
    for (int k = 0; k < 8; k++) v |= (uint8_t)((b[k] == 'r') << k);
    out[i] = v;
}
__global__ void k_bases_to_ascii(const uint8_t *bases, uint64_t n, uint8_t *out)
{
    const uint64_t i = harc_gid();
    if (i >= n) return;
    out[i] = (uint8_t)"ACGT"[bases[i] & 3];
}

// pack_order.cpp:36-65: every 32 values -> numbits u32 words, LSB-first bit stream. One thread per output word.
__global__ void k_pack_order(const uint32_t *order, uint32_t ngroups, int numbits, uint32_t *out)
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)ngroups * numbits) return;
    const uint32_t g = (uint32_t)(gid / numbits); const int w = (int)(gid % numbits);
    const int k0 = (32 * w) / numbits, k1 = (32 * w + 31) / numbits;
    uint32_t v = 0;
    for (int k = k0; k <= k1 && k < 32; k++) {
        const uint32_t o = order[(size_t)g * 32 + k];
        const int sh = k * numbits - 32 * w;
        v |= sh >= 0 ? (o << sh) : (o >> (-sh));
    }
    out[gid] = v;
}

// Digest of a byte range of a stream as it sits in HBM (harc_amd_stream_digest): the sum over its 8-byte words of a 64-bit mix of
// (word, index of the word, salt) -- position-dependent, so any byte that moves or changes shows, and a sum, so the order of the additions
// does not matter.  base is 8-byte aligned; the bytes behind `nbytes` in the last word are masked out (the buffers come from the pool
// uncleared).  Full-size runs are compared through four of these words instead of through gigabytes of host memory.
__global__ void k_stream_digest(const uint8_t *base, uint64_t nbytes, uint64_t salt, unsigned long long *out)
{
    const uint64_t nw = (nbytes + 7) / 8;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = reinterpret_cast<const uint64_t *>(base)[i];
        const uint64_t left = nbytes - 8 * i;
        if (left < 8) v &= ((uint64_t)1 << (8 * left)) - 1;
        acc += mix64(v ^ mix64(i * 0x9E3779B97F4A7C15ULL + salt));
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// ---------------------------------------------------------------------------------------------- host side
// One WAVE per item, four items per workgroup of 256 threads, in launches of at most 2^23 workgroups: a grid is dispatched with its size in WORK-ITEMS as a
// 32-bit number (devutil.h), and 2^26 items x 64 lanes is 2^32.  (Round 5: configs[3] with human-like repeats makes 402 M probes into bins above maxsearch; the launch
// over all of them was cut short without an error, the window passes ended early and differently from run to run -- lossless, but neither deterministic
// nor the sequential result.  From 67 M probes on; nothing smaller had ever been run.)  `launch` sees nb_ (workgroups) and kb_ (first item).
#define WAVE_PER_ITEM(n, launch) do { const uint64_t nwg_ = ((uint64_t)(n) + 3) / 4; for (uint64_t b_ = 0; b_ < nwg_; b_ += (1u << 23)) { \
        const unsigned nb_ = (unsigned)(nwg_ - b_ < (1u << 23) ? nwg_ - b_ : (1u << 23)); const uint32_t kb_ = (uint32_t)(b_ * 4); launch; } } while (0)
#define G256(n) harc_grid256((uint64_t)(n)), dim3(256), 0, c->stream
static int digest_range(harc_amd_ctx *c, const void *base, uint64_t nbytes, uint64_t salt, unsigned long long *d_out)
{
    if (!nbytes) return HARC_AMD_OK;
    if ((uintptr_t)base & 7) { harc_set_error("stream digest: a stream does not start on an 8-byte boundary"); return HARC_AMD_EINTERNAL; }
    const uint64_t nw = (nbytes + 7) / 8;
    const unsigned nb = (unsigned)(nw / 1024 + 1 < 4096 ? nw / 1024 + 1 : 4096);
    hipLaunchKernelGGL(k_stream_digest, dim3(nb), dim3(256), 0, c->stream, (const uint8_t *)base, nbytes, salt, d_out);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}

// The host side of stage II.  On one GPU (and on every rank of a bucket-sharded run) the function covers all encoder shards.  In a design-(R)
// run (harc_amd_replicate_exchange: every rank holds the stage-I result of the WHOLE job) the work is partitioned over the ranks by encoder
// shard -- contigs never cross a shard start (encoder.cpp:171-180), so a rank's shards are a self-contained piece of the global column axis:
//   replicated (cheap or global by nature): the candidates and their two dictionaries, the contig structure (prefix sums over all reads), the
//     sort of the accepted candidates, the window passes over bins above maxsearch (rare; their events travel with their window words);
//   per rank: consensus, realignment proposals, merge, noise / pos / rev / order streams of ITS shards [e0, e1), and the leftovers (unaligned
//     singletons and N reads) of ITS share of the candidates [t0, t1);
//   ONE ncclAllReduce(min) over best[] (one packed (column, direction, dictionary) tuple per candidate): a candidate goes to the
//     smallest tuple that accepts it wherever the window lies (encoder.cpp:252-410 at num_thr = 1).
// Every rank ends with the stream files of its shards -- byte for byte the single-GPU ones -- and its parts of the whole-job files, which
// harc_amd_merge_shard_files lays out (aligned parts in rank order = shard order, then the unaligned parts in candidate order).
int stage2_run(harc_amd_ctx *c)
{
    const harc_amd_params &P = c->P;
    const int L = P.readlen, W = c->W, W3 = c->W3;
    const uint32_t M = c->M, S = c->S, NN = c->NN, T = S + NN, E = (uint32_t)P.num_thr;
    // drop earlier stage-II outputs
    for (auto it = c->out.begin(); it != c->out.end();) { if (it->first.first >= HARC_AMD_S2_SEQ) it = c->out.erase(it); else ++it; }
    const bool part = c->s2_part;                                 // partitioned over the ranks (harc_amd_encode decided)
    HarcComm *const cm = (part && c->comm && c->replicated) ? c->comm : nullptr;       // null with HARC_AMD_S2_SIM (experiment builds): one rank's share without peers (profiling); world 1 with HARC_AMD_S2_PART=2 (tests)
    const uint32_t pw = part ? (uint32_t)c->s2_world : 1u, pr = part ? (uint32_t)c->s2_rank : 0u;
    const uint32_t e0 = part ? (uint32_t)c->s2_e0 : 0u, e1 = part ? (uint32_t)c->s2_e1 : E;
    const uint32_t t0 = (uint32_t)((uint64_t)T * pr / pw), t1 = (uint32_t)((uint64_t)T * (pr + 1u) / pw);
    const bool trace2 = getenv("HARC_AMD_TRACE") != nullptr;
    struct timespec tl0; clock_gettime(CLOCK_MONOTONIC, &tl0);
    auto lap = [&](const char *what) { if (trace2) { (void)hipStreamSynchronize(c->stream); struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); fprintf(stderr, "[stage II%s] %s: %.1f ms\n", part ? " (partitioned)" : "", what, (t.tv_sec - tl0.tv_sec) * 1e3 + (t.tv_nsec - tl0.tv_nsec) * 1e-6); tl0 = t; } };

    S2Args a; memset(&a, 0, sizeof a);
    a.L = L; a.W = W; a.W3 = W3; a.thresh_s = P.thresh_s; a.maxsearch = P.maxsearch;
    if (L > 50) { a.ds[0] = 0; a.de[0] = 20; a.ds[1] = 21; a.de[1] = 41; }                     // encoder.cpp:132-145
    else { a.ds[0] = 0; a.de[0] = 20 * L / 50; a.ds[1] = 20 * L / 50 + 1; a.de[1] = 41 * L / 50; }
    for (int l = 0; l < 2; l++) a.kbits[l] = 3 * (a.de[l] - a.ds[l] + 1);
    a.M = M; a.S = S; a.T = T;
    a.q = 1u + (uint32_t)((M - 1u) / E);                          // uint32 arithmetic as encoder.cpp:171
    if (a.q == 0) a.q = 1;
    a.oreads = c->d_oreads; a.flag = c->d_flag; a.pos = c->d_pos; a.rc = c->d_rc; a.order = c->d_order;
    const uint32_t i0 = (uint32_t)((uint64_t)e0 * a.q > M ? M : (uint64_t)e0 * a.q), i1 = e1 >= E ? M : (uint32_t)((uint64_t)e1 * a.q > M ? M : (uint64_t)e1 * a.q);   // this rank's reordered reads

    // the two order streams outlive the scratch: worst-case sized, copied to the host only when asked for (harc_amd_get_stream)
    uint32_t *order_out = nullptr, *orderN_out = nullptr;
    RC_TRY(dalloc(c, &order_out, (size_t)(i1 - i0) + (size_t)S + 1)); RC_TRY(dalloc(c, &orderN_out, (size_t)NN + 1));     // every singleton / N read is in at most one rank's lists
    c->d_s2_order = c->d_s2_orderN = nullptr; c->n_s2_order = c->n_s2_orderN = 0;
    PoolScope s2_scope(c);                                        // the scratch goes on every way out (the two order streams above stay)
    // harc_amd_params.stream_digest: every stream is folded into four words where it sits in HBM (k_stream_digest), [0] read_seq(+tail) of
    // all shards, [1] noise + noisepos + pos, [2] rev(+tail) + singleton(+tail) + input_N.dna, [3] the two order streams
    const bool want_digest = P.stream_digest != 0;
    unsigned long long *d_digest = nullptr;
    c->have_digest = false;
    if (want_digest) { RC_TRY(dalloc(c, &d_digest, 4)); HIP_TRY(hipMemsetAsync(d_digest, 0, 32, c->stream)); }

    // ---- candidates: singletons then N reads, 3-bit (readsingletons, encoder.cpp:823-872)
    uint64_t *cand3 = nullptr; uint32_t *cand_order = nullptr; unsigned long long *best = nullptr;
    RC_TRY(dalloc(c, &cand3, (size_t)T * W3 + 1)); RC_TRY(dalloc(c, &cand_order, (size_t)T + 1)); RC_TRY(dalloc(c, &best, (size_t)T + 1));
    if (S) {
        if (c->d_sreads) hipLaunchKernelGGL(k_cand3_from2, G256((size_t)S * W3), c->d_sreads, (const uint32_t *)nullptr, S, L, W, W3, cand3);
        else hipLaunchKernelGGL(k_cand3_from2, G256((size_t)S * W3), c->d_reads, c->d_order_s, S, L, W, W3, cand3);
    }
    if (NN) HIP_TRY(hipMemcpyAsync(cand3 + (size_t)S * W3, c->d_nreads3, (size_t)NN * W3 * 8, hipMemcpyDeviceToDevice, c->stream));
    if (T) hipLaunchKernelGGL(k_cand_order, G256(T), c->d_order_s, S, T, cand_order);
    HIP_TRY(hipMemsetAsync(best, 0xFF, ((size_t)T + 1) * 8, c->stream));
    a.cand3 = cand3; a.cand_order = cand_order; a.best = best;
    // ---- contig structure on the global column axis (all reads, on every rank: prefix sums).  Allocation order is the lifetime order (the pool is a
    //      stack): what the streams need to the end first, then -- in a scope of their own, released after the realignment -- the candidates'
    //      dictionaries, bitmaps, 2-bit copies, the consensus bytes and the window passes' state; the merge / noise arrays then take their place
    //      (configs[3]: 185 -> 140 GB peak)
    uint8_t *head = nullptr; uint32_t *u1 = nullptr; uint64_t *gstart = nullptr; uint32_t *chead = nullptr;
    RC_TRY(dalloc(c, &head, (size_t)M + 1)); RC_TRY(dalloc(c, &u1, (size_t)M + 1)); RC_TRY(dalloc(c, &gstart, (size_t)M + 1));
    uint32_t nC = 0; uint64_t total = 0;
    if (M) {
        PoolScope cscope(c);
        uint32_t *u0 = nullptr; uint64_t *d64 = nullptr;
        RC_TRY(dalloc(c, &u0, (size_t)M + 1)); RC_TRY(dalloc(c, &d64, (size_t)M + 1));
        hipLaunchKernelGGL(k_heads1, G256(M), c->d_flag, M, a.q, head, u0);
        RC_TRY(prim_incl_max_u32(c, u0, u1, M));
        hipLaunchKernelGGL(k_heads2, G256(M), M, head, u1, u0);                  // u0 := head flags as u32
        RC_TRY(prim_excl_scan_u32(c, u0, u1, M));                                // u1 := contig id
        hipLaunchKernelGGL(k_col_steps, G256(M), head, c->d_pos, M, L, d64);
        RC_TRY(prim_incl_scan_u64(c, d64, gstart, M));
        uint32_t lastcid = 0, lasthead = 0; uint64_t lastg = 0;
        HIP_TRY(hipMemcpyAsync(&lastcid, u1 + (M - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&lasthead, u0 + (M - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&lastg, gstart + (M - 1), 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        nC = lastcid + lasthead; total = lastg + (uint64_t)L;
    }
    RC_TRY(dalloc(c, &chead, (size_t)nC + 1));
    if (M) hipLaunchKernelGGL(k_contig_heads, G256(M), head, u1, M, chead);
    a.head = head; a.gstart = gstart; a.chead = chead; a.nC = nC; a.total = total;
    c->C.contigs = nC; c->C.seq_bases = total;
    std::vector<uint64_t> sh_col(E + 1, total), seq_off(E, 0), seq_nb(E, 0), seq_tl(E, 0);
    for (uint32_t e = 0; e <= E; e++) {
        const uint64_t st = (uint64_t)e * a.q; const uint32_t i = e == E ? M : (uint32_t)(st > M ? M : st);
        if (i >= M) { sh_col[e] = total; continue; }
        HIP_TRY(hipMemcpyAsync(&sh_col[e], gstart + i, 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t col0 = sh_col[e0], col1 = sh_col[e1];
    a.col0 = col0; a.col1 = col1; a.tile_base = (uint32_t)(col0 / CTILE);
    if (part) c->C.seq_bases = col1 - col0;                       // this rank's share (the merge adds the ranks up)
    lap("contig structure (all reads, on every rank)");
    // the packed consensus (the noise pass reads it) and the packed read_seq of this rank's shards (on its way to the host beside the realignment)
    uint64_t *cons2 = nullptr;
    const uint64_t ncw = (total + 31) / 32;
    RC_TRY(dalloc(c, &cons2, (size_t)ncw + 2 * W + 4));
    HIP_TRY(hipMemsetAsync(cons2 + ncw, 0, ((size_t)2 * W + 4) * 8, c->stream));
    a.cons2 = cons2;
    uint64_t seq_total = 0;
    for (uint32_t e = e0; e < e1; e++) { const uint64_t cc = sh_col[e + 1] - sh_col[e]; seq_nb[e] = cc / 4; seq_tl[e] = cc % 4; seq_off[e] = seq_total; seq_total += (seq_nb[e] + seq_tl[e] + 15) & ~15ull; }
    uint8_t *seqpk = nullptr; RC_TRY(dalloc(c, &seqpk, (size_t)seq_total + 64));
    PoolScope ascope(c);                                          // ---- from here to the end of the realignment: released before the merge
    uint64_t *cand2 = nullptr, *candN = nullptr;
    RC_TRY(dalloc(c, &cand2, (size_t)T * W + 1)); RC_TRY(dalloc(c, &candN, (size_t)T * W + 1));
    if (T) hipLaunchKernelGGL(k_cand2_from3, G256((size_t)T * W), (const uint64_t *)cand3, T, L, W, W3, cand2, candN);
    a.cand2 = cand2; a.candN = candN;
    a.maxevents = 1u << 20;                                       // 16 MB of events to start with; the pass is repeated with a larger buffer when it asks for one
    if (c->s2_events_hint > a.maxevents) a.maxevents = c->s2_events_hint;   // what the context's last run needed, and a quarter more: no second pass on a repeated workload
    if (const char *e = getenv("HARC_AMD_MAXEVENTS")) { a.maxevents = (uint32_t)strtoul(e, nullptr, 10); if (a.maxevents < 1) a.maxevents = 1; }   // tests: force the growth path
    RC_TRY(dalloc(c, &a.events, (size_t)a.maxevents)); RC_TRY(dalloc(c, &a.nevents, 4));
    HIP_TRY(hipMemsetAsync(a.nevents, 0, 16, c->stream));

    lap("  candidates: 3-bit store, 2-bit copies");
    // ---- dictionaries over the candidates (encoder.cpp:886-992)
    DictDev dict[2];
    unsigned long long *d_big = nullptr; RC_TRY(dalloc(c, &d_big, 1));
    HIP_TRY(hipMemsetAsync(d_big, 0, 8, c->stream));
    uint32_t *bloom[2] = { nullptr, nullptr }; int bloom_shift[2] = { 63, 63 };
    // windows of equal width (read lengths >= 50, encoder.cpp:132-145): one combined bitmap, one lookup per consensus k-mer (k_realign_propose1)
    const bool bloom4 = a.kbits[0] == a.kbits[1] && !getenv("HARC_AMD_BLOOM1");
    bool bloom4_tiled = false;
    int bloom_per_key = 16;
    if (const char *e = getenv("HARC_AMD_BLOOMBITS")) { bloom_per_key = atoi(e); if (bloom_per_key < 1) bloom_per_key = 1; }
    if (T) {
        RC_TRY(harc_dict_alloc(c, &dict[0], T, 0)); RC_TRY(harc_dict_alloc(c, &dict[1], T, dict[0].cap));
        // probes into bins of more than 32 candidates are recorded and scanned by a wave each (k_realign_big: 64 candidates per round trip)
        // instead of lane-serially inside k_realign_propose; above maxsearch that pass is also where the sliding window is exact
        dict[0].bigthresh = dict[1].bigthresh = (uint32_t)P.maxsearch < 32u ? (uint32_t)P.maxsearch : 32u;
        int lb = 16; while (lb < 36 && (1ULL << lb) < (unsigned long long)bloom_per_key * T) lb++;
        if (bloom4) {   // 4-bit entries, 16 per key and two of them set: ~1.5 % of absent k-mers pass per plane
            RC_TRY(dalloc(c, &bloom[0], ((size_t)1 << (lb - 3)) + 1));
            // from 64 MB on (16 M candidates) the bitmap is built from sorted items, every word written once (HARC_AMD_S2BLOOM_TILED=0/1 forces either)
            bloom4_tiled = getenv("HARC_AMD_S2BLOOM_TILED") ? atoi(getenv("HARC_AMD_S2BLOOM_TILED")) != 0 : (((size_t)1 << (lb - 3)) * 4 >= ((size_t)64 << 20));
            if ((uint64_t)T * 4 > 0xFFFFFFFFull || harc_bitmap_tiles((uint64_t)1 << (lb - 3)) + 1 >= (1u << BL_PART_BITS)) bloom4_tiled = false;
            if (!bloom4_tiled) HIP_TRY(hipMemsetAsync(bloom[0], 0, ((size_t)1 << (lb - 3)) * 4, c->stream));
            bloom[1] = bloom[0]; bloom_shift[0] = bloom_shift[1] = 64 - lb;
            a.bloom_lbits = lb - 7;                                            // 128 entries to a 64-byte line
            a.bloom_nwin = (a.kbits[0] / 3 == 21 && !getenv("HARC_AMD_BLOOM4_HASHED")) ? 21 - BLOOM4_M + 1 : 0;    // lines by minimizer (read lengths above 50)
        } else for (int l = 0; l < 2; l++) {                                    // one bit per key and dictionary: ~6 % pass
            RC_TRY(dalloc(c, &bloom[l], ((size_t)1 << (lb - 5)) + 1));
            HIP_TRY(hipMemsetAsync(bloom[l], 0, ((size_t)1 << (lb - 5)) * 4, c->stream));
            bloom_shift[l] = 64 - lb;
        }
        PoolScope kscope(c);
        uint64_t *k0 = nullptr; uint32_t *i0k = nullptr; uint64_t *items = nullptr, *items_tmp = nullptr;
        RC_TRY(dalloc(c, &k0, T)); RC_TRY(dalloc(c, &i0k, T));
        const uint64_t b4words = (uint64_t)1 << (lb - 3);
        if (bloom4 && bloom4_tiled) { RC_TRY(dalloc(c, &items, (size_t)T * 4 + 1)); RC_TRY(dalloc(c, &items_tmp, (size_t)T * 4 + 1)); }
        for (int l = 0; l < 2; l++) {
            hipLaunchKernelGGL(k_key3, G256(T), cand3, T, W3, 3 * a.ds[l], a.kbits[l], k0, i0k);
            if (bloom4 && bloom4_tiled) hipLaunchKernelGGL(k_bloom4_items, G256(T), (const uint64_t *)k0, T, a.bloom_lbits, a.bloom_nwin, l, a.kbits[l] / 3, harc_bitmap_tiles(b4words), items + (size_t)l * 2 * T);
            else if (bloom4) hipLaunchKernelGGL(k_bloom4_set, G256(T), (const uint64_t *)k0, T, bloom[0], a.bloom_lbits, a.bloom_nwin, l, a.kbits[l] / 3);
            else hipLaunchKernelGGL(k_bloom_set, G256(T), (const uint64_t *)k0, T, bloom[l], bloom_shift[l]);
            lap("  keys, bitmap items");
            RC_TRY(harc_dict_build(c, &dict[l], k0, i0k, T, (unsigned)a.kbits[l]));
            hipLaunchKernelGGL(k_count_big_bins, G256(dict[l].cap), dict[l].slots, dict[l].cap, (uint32_t)P.maxsearch, d_big);
            lap("  dictionary built");
        }
        if (bloom4 && bloom4_tiled) {
            RC_TRY(harc_bitmap_from_items(c, items, items_tmp, (size_t)T * 4, b4words, bloom[0]));
            lap("  combined bitmap from sorted items");
            if (getenv("HARC_AMD_S2BLOOM_VERIFY")) {                // tests: word for word what the atomics build
                uint32_t *ref = nullptr; unsigned long long *nd = nullptr, hnd = 0;
                RC_TRY(dalloc(c, &ref, (size_t)b4words + 1)); RC_TRY(dalloc(c, &nd, 1));
                HIP_TRY(hipMemsetAsync(ref, 0, (size_t)b4words * 4, c->stream)); HIP_TRY(hipMemsetAsync(nd, 0, 8, c->stream));
                for (int l = 0; l < 2; l++) {
                    hipLaunchKernelGGL(k_key3, G256(T), cand3, T, W3, 3 * a.ds[l], a.kbits[l], k0, i0k);
                    hipLaunchKernelGGL(k_bloom4_set, G256(T), (const uint64_t *)k0, T, ref, a.bloom_lbits, a.bloom_nwin, l, a.kbits[l] / 3);
                }
                hipLaunchKernelGGL(k_words_differ, harc_grid256(b4words), dim3(256), 0, c->stream, (const uint32_t *)ref, (const uint32_t *)bloom[0], b4words, nd);
                HIP_TRY(hipMemcpyAsync(&hnd, nd, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (hnd) { harc_set_error("stage II bitmap built by tiles differs from the one built with atomics in %llu words", hnd); return HARC_AMD_EINTERNAL; }
            }
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int l = 0; l < 2; l++) { a.slots[l] = dict[l].slots; a.cap[l] = dict[l].cap; a.ids[l] = dict[l].ids; a.bloom[l] = bloom[l]; a.bloom_shift[l] = bloom_shift[l]; }
    lap("candidates and their dictionaries (all candidates, on every rank)");

    // ---- consensus + realignment proposals: the columns of this rank's shards
    uint8_t *h_seq = nullptr;
    struct CopyJoin { hipStream_t s; ~CopyJoin() { (void)hipStreamSynchronize(s); } } copy_join{ c->copy_stream };   // on every way out: nothing of this run is still in flight
    uint8_t *cons = nullptr; RC_TRY(dalloc(c, &cons, (size_t)total + 8));
    a.cons = cons;
    unsigned int nev = 0;
    if (col1 > col0) {
        unsigned long long *cinfo = nullptr; RC_TRY(dalloc(c, &cinfo, (size_t)nC + 1));
        hipLaunchKernelGGL(k_contig_info, G256(nC), a, cinfo);
        const uint32_t ntiles = (uint32_t)((col1 + CTILE - 1) / CTILE) - a.tile_base;      // the tiles that hold a column of [col0, col1)
        uint32_t *tlo = nullptr, *thi = nullptr; RC_TRY(dalloc(c, &tlo, (size_t)ntiles + 1)); RC_TRY(dalloc(c, &thi, (size_t)ntiles + 1));
        hipLaunchKernelGGL(k_consensus_tiles, G256(ntiles), a, ntiles, tlo, thi);
        hipLaunchKernelGGL(k_consensus, dim3(ntiles), dim3(256), (size_t)CCHUNK * (W * 8 + 4), c->stream, a, (const uint32_t *)u1, (const unsigned long long *)cinfo, T ? 1 : 0,
                           (const uint32_t *)tlo, (const uint32_t *)thi);
        { const uint64_t w0 = col0 / 32, w1 = (col1 + 31) / 32; hipLaunchKernelGGL(k_pack_cons2, G256(w1 - w0), (const uint8_t *)a.cons, total, w0, w1 - w0, cons2); }
        {   // read_seq (packbits, encoder.cpp:527-548) is final here -- the realignment below does not touch the consensus -- and the shard
            // boundaries on the column axis follow from gstart alone: it is packed now and goes to the host on the copy stream while the
            // realignment, the merge and the noise kernels run (the largest stream: a quarter of a byte per consensus base)
            const uint64_t soff = seq_total;
            for (uint32_t e = e0; e < e1; e++) {
                const uint64_t cc0 = sh_col[e];
                if (seq_nb[e]) hipLaunchKernelGGL(k_pack2_bytes, G256(seq_nb[e]), cons + cc0, seq_nb[e], seqpk + seq_off[e]);
                if (seq_tl[e]) hipLaunchKernelGGL(k_bases_to_ascii, G256(seq_tl[e]), cons + cc0 + 4 * seq_nb[e], seq_tl[e], seqpk + seq_off[e] + seq_nb[e]);
            }
            HIP_TRY(hipGetLastError());
            if (want_digest) for (uint32_t e = e0; e < e1; e++) RC_TRY(digest_range(c, seqpk + seq_off[e], seq_nb[e] + seq_tl[e], 0x100 + e, d_digest + 0));
            RC_TRY(harc_host_alloc(c, (void **)&h_seq, (size_t)soff));
            HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
            HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
            if (soff) HIP_TRY(hipMemcpyAsync(h_seq, seqpk, (size_t)soff, hipMemcpyDeviceToHost, c->copy_stream));
        }
        if (T) {
            const dim3 rg(ntiles);
            auto propose = [&]() {
                switch (W) {
#define REALIGN_CASE(WW) case WW: if (bloom4 && a.bloom_nwin) hipLaunchKernelGGL((k_realign_propose1<WW, 21 - BLOOM4_M + 1>), rg, dim3(256), 0, c->stream, a); \
                                 else if (bloom4) hipLaunchKernelGGL((k_realign_propose1<WW, 0>), rg, dim3(256), 0, c->stream, a); \
                                 else hipLaunchKernelGGL((k_realign_propose<WW>), rg, dim3(256), 0, c->stream, a); break;
                    REALIGN_CASE(1) REALIGN_CASE(2) REALIGN_CASE(3) REALIGN_CASE(4) REALIGN_CASE(5) REALIGN_CASE(6) REALIGN_CASE(7) REALIGN_CASE(8)
#undef REALIGN_CASE
                }
            };
            propose();
            HIP_TRY(hipMemcpyAsync(&nev, a.nevents, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            { const uint64_t want = (uint64_t)nev + nev / 4; c->s2_events_hint = want > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)want; }
            if (nev > a.maxevents) {
                // more probes into bins above maxsearch than the buffer holds (low-complexity reads against a long consensus): the pass
                // is repeated with a buffer of the size it asked for (what it did to best[] is idempotent)
                a.maxevents = nev;
                RC_TRY(dalloc(c, &a.events, (size_t)a.maxevents));
                HIP_TRY(hipMemsetAsync(a.nevents, 0, 16, c->stream));
                propose();
                HIP_TRY(hipMemcpyAsync(&nev, a.nevents, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (nev > a.maxevents) { harc_set_error("stage II: event count changed between two identical passes (%u > %u)", nev, a.maxevents); return HARC_AMD_EINTERNAL; }
            }
        }
    }
    HIP_TRY(hipGetLastError());
    // trace: the claims folded into one word after the proposals and after the window passes (two runs on the same input must print the same words)
    unsigned long long *d_bdig = nullptr;
    auto best_digest = [&](const char *what) -> int {
        if (!getenv("HARC_AMD_TRACE") || !T) return HARC_AMD_OK;
        if (!d_bdig) RC_TRY(dalloc(c, &d_bdig, 2));
        unsigned long long hv = 0;
        HIP_TRY(hipMemsetAsync(d_bdig, 0, 8, c->stream));
        RC_TRY(digest_range(c, best, (uint64_t)T * 8, 0xBE57ull, d_bdig));
        HIP_TRY(hipMemcpyAsync(&hv, d_bdig, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        fprintf(stderr, "[stage II] digest of the claims %s: %016llx (%u probes into large bins)\n", what, hv, nev);
        return HARC_AMD_OK;
    };
    RC_TRY(best_digest("after the proposals"));
    lap("consensus + read_seq + proposals (this rank's columns)");
    // ---- the window words of the events (k_realign_big reads them instead of the consensus: with the columns partitioned the consensus under an
    //      event of another rank is not here)
    uint64_t *evwin = nullptr;
    if (nev) {
        RC_TRY(dalloc(c, &evwin, (size_t)nev * W3 + 1));
        WAVE_PER_ITEM(nev, hipLaunchKernelGGL(k_ev_windows, dim3(nb_), dim3(256), 0, c->stream, a, (const uint4 *)a.events, nev, evwin, kb_));
    }
    if (cm && T) {
        // ---- the claims of all ranks: ONE all-reduce(min) over the packed tuples; then every rank's events (with their windows) to everybody
        RC_TRY(cm->allreduce_min_u64(c, best, (size_t)T));
        RC_TRY(cm->wait(c, "all-reduce of the singleton / N-read claims"));
        std::vector<uint64_t> allnev((size_t)cm->world);
        { const uint64_t mine = nev; RC_TRY(cm->allgather_u64(c, &mine, 1, allnev.data())); }
        uint64_t tot = 0; for (uint64_t x : allnev) tot += x;
        if (tot > 0xFFFFFFFFull) { harc_set_error("stage II: more than 2^32 probes into large bins"); return HARC_AMD_EINVAL; }
        if (tot) {
            uint4 *ev_all = nullptr; uint64_t *win_all = nullptr;
            RC_TRY(dalloc(c, &ev_all, (size_t)tot + 1)); RC_TRY(dalloc(c, &win_all, (size_t)tot * W3 + 1));
            std::vector<size_t> so[2], sb[2], ro[2], rb[2];
            for (int k = 0; k < 2; k++) { so[k].assign(cm->world, 0); sb[k].resize(cm->world); ro[k].resize(cm->world); rb[k].resize(cm->world); }
            uint64_t acc = 0;
            for (int p = 0; p < cm->world; p++) {
                sb[0][p] = (size_t)nev * 16; sb[1][p] = (size_t)nev * W3 * 8;
                ro[0][p] = (size_t)acc * 16; rb[0][p] = (size_t)allnev[p] * 16; ro[1][p] = (size_t)acc * W3 * 8; rb[1][p] = (size_t)allnev[p] * W3 * 8;
                acc += allnev[p];
            }
            const void *sp[2] = { a.events, evwin }; void *rp[2] = { ev_all, win_all };
            const size_t *sop[2] = { so[0].data(), so[1].data() }, *sbp[2] = { sb[0].data(), sb[1].data() }, *rop[2] = { ro[0].data(), ro[1].data() }, *rbp[2] = { rb[0].data(), rb[1].data() };
            RC_TRY(cm->alltoallv(c, 2, sp, sop, sbp, rp, rop, rbp));
            RC_TRY(cm->wait(c, "all-gather of the probes into large bins"));
            a.events = ev_all; evwin = win_all; nev = (unsigned int)tot;
        } else nev = 0;
    }
    a.evwin = evwin;
    lap("all-reduce of the claims, probes into large bins gathered");
    if (nev) {                                                        // exact sliding-window semantics as a fixed point (k_realign_big)
        unsigned int *d_changed = nullptr; uint32_t *estart = nullptr, *lastver = nullptr; unsigned long long *binmin[2] = { nullptr, nullptr };
        if (total >> (EV_TBITS - 2)) { harc_set_error("stage II: more than 2^%d consensus columns", EV_TBITS - 2); return HARC_AMD_EINVAL; }
        RC_TRY(dalloc(c, &d_changed, 8)); RC_TRY(dalloc(c, &estart, (size_t)nev + 1)); RC_TRY(dalloc(c, &lastver, (size_t)nev + 1));
        HIP_TRY(hipMemsetAsync(estart, 0xFF, ((size_t)nev + 1) * 4, c->stream));
        HIP_TRY(hipMemsetAsync(lastver, 0xFF, ((size_t)nev + 1) * 4, c->stream));
        for (int l = 0; l < 2; l++) { RC_TRY(dalloc(c, &binmin[l], 2 * ((size_t)T + 1))); HIP_TRY(hipMemsetAsync(binmin[l], 0xFF, 2 * ((size_t)T + 1) * 8, c->stream)); }
        a.binmax_on = !(getenv("HARC_AMD_S2_RANGE") && atoi(getenv("HARC_AMD_S2_RANGE")) == 0);
        for (int l = 0; l < 2; l++) { RC_TRY(dalloc(c, &a.binmax[l], 2 * ((size_t)T + 1))); HIP_TRY(hipMemsetAsync(a.binmax[l], 0, 2 * ((size_t)T + 1) * 8, c->stream)); }
        const bool trace = getenv("HARC_AMD_TRACE") != nullptr, nochase = getenv("HARC_AMD_S2_NOCHASE") != nullptr;
        a.trace = trace ? 1 : 0;
        struct timespec tw0; clock_gettime(CLOCK_MONOTONIC, &tw0);
        // events in (bin, tuple) order and their rank inside the bin (k_realign_big's header)
        uint32_t *perm = nullptr, *rank = nullptr, *seglen = nullptr, *order2 = nullptr, *firsts = nullptr, *looklist = nullptr; unsigned int maxrank = 0, *d_nlist = nullptr, nfirsts = 0;
        uint32_t rhi0 = 64, range_end[32];
        if (const char *e = getenv("HARC_AMD_S2_RANK0")) { const int v = atoi(e); rhi0 = v < 1 ? 1u : (uint32_t)v; }       // tests: narrow ranges on small inputs
        for (int k = 0; k < 32; k++) range_end[k] = nev;
        if (!getenv("HARC_AMD_S2_FLATPASSES")) {
            RC_TRY(dalloc(c, &perm, (size_t)nev + 1)); RC_TRY(dalloc(c, &rank, (size_t)nev + 1));
            uint64_t *k0 = nullptr, *k1 = nullptr; uint32_t *i0e = nullptr, *i1e = nullptr, *hd = nullptr;
            RC_TRY(dalloc(c, &k0, (size_t)nev + 1)); RC_TRY(dalloc(c, &k1, (size_t)nev + 1)); RC_TRY(dalloc(c, &i0e, (size_t)nev + 1)); RC_TRY(dalloc(c, &i1e, (size_t)nev + 1)); RC_TRY(dalloc(c, &hd, (size_t)nev + 1));
            unsigned tbits = 2; while (tbits < 64 && (total >> (tbits - 2)) != 0) tbits++;
            hipLaunchKernelGGL(k_ev_key_tuple, G256(nev), (const uint4 *)a.events, nev, k0, i0e);
            RC_TRY(prim_sort_pairs_u64_u32(c, k0, k1, i0e, i1e, nev, tbits));
            hipLaunchKernelGGL(k_ev_key_bin, G256(nev), (const uint4 *)a.events, (const uint32_t *)i1e, nev, k0);
            RC_TRY(prim_sort_pairs_u64_u32(c, k0, k1, i1e, perm, nev, 33));            // stable: tuple order inside a bin
            hipLaunchKernelGGL(k_ev_heads, G256(nev), (const uint64_t *)k1, nev, hd);
            RC_TRY(prim_incl_max_u32(c, hd, rank, nev));
            HIP_TRY(hipMemsetAsync(d_changed, 0, 16, c->stream));
            hipLaunchKernelGGL(k_ev_rank, G256(nev), rank, nev, d_changed + 1);
            RC_TRY(dalloc(c, &seglen, (size_t)nev + 1));
            hipLaunchKernelGGL(k_ev_seglen, G256(nev), (const uint32_t *)rank, nev, seglen);
            RC_TRY(dalloc(c, &firsts, (size_t)nev + 1)); RC_TRY(dalloc(c, &d_nlist, 4)); HIP_TRY(hipMemsetAsync(d_nlist, 0, 16, c->stream));
            hipLaunchKernelGGL(k_ev_firsts, G256(nev), (const uint32_t *)rank, nev, firsts, d_nlist);
            HIP_TRY(hipMemcpyAsync(&nfirsts, d_nlist, 4, hipMemcpyDeviceToHost, c->stream));
            {
                uint4 *evs = nullptr; uint64_t *wins = nullptr; RC_TRY(dalloc(c, &evs, (size_t)nev + 1)); RC_TRY(dalloc(c, &wins, (size_t)nev * W3 + 1));
                hipLaunchKernelGGL(k_ev_gather, G256(nev), (const uint4 *)a.events, (const uint32_t *)perm, nev, evs);
                hipLaunchKernelGGL(k_ev_gather_win, G256((uint64_t)nev * W3), (const uint64_t *)a.evwin, (const uint32_t *)perm, nev, W3, wins);
                a.events = evs; a.evwin = wins;                    // from here on event i is the i-th in (bin, tuple) order (the unordered list is not used again)
            }
            // the events grouped by rank range ((bin, tuple) order kept inside a range), and the size of every range
            unsigned int *hist = nullptr; RC_TRY(dalloc(c, &hist, 32)); HIP_TRY(hipMemsetAsync(hist, 0, 32 * 4, c->stream));
            RC_TRY(dalloc(c, &order2, (size_t)nev + 1));
            hipLaunchKernelGGL(k_ev_range_keys, G256(nev), (const uint32_t *)rank, nev, rhi0, k0, i0e, hist);
            RC_TRY(prim_sort_pairs_u64_u32(c, k0, k1, i0e, order2, nev, 5));
            unsigned int hh[32];
            HIP_TRY(hipMemcpyAsync(hh, hist, sizeof hh, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(&maxrank, d_changed + 1, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            range_end[0] = hh[0];
            for (int k = 1; k < 32; k++) range_end[k] = range_end[k - 1] + hh[k];
        }
        uint64_t npass = 0; uint32_t rlo = 0, rhi = rhi0; int nall = 0, ridx = 0;
        uint32_t *ebot = nullptr; bool ebot_tried = false;           // k_realign_block: how far down every event has tested its bin
        const uint32_t pipeline_from = getenv("HARC_AMD_S2_PIPE") ? (uint32_t)atoi(getenv("HARC_AMD_S2_PIPE")) : 1024u;
        RC_TRY(dalloc(c, &a.bestbin[0], (size_t)T + 1)); RC_TRY(dalloc(c, &a.bestbin[1], (size_t)T + 1));
        for (bool ranges = perm != nullptr;;) {
            HIP_TRY(hipMemsetAsync(d_changed, 0, 4, c->stream)); HIP_TRY(hipMemsetAsync(d_changed + 2, 0, 12, c->stream));
            const uint32_t nact = (ranges && order2) ? range_end[ridx < 31 ? ridx : 31] : nev;      // the events of the ranges reached so far come first in order2
            // (from a million events of a pass on: below that a wave per 64 events leaves the chip empty where a wave per event fills it -- a 3.3 M-read
            // repeat-rich set spent 8.4 instead of 5.0 ms in stage II with the block form throughout; HARC_AMD_S2_BLOCK=1 / 0 force either)
            const bool block = nact > 0 && !getenv("HARC_AMD_S2_TWOKERNELS") && !getenv("HARC_AMD_S2_ONEKERNEL") && (getenv("HARC_AMD_S2_BLOCK") ? atoi(getenv("HARC_AMD_S2_BLOCK")) != 0 : nact >= (1u << 20));
            // the block form walks COMPACTED bins (S2Args.cbb; HARC_AMD_S2_COMPACT=0: the whole bins): who looks is asked first, per bin the earliest tuple among them
            const bool compact = block && firsts && nfirsts && !(getenv("HARC_AMD_S2_COMPACT") && atoi(getenv("HARC_AMD_S2_COMPACT")) == 0);
            if (compact && !a.cbb[0]) for (int l = 0; l < 2; l++) {
                RC_TRY(dalloc(c, &a.cbb[l], (size_t)T + 1)); RC_TRY(dalloc(c, &a.cpos[l], (size_t)T + 1)); RC_TRY(dalloc(c, &a.ccnt[l], (size_t)T + 1)); RC_TRY(dalloc(c, &a.tminbin[l], (size_t)T + 1));
                HIP_TRY(hipMemsetAsync(a.tminbin[l], 0xFF, ((size_t)T + 1) * 8, c->stream));
            }
            a.compact = compact ? 1 : 0;
            if (compact) hipLaunchKernelGGL(k_ev_validate_tmin, G256(nact), a, nact, (const uint32_t *)(ranges ? order2 : nullptr), (const uint32_t *)(perm ? rank : nullptr), ranges ? rhi : 0xFFFFFFFFu, estart,
                                            (const unsigned long long *)binmin[0], (const unsigned long long *)binmin[1], lastver, (uint32_t)npass + 1u, T + 1u);
            if (firsts && nfirsts) hipLaunchKernelGGL(k_bestbin_refresh_bins, dim3(nfirsts < 65535u ? nfirsts : 65535u, (nfirsts + 65534u) / 65535u), dim3(256), 0, c->stream, (const unsigned long long *)best, (const uint32_t *)a.ids[0], (const uint32_t *)a.ids[1],
                                                      (const uint4 *)a.events, (const uint32_t *)firsts, nfirsts, a.bestbin[0], a.bestbin[1], a, compact ? 1 : 0);
            else hipLaunchKernelGGL(k_bestbin_refresh, G256(T), (const unsigned long long *)best, (const uint32_t *)a.ids[0], (const uint32_t *)a.ids[1], T, a.bestbin[0], a.bestbin[1]);
            const bool two_kernels = (nact >= (1u << 20) || getenv("HARC_AMD_S2_TWOKERNELS")) && nact > 0 && !getenv("HARC_AMD_S2_ONEKERNEL");      // (tests force either form)
            // default for large passes: an event per lane (k_realign_block); otherwise, and with HARC_AMD_S2_BLOCK=0 or either of the two variables above: a wave per event
            if (!ebot && !ebot_tried && !(getenv("HARC_AMD_S2_EBOT") && atoi(getenv("HARC_AMD_S2_EBOT")) == 0)) { ebot_tried = true; RC_TRY(dalloc(c, &ebot, (size_t)nev + 1)); HIP_TRY(hipMemsetAsync(ebot, 0xFF, ((size_t)nev + 1) * 4, c->stream)); }
            if (block) {
#define BLOCK_ARGS a, nact, (const uint32_t *)(ranges ? order2 : nullptr), (const uint32_t *)(perm ? rank : nullptr), ranges ? rhi : 0xFFFFFFFFu, estart, d_changed, binmin[0], binmin[1], lastver, (uint32_t)npass + 1u, T + 1u, perm ? 1 : 0, ebot
                if (W3 <= 5) hipLaunchKernelGGL((k_realign_block<5>), G256(nact), BLOCK_ARGS);
                else if (W3 <= 8) hipLaunchKernelGGL((k_realign_block<8>), G256(nact), BLOCK_ARGS);
                else hipLaunchKernelGGL((k_realign_block<HARC_MAXW3>), G256(nact), BLOCK_ARGS);
#undef BLOCK_ARGS
            } else if (two_kernels) {
                if (!looklist) RC_TRY(dalloc(c, &looklist, (size_t)nev + 1));
                if (!d_nlist) RC_TRY(dalloc(c, &d_nlist, 4));
                unsigned int nl = 0;
                HIP_TRY(hipMemsetAsync(d_nlist, 0, 4, c->stream));
                hipLaunchKernelGGL(k_ev_validate, G256(nact), a, nact, (const uint32_t *)(ranges ? order2 : nullptr), (const uint32_t *)perm, (const uint32_t *)rank, ranges ? rhi : 0xFFFFFFFFu, estart,
                                   (const unsigned long long *)binmin[0], (const unsigned long long *)binmin[1], lastver, (uint32_t)npass + 1u, T + 1u, looklist, d_nlist);
                HIP_TRY(hipMemcpyAsync(&nl, d_nlist, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                WAVE_PER_ITEM(nl, hipLaunchKernelGGL(k_realign_list, dim3(nb_), dim3(256), 0, c->stream, a, (const uint32_t *)looklist, nl, estart, d_changed, binmin[0], binmin[1], lastver, (uint32_t)npass + 1u, T + 1u, kb_));
            } else
            WAVE_PER_ITEM(nact, hipLaunchKernelGGL(k_realign_big, dim3(nb_), dim3(256), 0, c->stream, a, nact, estart, d_changed, binmin[0], binmin[1], lastver, (uint32_t)npass + 1u, T + 1u,
                               (const uint32_t *)perm, (const uint32_t *)rank, 0u, ranges ? rhi : 0xFFFFFFFFu, (const uint32_t *)(ranges ? order2 : nullptr), kb_));     // the ranks below rlo are validated in passing: no event is ever left unlooked-at for a pass
            // the chaser, after the passes over everything from the second one on (the ranges settle by themselves; the chains it is for show
            // when the ranges meet); its claims count for the pass (the same stamp).  One wave per bin (firsts)
            if (perm && !nochase && !ranges && nall++ > 0) WAVE_PER_ITEM(nfirsts, hipLaunchKernelGGL(k_realign_chase, dim3(nb_), dim3(256), 0, c->stream, a, nfirsts, estart, d_changed, binmin[0], binmin[1], lastver, (uint32_t)npass + 1u, T + 1u,
                                                     (const uint32_t *)perm, (const uint32_t *)rank, (const uint32_t *)seglen, ranges ? rhi : 0xFFFFFFFFu, kb_, (const uint32_t *)firsts));
            unsigned int chg = 0, nlook = 0, nwaves = 0, nclaim = 0;
            HIP_TRY(hipMemcpyAsync(&chg, d_changed, 4, hipMemcpyDeviceToHost, c->stream));
            if (trace) { HIP_TRY(hipMemcpyAsync(&nlook, d_changed + 2, 4, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(&nwaves, d_changed + 3, 4, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(&nclaim, d_changed + 4, 4, hipMemcpyDeviceToHost, c->stream)); }
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (trace) { struct timespec tw; clock_gettime(CLOCK_MONOTONIC, &tw); fprintf(stderr, "[stage II] window pass %llu over %u events (%u looked in %u waves, %u of them moved a claim), ranks [%u, %u)%s: %s, %.2f ms since the first\n", (unsigned long long)npass, nev, nlook, nwaves, nclaim, ranges ? rlo : 0u, ranges ? rhi : maxrank + 1, ranges ? "" : " (all)", chg ? "claims moved" : "quiet", (tw.tv_sec - tw0.tv_sec) * 1e3 + (tw.tv_nsec - tw0.tv_nsec) * 1e-6); }
            if (++npass > (uint64_t)T + 256) { harc_set_error("stage II: the window passes over the large bins did not settle"); return HARC_AMD_EINTERNAL; }
            // a range is repeated until quiet only while it is small (its passes cost next to nothing and the early events of a bin decide
            // what all later ones see); a large range moves on at once: every later pass validates its events, and those with an
            // earlier claim on their bin look again then -- the separate quiet pass per large range was a third of the time
            if (chg && (!ranges || rhi <= pipeline_from)) continue;
            if (!ranges && !chg) break;                                           // a pass over ALL events changed nothing: the fixed point
            if (!ranges) continue;
            if (rhi > maxrank) ranges = false;                                    // every range has settled: now the passes over everything
            else { rlo = rhi; rhi = rhi > 0x20000000u ? 0xFFFFFFFFu : rhi * 4; ridx++; }
        }
    }
    HIP_TRY(hipGetLastError());

    RC_TRY(best_digest("after the window passes"));
    lap("window passes over the large bins (all probes, on every rank)");
    unsigned long long big = 0;
    HIP_TRY(hipMemcpyAsync(&big, d_big, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                     // nothing of the realignment is still running: its arrays go
    ascope.release_now();
    // ---- leftovers (unaligned singletons and N reads, encoder.cpp:484-499): this rank's share of the candidates.  They depend on the claims alone,
    //      so their bases and text are made NOW and leave for the host beside the merge and the noise passes (configs[3]: 1.4 GB of input_N.dna)
    const uint32_t nt = t1 - t0;
    uint32_t *ls = nullptr, *ln = nullptr, *rs = nullptr, *rn = nullptr;
    RC_TRY(dalloc(c, &ls, (size_t)nt + 1)); RC_TRY(dalloc(c, &ln, (size_t)nt + 1)); RC_TRY(dalloc(c, &rs, (size_t)nt + 1)); RC_TRY(dalloc(c, &rn, (size_t)nt + 1));
    HIP_TRY(hipMemsetAsync(ls, 0, ((size_t)nt + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(ln, 0, ((size_t)nt + 1) * 4, c->stream));
    if (nt) hipLaunchKernelGGL(k_left_flags, G256(nt), best, t0, nt, S, ls, ln);
    RC_TRY(prim_excl_scan_u32(c, ls, rs, (size_t)nt + 1)); RC_TRY(prim_excl_scan_u32(c, ln, rn, (size_t)nt + 1));
    uint32_t US = 0, UN = 0;
    HIP_TRY(hipMemcpyAsync(&US, rs + nt, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&UN, rn + nt, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint8_t *sing_bases = nullptr, *spk = nullptr; char *ntext = nullptr;
    const uint64_t sing_nb = (uint64_t)US * L / 4, sing_tl = (uint64_t)US * L % 4;
    const size_t n_ntext = (size_t)UN * (L + 1);
    RC_TRY(dalloc(c, &sing_bases, (size_t)US * L + 8)); RC_TRY(dalloc(c, &ntext, n_ntext + 1)); RC_TRY(dalloc(c, &spk, (size_t)(sing_nb + sing_tl) + 64));
    uint8_t *h_sing = nullptr, *h_ntext = nullptr;
    RC_TRY(harc_host_alloc(c, (void **)&h_sing, (size_t)(sing_nb + sing_tl))); RC_TRY(harc_host_alloc(c, (void **)&h_ntext, n_ntext));
    if (getenv("HARC_AMD_LEFT_ALL")) { if (nt) hipLaunchKernelGGL(k_left_emit, G256((uint64_t)nt * ((L + 15) / 16)), a, t0, nt, rs, rn, sing_bases, ntext); }     // tests: over all candidates, as before
    else if (US + UN) {
        uint32_t *llist = nullptr;
        RC_TRY(dalloc(c, &llist, (size_t)US + UN + 1));
        hipLaunchKernelGGL(k_left_list, G256(nt), (const uint32_t *)ls, (const uint32_t *)ln, (const uint32_t *)rs, (const uint32_t *)rn, nt, US, llist);
        WAVE_PER_ITEM(US + UN, hipLaunchKernelGGL(k_left_emit_w, dim3(nb_), dim3(256), 0, c->stream, a, t0, (const uint32_t *)llist, US + UN, US, sing_bases, ntext, kb_));
    }
    if (sing_nb) hipLaunchKernelGGL(k_pack2_bytes, G256(sing_nb), sing_bases, sing_nb, spk);
    if (sing_tl) hipLaunchKernelGGL(k_bases_to_ascii, G256(sing_tl), sing_bases + 4 * sing_nb, sing_tl, spk + sing_nb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
    if (sing_nb + sing_tl) HIP_TRY(hipMemcpyAsync(h_sing, spk, (size_t)(sing_nb + sing_tl), hipMemcpyDeviceToHost, c->copy_stream));
    if (n_ntext) HIP_TRY(hipMemcpyAsync(h_ntext, ntext, n_ntext, hipMemcpyDeviceToHost, c->copy_stream));
    // ---- accepted candidates sorted by (tuple, rid descending): all of them, on every rank (a sort of 16-byte pairs)
    uint32_t A = 0;
    uint32_t *ta = nullptr, *tb = nullptr; uint64_t *tup0 = nullptr, *tup = nullptr; uint32_t *rid0 = nullptr, *rid = nullptr;
    RC_TRY(dalloc(c, &ta, (size_t)T + 1)); RC_TRY(dalloc(c, &tb, (size_t)T + 1));
    if (T) {
        HIP_TRY(hipMemsetAsync(ta, 0, ((size_t)T + 1) * 4, c->stream));
        hipLaunchKernelGGL(k_acc_flags, G256(T), best, T, ta);
        RC_TRY(prim_excl_scan_u32(c, ta, tb, (size_t)T + 1));
        HIP_TRY(hipMemcpyAsync(&A, tb + T, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    RC_TRY(dalloc(c, &tup0, (size_t)A + 1)); RC_TRY(dalloc(c, &tup, (size_t)A + 1)); RC_TRY(dalloc(c, &rid0, (size_t)A + 1)); RC_TRY(dalloc(c, &rid, (size_t)A + 1));
    if (A) {
        hipLaunchKernelGGL(k_acc_compact, G256(T), best, tb, T, A, tup0, rid0);
        unsigned bits = 2; while (bits < 64 && (total >> (bits - 2)) != 0) bits++;
        RC_TRY(prim_sort_pairs_u64_u32(c, tup0, tup, rid0, rid, A, bits));
    }

    lap("accepted candidates sorted (all, on every rank)");
    // ---- shard boundaries in final-list coordinates; this rank's piece of the final read list is [fbase, fend)
    std::vector<uint32_t> sh_f(E + 1), sh_a(E + 1);
    {
        uint32_t *d_shf = nullptr, *d_sha = nullptr; uint64_t *d_shc = nullptr;
        RC_TRY(dalloc(c, &d_shf, (size_t)E + 2)); RC_TRY(dalloc(c, &d_sha, (size_t)E + 2)); RC_TRY(dalloc(c, &d_shc, (size_t)E + 2));
        hipLaunchKernelGGL(k_shard_bounds, G256(E + 1), (const uint64_t *)gstart, M, a.q, E, (const uint64_t *)tup, A, total, d_shf, d_sha, d_shc);
        HIP_TRY(hipMemcpyAsync(sh_f.data(), d_shf, ((size_t)E + 1) * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(sh_a.data(), d_sha, ((size_t)E + 1) * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    const uint32_t fbase = sh_f[e0], fend = sh_f[e1], F = fend - fbase, na = sh_a[e1] - sh_a[e0];
    if ((uint64_t)(i1 - i0) + na != F) { harc_set_error("stage II bookkeeping: %u reads + %u accepted candidates in a piece of %u", i1 - i0, na, F); return HARC_AMD_EINTERNAL; }

    // ---- merge into the final read list, noise / pos / order / rc streams: GROUP BY GROUP.  A group is one encoder shard (all of this rank's shards
    //      when the output is small): its reads and accepted candidates are merged, sized (writecontig's sizes first, encoder.cpp:654-717),
    //      scanned and emitted, and its noise / noisepos / pos bytes leave on the copy stream while the next group is merged and sized --
    //      the host only waits for a group's two totals.  (Until round 4 the whole piece was merged, sized and scanned before the first byte
    //      left: configs[3] has 4 GB of these streams, 80 ms of PCIe, behind 65 ms of kernels.)
    FinalArrays f;
    RC_TRY(dalloc(c, &f.ref, (size_t)F + 1)); RC_TRY(dalloc(c, &f.kind, (size_t)F + 1)); RC_TRY(dalloc(c, &f.g, (size_t)F + 1));
    uint32_t *nm = nullptr, *nonN = nullptr, *nonNrank = nullptr; uint64_t *nmoff = nullptr;
    RC_TRY(dalloc(c, &nm, (size_t)F + 1)); RC_TRY(dalloc(c, &nonN, (size_t)F + 1)); RC_TRY(dalloc(c, &nonNrank, (size_t)F + 1)); RC_TRY(dalloc(c, &nmoff, (size_t)F + 1));
    HIP_TRY(hipMemsetAsync(nm, 0, ((size_t)F + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(nonN, 0, ((size_t)F + 1) * 4, c->stream));
    uint8_t *posb = nullptr, *rcb = nullptr;
    RC_TRY(dalloc(c, &posb, (size_t)F + 1)); RC_TRY(dalloc(c, &rcb, (size_t)F + 8));
    uint8_t *h_packed = nullptr, *h_pos = nullptr, *h_meta = nullptr;
    RC_TRY(harc_host_alloc(c, (void **)&h_pos, F)); RC_TRY(harc_host_alloc(c, (void **)&h_meta, 32));
    unsigned long long *h_tot = nullptr; RC_TRY(harc_host_alloc(c, (void **)&h_tot, 16 * ((size_t)E + 2)));      // pinned: a group's totals and shard cuts
    std::vector<uint64_t> sh_nm(E + 1, 0);                        // noise coordinates of a shard's first read, relative to its GROUP
    std::vector<uint8_t *> g_noise(E, nullptr), g_noisepos(E, nullptr), gh_noise(E, nullptr), gh_noisepos(E, nullptr);      // per group (indexed by its first shard): device and host buffers
    std::vector<uint32_t> g_of(E, 0);                             // the first shard of the group a shard belongs to
    std::vector<uint64_t> gtot(E, 0);                             // noise entries of a group
    std::vector<uint64_t> s_nz0(E, 0), s_nz1(E, 0), s_np0(E, 0), s_np1(E, 0);      // a shard's noise / noisepos bytes inside its group's buffers
    long long *mcut = nullptr; RC_TRY(dalloc(c, &mcut, (size_t)(i1 - i0) / 256 + 4));      // k_merge_cuts: per block of 256 reads of a group
    uint64_t nmtot = 0; uint32_t n_nonN = 0, n_N_aligned = 0;
    const uint32_t estep = (uint64_t)F * 3 >= ((uint64_t)32 << 20) ? 1u : (e1 - e0 ? e1 - e0 : 1u);       // small outputs: one group (a launch and copies per shard cost a 3 M-read input more than they hide)
    for (uint32_t ea = e0; ea < e1; ea += estep) {
        const uint32_t eb = ea + estep < e1 ? ea + estep : e1;
        for (uint32_t e = ea; e < eb; e++) g_of[e] = ea;
        const uint32_t fa = sh_f[ea] - fbase, fb = sh_f[eb] - fbase;
        if (fb <= fa) continue;
        const uint32_t ia = (uint32_t)((uint64_t)ea * a.q > M ? M : (uint64_t)ea * a.q), ib = eb >= E ? M : (uint32_t)((uint64_t)eb * a.q > M ? M : (uint64_t)eb * a.q);
        if (ib > ia) {
            hipLaunchKernelGGL(k_merge_cuts, G256((ib - ia + 255u) / 256u + 1u), gstart, ia, ib - ia, tup, A, mcut);
            hipLaunchKernelGGL(k_merge_orig, G256(ib - ia), gstart, ia, ib - ia, tup, A, f, fbase, (const long long *)mcut);
        }
        if (sh_a[eb] > sh_a[ea]) hipLaunchKernelGGL(k_merge_acc, G256(sh_a[eb] - sh_a[ea]), gstart, M, tup, rid, sh_a[ea], sh_a[eb] - sh_a[ea], f, fbase);
        launch_noise<false>(c, a, f, cons2, fa, fb, nm, nonN, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        // exclusive scans over the group (nm[fb] = nonN[fb] = 0 still: the next group writes them after this)
        RC_TRY(prim_excl_scan_u32_to_u64(c, nm + fa, nmoff + fa, (size_t)(fb - fa) + 1));
        RC_TRY(prim_excl_scan_u32(c, nonN + fa, nonNrank + fa, (size_t)(fb - fa) + 1));
        HIP_TRY(hipMemcpyAsync(&h_tot[0], nmoff + fb, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<uint32_t *>(&h_tot[1]), nonNrank + fb, 4, hipMemcpyDeviceToHost, c->stream));
        for (uint32_t e = ea + 1; e < eb; e++) HIP_TRY(hipMemcpyAsync(&h_tot[2 + (e - ea)], nmoff + (sh_f[e] - fbase), 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));                 // the copies of the groups before go on meanwhile (copy stream)
        const uint64_t gnm = h_tot[0]; const uint32_t gnon = *reinterpret_cast<uint32_t *>(&h_tot[1]), gF = fb - fa;
        sh_nm[ea] = 0; for (uint32_t e = ea + 1; e < eb; e++) sh_nm[e] = h_tot[2 + (e - ea)];
        RC_TRY(dalloc(c, &g_noise[ea], (size_t)gnm + gF + 8)); RC_TRY(dalloc(c, &g_noisepos[ea], (size_t)gnm + 8));
        RC_TRY(harc_host_alloc(c, (void **)&gh_noise[ea], (size_t)gnm + gF)); RC_TRY(harc_host_alloc(c, (void **)&gh_noisepos[ea], (size_t)gnm));
        // k_noise indexes noise with nmoff[i] + i and the N-read orders with i - nonNrank[i], i counted from the rank's piece: the bases move by fa
        launch_noise<true>(c, a, f, cons2, fa, fb, nullptr, nullptr, nmoff, nonNrank, g_noise[ea] - fa, g_noisepos[ea], posb, rcb, order_out + n_nonN, orderN_out + n_N_aligned - fa);
        HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
        if (gnm + gF) HIP_TRY(hipMemcpyAsync(gh_noise[ea], g_noise[ea], (size_t)gnm + gF, hipMemcpyDeviceToHost, c->copy_stream));
        if (gnm) HIP_TRY(hipMemcpyAsync(gh_noisepos[ea], g_noisepos[ea], (size_t)gnm, hipMemcpyDeviceToHost, c->copy_stream));
        HIP_TRY(hipMemcpyAsync(h_pos + fa, posb + fa, (size_t)gF, hipMemcpyDeviceToHost, c->copy_stream));
        nmtot += gnm; n_nonN += gnon; n_N_aligned += gF - gnon;     // (nmtot: the trace's total)
        gtot[ea] = gnm;
        for (uint32_t e = ea; e < eb; e++) {
            const uint64_t n0 = sh_nm[e], n1 = e + 1 < eb ? sh_nm[e + 1] : gnm;
            s_np0[e] = n0; s_np1[e] = n1; s_nz0[e] = n0 + (sh_f[e] - sh_f[ea]); s_nz1[e] = n1 + (sh_f[e + 1] - sh_f[ea]);
        }
    }
    if ((size_t)n_nonN + US > (size_t)(i1 - i0) + (size_t)S || (size_t)n_N_aligned + UN > (size_t)NN) { harc_set_error("stage II bookkeeping: %u + %u clean, %u + %u N order entries", n_nonN, US, n_N_aligned, UN); return HARC_AMD_EINTERNAL; }
    const size_t n_order = ((size_t)n_nonN + US) * 4, n_orderN = ((size_t)n_N_aligned + UN) * 4;
    if (nt) hipLaunchKernelGGL(k_left_orders, G256(nt), a, t0, nt, rs, rn, order_out, n_nonN, orderN_out, n_N_aligned);
    HIP_TRY(hipGetLastError());
    if (c->d_gid && !c->s1_from_files) {
        // multi-GPU shard (harc_amd_shard_exchange): the order streams carry the GLOBAL ids of the reads, so that the merged archive
        // restores the original order like a single-GPU one (decoder_preserve.cpp:246-290, :212-244)
        unsigned int *d_merr = nullptr; RC_TRY(dalloc(c, &d_merr, 4));
        HIP_TRY(hipMemsetAsync(d_merr, 0, 16, c->stream));
        RC_TRY(shard_map_ids(c, order_out, (uint64_t)n_nonN + US, c->d_gid, c->N, d_merr));
        RC_TRY(shard_map_ids(c, orderN_out, (uint64_t)n_N_aligned + UN, c->d_ngid, c->NN, d_merr));
        unsigned int merr = 0;
        HIP_TRY(hipMemcpyAsync(&merr, d_merr, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (merr) { harc_set_error("stage II: %u order entries outside the shard", merr); return HARC_AMD_EINTERNAL; }
    }

    // ---- packbits per shard into ONE device buffer, then a handful of device -> pinned-host copies; per-shard streams are slices
    std::vector<uint64_t> rev_off(E, 0), rev_nb(E, 0), rev_tl(E, 0);
    uint64_t poff = 0;
    for (uint32_t e = e0; e < e1; e++) {
        const uint64_t ff = sh_f[e + 1] - sh_f[e]; rev_nb[e] = ff / 8; rev_tl[e] = ff % 8; rev_off[e] = poff; poff += (rev_nb[e] + rev_tl[e] + 15) & ~15ull;
    }
    uint8_t *packed = nullptr; RC_TRY(dalloc(c, &packed, (size_t)poff + 64));
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t f0 = sh_f[e] - fbase;
        if (rev_nb[e]) hipLaunchKernelGGL(k_pack1_bytes, G256(rev_nb[e]), rcb + f0, rev_nb[e], packed + rev_off[e]);
        if (rev_tl[e]) HIP_TRY(hipMemcpyAsync(packed + rev_off[e] + rev_nb[e], rcb + f0 + 8 * rev_nb[e], rev_tl[e], hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipGetLastError());
    RC_TRY(harc_host_alloc(c, (void **)&h_packed, (size_t)poff));
    // it follows the shard copies on the copy stream (one PCIe link: side by side they would only share it)
    HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
    if (poff) HIP_TRY(hipMemcpyAsync(h_packed, packed, (size_t)poff, hipMemcpyDeviceToHost, c->copy_stream));
    if (want_digest) {
        // noise / noisepos group by group (a group's buffers start on an 8-byte boundary), pos as a whole; then the shard cuts themselves
        for (uint32_t e = e0; e < e1; e++) if (g_of[e] == e && g_noise[e]) {
            uint32_t eb = e; while (eb < e1 && g_of[eb] == e) eb++;
            RC_TRY(digest_range(c, g_noise[e], gtot[e] + (sh_f[eb] - sh_f[e]), 0x200 + 16 * e, d_digest + 1)); RC_TRY(digest_range(c, g_noisepos[e], gtot[e], 0x201 + 16 * e, d_digest + 1));
        }
        RC_TRY(digest_range(c, posb, F, 0x202, d_digest + 1));
        for (uint32_t e = e0; e < e1; e++) RC_TRY(digest_range(c, packed + rev_off[e], rev_nb[e] + rev_tl[e], 0x300 + e, d_digest + 2));
        RC_TRY(digest_range(c, spk, sing_nb + sing_tl, 0x3F0, d_digest + 2)); RC_TRY(digest_range(c, ntext, n_ntext, 0x3F1, d_digest + 2));
        RC_TRY(digest_range(c, order_out, n_order, 0x400, d_digest + 3)); RC_TRY(digest_range(c, orderN_out, n_orderN, 0x401, d_digest + 3));
        unsigned long long *h_dig = nullptr; RC_TRY(harc_host_alloc(c, (void **)&h_dig, 32));
        HIP_TRY(hipMemcpyAsync(h_dig, d_digest, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 4; k++) c->digest[k] = h_dig[k];
        for (uint32_t e = e0; e <= e1; e++) { c->digest[1] += mix64(0x2F0ull + e + ((uint64_t)(sh_f[e] - fbase) << 20)) + (e < e1 ? mix64(0x2F8ull + e + (s_np1[e] << 20)) : 0ull); c->digest[0] += mix64(0x1F0ull + e + ((sh_col[e] - col0) << 20)); }
        c->have_digest = true;
    }
    for (uint32_t e = 0; e < E; e++) {
        if (e < e0 || e >= e1) {                                  // another rank's shard: nothing of it is here
            for (int id : { HARC_AMD_S2_SEQ, HARC_AMD_S2_SEQ_TAIL, HARC_AMD_S2_REV, HARC_AMD_S2_REV_TAIL, HARC_AMD_S2_POS, HARC_AMD_S2_NOISE, HARC_AMD_S2_NOISEPOS }) out_slice(c, id, e, h_meta, 0);
            continue;
        }
        const uint32_t f0 = sh_f[e] - fbase, f1 = sh_f[e + 1] - fbase;
        out_slice(c, HARC_AMD_S2_SEQ, e, h_seq + seq_off[e], seq_nb[e]); out_slice(c, HARC_AMD_S2_SEQ_TAIL, e, h_seq + seq_off[e] + seq_nb[e], seq_tl[e]);
        out_slice(c, HARC_AMD_S2_REV, e, h_packed + rev_off[e], rev_nb[e]); out_slice(c, HARC_AMD_S2_REV_TAIL, e, h_packed + rev_off[e] + rev_nb[e], rev_tl[e]);
        out_slice(c, HARC_AMD_S2_POS, e, h_pos + f0, f1 - f0);
        const uint32_t g = g_of[e];
        out_slice(c, HARC_AMD_S2_NOISE, e, gh_noise[g] ? gh_noise[g] + s_nz0[e] : h_meta, gh_noise[g] ? s_nz1[e] - s_nz0[e] : 0);
        out_slice(c, HARC_AMD_S2_NOISEPOS, e, gh_noisepos[g] ? gh_noisepos[g] + s_np0[e] : h_meta, gh_noisepos[g] ? s_np1[e] - s_np0[e] : 0);
    }
    out_slice(c, HARC_AMD_S2_SINGLETON, 0, h_sing, sing_nb); out_slice(c, HARC_AMD_S2_SINGLETON_TAIL, 0, h_sing + sing_nb, sing_tl);
    c->d_s2_order = order_out; c->n_s2_order = n_order; c->d_s2_orderN = orderN_out; c->n_s2_orderN = n_orderN;
    out_slice(c, HARC_AMD_S2_INPUT_N, 0, h_ntext, n_ntext);
    { const int ml = snprintf((char *)h_meta, 32, "%d\n", L); out_slice(c, HARC_AMD_S2_META, 0, h_meta, (size_t)ml); }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));                // read_seq has arrived
    if (trace2) fprintf(stderr, "[stage II] %llu noise entries, %u reads in this rank's piece\n", (unsigned long long)nmtot, F);
    lap("merge, noise / pos / rev / order streams, leftovers, device -> host (this rank's shards)");
    c->C.bins_over_maxsearch = big;
    // encoder.cpp:506-508; partitioned: this rank's share of the candidates (the merge adds the ranks up)
    const uint32_t s_lo = t0 < S ? t0 : S, s_hi = t1 < S ? t1 : S;
    c->C.aligned_singletons = (uint64_t)(s_hi - s_lo) - US;
    c->C.aligned_N = (uint64_t)((t1 - t0) - (s_hi - s_lo)) - UN;
    return HARC_AMD_OK;
}

int pack_order_run(harc_amd_ctx *c)
{
    // after harc_amd_shard_exchange the order stream holds GLOBAL ids (< the whole job's clean reads) while `numbits` below is taken from
    // the shard's own entry count: the packed fields would truncate them.  pack_order belongs to the merged read_order.bin (./harc -g).
    if (c->d_gid || c->s2_part) { harc_set_error("pack_order: the context holds a piece of a multi-GPU run; pack the merged read_order.bin instead"); return HARC_AMD_ESTATE; }
    { const void *p0 = nullptr; size_t n0 = 0; if (c->d_s2_order) RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_ORDER, 0, &p0, &n0)); }   // the order stream waits in HBM until wanted
    auto it = c->out.find(std::make_pair((int)HARC_AMD_S2_ORDER, 0));
    if (it == c->out.end()) { harc_set_error("pack_order: no read_order.bin"); return HARC_AMD_ESTATE; }
    const uint8_t *in_p = it->second.ptr ? it->second.ptr : it->second.own.data();
    const size_t in_len = it->second.ptr ? it->second.len : it->second.own.size();
    const uint32_t n = (uint32_t)(in_len / 4);
    if (n == 0) { harc_set_error("pack_order: empty read_order.bin (the reference evaluates log2(0), pack_order.cpp:36)"); return HARC_AMD_EINVAL; }
    int numbits = 0; { uint32_t x = n; while (x) { numbits++; x >>= 1; } }        // (int)(log2(n)+1)
    const uint32_t ng = n / 32;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    PoolScope scope(c);
    RC_TRY(dalloc(c, &d_in, (size_t)n + 1)); RC_TRY(dalloc(c, &d_out, (size_t)ng * numbits + 1));
    HIP_TRY(hipMemcpyAsync(d_in, in_p, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (ng) hipLaunchKernelGGL(k_pack_order, G256((uint64_t)ng * numbits), d_in, ng, numbits, d_out);
    // header + body straight into the pinned output arena (one device->host copy, no staging vectors)
    const size_t body_bytes = (size_t)ng * numbits * 4;
    uint8_t *hp = nullptr;
    RC_TRY(harc_host_alloc(c, (void **)&hp, 8 + body_bytes));
    memcpy(hp, &numbits, 4); memcpy(hp + 4, &n, 4);                               // pack_order.cpp:37-38
    if (body_bytes) HIP_TRY(hipMemcpyAsync(hp + 8, d_out, body_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out_slice(c, HARC_AMD_P_ORDER, 0, hp, 8 + body_bytes);
    { std::vector<uint8_t> tailv(in_p + (size_t)ng * 32 * 4, in_p + (size_t)ng * 32 * 4 + (size_t)(n % 32) * 4); out_buf(c, HARC_AMD_P_ORDER_TAIL, 0) = tailv; }
    return HARC_AMD_OK;
}
