// internal.h -- shared between the translation units of libharc_amd.so (not installed; the public ABI is include/harc_amd.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "../../include/harc_amd.h"

#define HARC_NONE 0xFFFFFFFFu
#define HARC_MAXW 8      // ceil(2*255/64) words of a 2-bit read
#define HARC_MAXW3 12    // ceil(3*255/64) words of a 3-bit read
#define HARC_MAXK (1u << 20)
#define HARC_LOOK_CHUNKS 16   // k_reseed looks for look-ahead seeds in at most this many chunks of 1024 bitmap words below the cursor (oracle: LOOK_CHUNKS)
#ifndef HARC_NSUGG
#define HARC_NSUGG 8      // look-ahead seeds handed to a chain at every reseed (oracle: NSUGG)
#endif

void harc_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) {                                                                                \
            harc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));               \
            return HARC_AMD_ENODEVICE;                                                                         \
        }                                                                                                      \
    } while (0)
#define RC_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != HARC_AMD_OK) return _r; \
    } while (0)

// Stage-I chain header, one per chain (32 B; AoS: a chain's wave reads it with one 32-B access)
struct ChainHdr {
    uint32_t cur;       // read the chain currently sits on
    uint32_t prev;      // pending seed (reorder.cpp `prev`)
    uint32_t flags;     // CH_* bits
    uint32_t mode;      // where the consensus of the next super-round comes from: 0 = cnt[parity], 1 = cnt[parity] + replay of the
                        // `nsteps>>8` kept steps (after a lost bid), 2 = reset from reads[cur] (fresh seed)
    uint32_t n_main;    // records emitted to the main stream so far
    uint32_t n_sing;    // records emitted to the singleton stream so far
    uint32_t nsteps;    // byte0: steps walked in the last k_steps launch; byte1: steps to replay; byte2: look-ahead seeds consumed; byte3: look-ahead seeds held
    uint32_t pad0;      // low 16 bits: 16 x running mean of the priority index of the chain's hits (width of the first probe batch); bits 16-23: look-ahead position reached by the last walk; bits 24-25 / 26-27: the form of the count matrices of parity 0 / 1 (0 = u8, 1 = u16, 2 = u32 counts: cons_store)
};
#define CH_ACTIVE 1u
#define CH_PREVUNM 2u
#define CH_PARITY 4u     // which half of cnt[2][K][Lp] holds the state at the start of the super-round
#define CH_NEEDSEED 8u   // the last walk stopped because a step found no candidate
#define CH_WIDE 32u      // (between the two launches of k_steps_grp only) the chain's column counts need more than the 16 bits the first form keeps in LDS: the u32 form walks it
#define CH_COOP 16u      // the walk stopped in front of a step that has to scan a bin of more than HARC_LARGEBIN reads: k_steps<.., COOP = true> makes that step

// One emitted record of stage I (16 B); scattered into stream order by k_s1_scatter
struct LogRec {
    uint32_t chain;
    uint32_t seq;       // index inside the chain's stream (main or singleton)
    uint32_t rid;
    uint32_t meta;      // pos | flag<<8 | rc<<9 | type<<10 (type 1 = singleton stream)
};

struct HashSlot {       // 16 B, one global_load_dwordx4
    uint64_t key;
    uint32_t start;     // first index into ids[]; the read id itself when SLOT_EMB is set
    uint32_t count;     // 0 = empty slot; low 30 bits = live entries [start, start+count); flags below
};
#define SLOT_DEAD 0x80000000u      // hint: every read of the bin is claimed
#define SLOT_EMB 0x40000000u       // bin holds exactly one read and `start` is its id
#define SLOT_BIG 0x20000000u       // (stage II only) more than maxsearch reads: handled by the sequential sliding-window pass
#define SLOT_OVF 0x10000000u       // (bucketed tables, slot 0 of a 4-slot bucket) some key that hashes to this bucket or passed through it lives further on
#define SLOT_CNT_MASK 0x0FFFFFFFu

struct DictDev {
    HashSlot *slots = nullptr;
    uint64_t cap = 0;          // number of slots
    uint32_t *ids = nullptr;   // read ids sorted by (key, id)
    uint32_t *d_nbins = nullptr;
    uint32_t nbins = 0;
    unsigned long long *large_list = nullptr; unsigned int *large_n = nullptr; uint32_t large_max = 0, large_tag = 0;   // stage I: bins worth compacting, listed at insert time
    uint32_t bigthresh = 0;    // > 0: bins with more entries get SLOT_BIG     // probing starts at a 64-B bucket of 4 slots (fetched whole by a latency-bound k_steps) instead of at the hashed slot
};

struct harc_amd_ctx {
    harc_amd_params P;
    int W = 0, W3 = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // device -> host copies that run beside the kernels of `stream` (stage II: read_seq)
    hipEvent_t ev_copy = nullptr;
    uint32_t s2_events_hint = 0;           // stage II: probes into large bins the last run recorded (+ 25 %): the event buffer of the next run
    uint64_t s1_seq_key = 0; int s1_seq_choice = -1;   // stage I: which scan of the small bins an earlier run over an input of this shape MEASURED to be faster (stage1_run_w)
    size_t dev_bytes = 0, dev_peak = 0;    // raw allocations + pool high-water mark
    std::vector<void *> owned;             // raw allocations (inputs, rocPRIM scratch), freed in destroy
    std::map<void *, size_t> sizes;
    // Device pool: hipMalloc/hipFree of multi-GB buffers cost more than the kernels they serve, so every per-run buffer is a bump
    // allocation out of a few large chunks that persist across runs; stack discipline through harc_pool_mark / harc_pool_release.
    struct PoolChunk { char *base; size_t size, used; };
    std::vector<PoolChunk> pool;
    size_t pool_cur = 0, pool_total = 0;

    // inputs.  d_reads / d_nreads3 are what the stages read: the context's own inputs (installed by harc_amd_set_*), or, after
    // harc_amd_shard_exchange, the shard this GPU received through the all-to-all (the own inputs stay, a later exchange starts from them)
    uint32_t N = 0;  uint64_t *d_reads = nullptr;      // N x W, 2-bit
    uint32_t NN = 0; uint64_t *d_nreads3 = nullptr;    // NN x W3, 3-bit (reads with N)
    struct InBuf { void *p = nullptr; size_t cap = 0; };  // raw allocation kept across runs, grown when too small
    InBuf own_reads, own_nreads3;                       // the context's own inputs
    uint32_t N_own = 0, NN_own = 0;
    uint64_t nrec_own = 0;                              // FASTQ records behind the own inputs (harc_amd_set_fastq_device), else N_own + NN_own
    // multi-GPU (shard.hip, comm.cpp)
    struct HarcComm *comm = nullptr;
    InBuf x_reads, x_nreads3, x_gid, x_ngid;            // the received shard and the global ids of its reads
    uint32_t *d_gid = nullptr, *d_ngid = nullptr;       // non-null after an exchange: stage II writes global ids into its order streams
    int s2_world = 1, s2_rank = 0;                      // the partition's geometry (the communicator's; or HARC_AMD_S2_SIM=rank/world: one rank's share without peers, profiling only)
    bool s2_part = false; int s2_e0 = 0, s2_e1 = 0;     // stage II partitioned over the ranks of a design-(R) run (harc_amd_encode decides): the encoder shards [s2_e0, s2_e1) are this rank's
    bool replicated = false;                            // after harc_amd_replicate_exchange: the context holds the reads of the WHOLE job in global id order (x_reads / x_nreads3)
                                                        // and stage I partitions the CHAINS over the ranks (stage1.hip)
    uint64_t shard_info[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };  // [0] clean reads of the whole job [1] N reads [2] records [3] this rank's first clean id [4] first N id [5] first record
    // stage-I result, device
    bool have_s1 = false;
    uint32_t M = 0, S = 0;
    uint32_t *d_order = nullptr; uint8_t *d_flag = nullptr, *d_pos = nullptr, *d_rc = nullptr; uint32_t *d_order_s = nullptr;
    // stage-II inputs, device
    uint64_t *d_oreads = nullptr;          // M x W oriented reads (= temp.dna)
    uint64_t *d_sreads = nullptr;          // S x W singleton reads (= temp.dna.singleton), only when set from files
    bool s1_from_files = false;
    bool have_s2 = false;
    // read_order.bin / read_order_N_pe.bin of stage II stay in HBM until somebody asks for them (they are only archived with -p)
    uint32_t *d_s2_order = nullptr, *d_s2_orderN = nullptr; size_t n_s2_order = 0, n_s2_orderN = 0;

    // (stream id, shard) -> bytes: either a slice of the pinned host arena (ptr/len) or an owned vector
    struct OutBuf { const uint8_t *ptr = nullptr; size_t len = 0; std::vector<uint8_t> own; };
    std::map<std::pair<int, int>, OutBuf> out;
    // pinned host arena for the output streams (device -> host at PCIe rate, no per-run allocation); reset with the results
    struct HostChunk { char *base; size_t size, used; };
    std::vector<HostChunk> harena;
    harc_amd_counters C;
    uint64_t digest[4] = { 0, 0, 0, 0 }; bool have_digest = false;   // harc_amd_stream_digest: the stage-II streams of the last encode, folded on the device (params.stream_digest)

    // pinned ring of the file feeder (ingest.hip: FASTQ file -> HBM by several reader threads), kept for the next file
    char *feed_ring = nullptr; size_t feed_ring_bytes = 0;

    // scratch for rocPRIM
    void *d_tmp = nullptr; size_t tmp_bytes = 0;
};

// ---- transport of the multi-GPU exchange (comm.cpp): RCCL in production, a directory of files for tests on one GPU
struct HarcComm {
    int world = 1, rank = 0;
    virtual ~HarcComm() {}
    // every rank contributes n u64 (host memory); out = world * n values, rank-major
    virtual int allgather_u64(harc_amd_ctx *c, const uint64_t *in, int n, uint64_t *out) = 0;
    // ONE all-to-all(v) over `narr` device arrays at once (enqueued on c->stream; wait() below tells when it has finished): for array a, the chunk for peer p is send[a] + soff[a][p] (sbytes[a][p] bytes),
    // the chunk from peer p lands at recv[a] + roff[a][p] (rbytes[a][p] bytes); enqueued on c->stream
    virtual int alltoallv(harc_amd_ctx *c, int narr, const void *const *send, const size_t *const *soff, const size_t *const *sbytes,
                          void *const *recv, const size_t *const *roff, const size_t *const *rbytes) = 0;
    // element-wise minimum over the ranks of n u64 in device memory, in place (ncclAllReduce(ncclMin): the singleton / N-read claims of stage II,
    // one packed (column, direction, dictionary) tuple per candidate); enqueued on c->stream like the all-to-all
    virtual int allreduce_min_u64(harc_amd_ctx *c, unsigned long long *d_buf, size_t n) = 0;
    // everything enqueued on c->stream so far (the collective included) has finished, or the peers did not answer in time (HARC_AMD_ETIMEOUT)
    virtual int wait(harc_amd_ctx *c, const char *what) = 0;
    virtual const char *name() const = 0;
};
int harc_in_reserve(harc_amd_ctx *c, harc_amd_ctx::InBuf *b, size_t bytes);   // raw allocation reused across runs
void harc_reset_shard(harc_amd_ctx *c);
void harc_drop_results(harc_amd_ctx *c);                                       // results of the last run go; inputs (and what came with them) stay                                       // stages read the context's own inputs again

// ---- device memory helpers (api.cpp)
int harc_dev_alloc(harc_amd_ctx *c, void **p, size_t bytes);
void harc_dev_free(harc_amd_ctx *c, void *p);
template <class T> static inline int dalloc(harc_amd_ctx *c, T **p, size_t n) { return harc_dev_alloc(c, (void **)p, n * sizeof(T) + 16); }
int harc_tmp_reserve(harc_amd_ctx *c, size_t bytes);
int harc_raw_alloc(harc_amd_ctx *c, void **p, size_t bytes);     // plain hipMalloc, for buffers that outlive a run
void harc_raw_free(harc_amd_ctx *c, void *p);
typedef unsigned long long harc_mark_t;
harc_mark_t harc_pool_mark(harc_amd_ctx *c);
void harc_pool_release(harc_amd_ctx *c, harc_mark_t m);          // everything allocated after the mark becomes reusable
// scope guard: what was allocated after the mark goes on EVERY way out of the scope (error returns included) unless keep() was called
struct PoolScope {
    harc_amd_ctx *c; harc_mark_t mk; bool armed = true;
    explicit PoolScope(harc_amd_ctx *c_) : c(c_), mk(harc_pool_mark(c_)) {}
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;
    ~PoolScope() { if (armed) harc_pool_release(c, mk); }
    void keep() { armed = false; }
    void release_now() { if (armed) { harc_pool_release(c, mk); armed = false; } }
};

// ---- primitives (prims.hip): thin wrappers over rocPRIM device-wide sort / scan
int prim_sort_pairs_u64_u32(harc_amd_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit);
// a bitmap larger than the caches is built from sorted items instead of with random atomics (stage1.hip: harc_bitmap_from_items)
#define BL_TILE_WORDS 16384u     // 64 KB of bitmap per workgroup = 1024 lines
#define BL_PART_BITS 20          // up to 2^20 - 1 tiles (64 GB of bitmap)
#define harc_bitmap_tiles(nwords) ((uint32_t)(((uint64_t)(nwords) + BL_TILE_WORDS - 1) / BL_TILE_WORDS))      // an item with this tile number sets nothing
// word w of the bitmap, bits b0 and b1 (0..31) of it
#define harc_bitmap_item(w, b0, b1) (((uint64_t)((((uint32_t)(w) % BL_TILE_WORDS) << 10) | ((uint32_t)(b0) << 5) | (uint32_t)(b1)) << BL_PART_BITS) | (uint64_t)((uint32_t)(w) / BL_TILE_WORDS))
int harc_bitmap_from_items(harc_amd_ctx *c, const uint64_t *items, uint64_t *tmp, size_t n, uint64_t nwords, uint32_t *bitmap);
int prim_sort_keys_u64(harc_amd_ctx *c, const uint64_t *kin, uint64_t *kout, size_t n, unsigned end_bit);       // by the low end_bit bits, stable
int prim_excl_scan_u32(harc_amd_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
int prim_excl_scan_u32_to_u64(harc_amd_ctx *c, const uint32_t *in, uint64_t *out, size_t n);
int prim_excl_scan_u8_to_u64(harc_amd_ctx *c, const uint8_t *in, uint64_t *out, size_t n);
int prim_incl_scan_u64(harc_amd_ctx *c, const uint64_t *in, uint64_t *out, size_t n);
int prim_incl_max_u32(harc_amd_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
int prim_incl_max_u64(harc_amd_ctx *c, const uint64_t *in, uint64_t *out, size_t n);

// ---- stages
int s1_pack_ascii(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *d_out);               // 2-bit
int s1_pack3_ascii(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *d_out);              // 3-bit
int s1_unpack_to_ascii(harc_amd_ctx *c, const uint64_t *d_reads, uint32_t n, char *d_out);                           // n x (L+1) text
int s1_bucket_reads(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint32_t *d_out);
int s1_partition_reads(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint64_t *d_out, unsigned long long *d_counts);
// the same for either read store (three_bit: reads with N, canonical minimizer over the windows without N), with the global id
// gid0 + (index of the read) travelling along; d_gid_out may be null
int shard_partition(harc_amd_ctx *c, const uint64_t *d_words, uint32_t n, int nwords, bool three_bit, uint32_t nb, uint32_t gid0,
                    uint64_t *d_out, uint32_t *d_gid_out, unsigned long long *d_counts);
int shard_map_ids(harc_amd_ctx *c, uint32_t *d_v, uint64_t n, const uint32_t *d_map, uint32_t nmap, unsigned int *d_err);   // v[i] = map[v[i]]
int stage1_run(harc_amd_ctx *c);
int stage1_make_oriented(harc_amd_ctx *c, uint32_t i0 = 0, uint32_t i1 = 0xFFFFFFFFu);      // d_oreads[i0, i1) from d_reads/d_order/d_rc (default: all)
int s1_orient(harc_amd_ctx *c, const uint64_t *reads, const uint32_t *order, const uint8_t *rc, uint32_t m, uint64_t *out);
// exact key->bin table over n keys (ids must hold 0..n-1 on entry); allocates d->slots / d->ids / d->d_nbins
int harc_dict_alloc(harc_amd_ctx *c, DictDev *d, uint32_t n, uint64_t cap_like);   // cap_like != 0: that many slots
int harc_dict_build(harc_amd_ctx *c, DictDev *d, uint64_t *keys, uint32_t *ids, uint32_t n, unsigned kbits);   // after harc_dict_alloc
void harc_dict_free(harc_amd_ctx *c, DictDev *d);
int stage2_run(harc_amd_ctx *c);
int pack_order_run(harc_amd_ctx *c);

static inline std::vector<uint8_t> &out_buf(harc_amd_ctx *c, int id, int shard) { harc_amd_ctx::OutBuf &o = c->out[std::make_pair(id, shard)]; o.ptr = nullptr; o.len = 0; return o.own; }
static inline void out_slice(harc_amd_ctx *c, int id, int shard, const void *p, size_t n) { harc_amd_ctx::OutBuf &o = c->out[std::make_pair(id, shard)]; o.own.clear(); o.ptr = (const uint8_t *)p; o.len = n; }
int harc_host_alloc(harc_amd_ctx *c, void **p, size_t bytes);    // pinned, valid until the results are dropped
void harc_host_reset(harc_amd_ctx *c);
int harc_d2h(harc_amd_ctx *c, std::vector<uint8_t> &dst, const void *d_src, size_t bytes);
