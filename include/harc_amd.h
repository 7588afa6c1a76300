/*
 * harc_amd.h -- C-ABI of libharc_amd.so: the MI355X (gfx950) implementation of HARC's hash-based read reordering
 * + reference-delta encoding hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * What each entry point replaces in the reference (file:line into shubhamchandak94/HARC):
 *
 *   harc_amd_reorder_files     == `reorder.out <basedir>`      src/reorder.cpp:100-131, invoked at harc:67
 *   harc_amd_encoder_files     == `encoder.out <basedir>`      src/encoder.cpp:108-152, invoked at harc:69
 *   harc_amd_pack_order_files  == `pack_order.out <basedir>`   src/pack_order.cpp:11-18, invoked at harc:112
 *   harc_amd_compress_files    == harc:65-69 fused (stage I -> stage II handed over in HBM, no temp.dna round trip)
 *   harc_amd_params            == the compile-time macros the bash driver writes to src/config.h (harc:52-63)
 *
 *   In-memory API (what a cgo/JNI/ctypes binding of the same path would bind; used by bench.py and the tests):
 *   harc_amd_create/destroy, harc_amd_set_* (inputs = what reorder.cpp:240-263 readDnaFile / encoder.cpp:823-872
 *   readsingletons parse), harc_amd_reorder (reorder.cpp:277-703), harc_amd_encode (encoder.cpp:154-616),
 *   harc_amd_pack_order (pack_order.cpp:20-77), harc_amd_get_stream (the files of reorder.cpp:722-830 and
 *   encoder.cpp:190-196,457-503 as byte ranges).
 *
 * Conventions: every function returns 0 on success, a negative HARC_AMD_E* code otherwise (the reference's
 * programs return 0 / print a message; the bash driver runs under `set -e`, harc:2).  No exceptions cross the ABI.
 * Input buffers are caller-owned and may be released after the call returns.  Device input buffers must be COMPLETE when
 * the call is made: the library works on its own HIP stream and does not wait for the caller's streams.  Output buffers returned by
 * harc_amd_get_stream are library-owned host memory, valid until the next harc_amd_reorder/encode/pack_order on
 * the same context or harc_amd_destroy.  One context per host thread; one HIP device per context.
 *
 * There is NO CPU fallback: every compute entry point fails with HARC_AMD_ENODEVICE when no gfx950 device is usable.
 */
#ifndef HARC_AMD_H
#define HARC_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HARC_AMD_OK 0
#define HARC_AMD_EINVAL (-1)      /* bad argument / parameter out of range (readlen > 255: harc:46-49, preprocess.cpp:122) */
#define HARC_AMD_ENODEVICE (-2)   /* no usable HIP device, or a HIP call failed (message via harc_amd_last_error) */
#define HARC_AMD_EIO (-3)         /* file contract violated (missing / short file) */
#define HARC_AMD_ESTATE (-4)      /* call order violated (e.g. encode before reorder and without stage-I inputs) */
#define HARC_AMD_ENOMEM (-5)
#define HARC_AMD_ETIMEOUT (-7)    /* multi-GPU: the peers did not answer a collective within HARC_AMD_COMM_TIMEOUT seconds (default 600); the communicator is gone */
#define HARC_AMD_EINTERNAL (-6)    /* an invariant of the library failed (bookkeeping mismatch, a schedule that does not settle): a bug, not an input problem */

/* Runtime equivalent of src/config.h (harc:52-63).  Fill with harc_amd_default_params, then override. */
typedef struct harc_amd_params {
    int32_t readlen;          /* `readlen`; 1..255 */
    int32_t num_thr;          /* `num_thr`: E, the number of encoder shards; visible in the output file suffixes (encoder.cpp:190-196) */
    int32_t num_chains;       /* K concurrent chains of stage I.  1 reproduces the reference at num_thr=1 byte for byte.
                                 0 = auto (throughput mode).  The reference's own num_thr>1 is a race (reorder.cpp:545-552);
                                 K>1 here is the deterministic round-synchronous schedule documented in DESIGN.md. */
    int32_t maxmatch;         /* `maxmatch` = readlen/2 */
    int32_t thresh;           /* `thresh` = 4 */
    int32_t thresh_s;         /* `thresh_s` = 24 */
    int32_t maxsearch;        /* `maxsearch` = 1000 */
    int32_t dict_start[2];    /* `dict1_start`,`dict2_start` */
    int32_t dict_end[2];      /* `dict1_end`,`dict2_end` */
    int32_t device;           /* HIP device ordinal */
    int32_t profile;          /* 1: time every launch of the dominant kernel with HIP events on the context's stream */
    int32_t num_steps;        /* S: speculative chain steps per launch of the chain kernel (1..64; 0 = auto: 16; 64 for one chain; 32 from
                                 16 385 chains on -- and from 2048 chains on where the input is not a low-coverage one (at most 98 % of the reads
                                 alone in their first-dictionary bin) -- where the index's bins of more than 16 reads hold at most 2 % of N
                                 entries: a function of the input alone).  Output is independent of S when num_chains = 1; for num_chains > 1
                                 the pair (K,S) defines the schedule */
    int32_t reads_per_chain;  /* auto mode (num_chains = 0): one chain per this many reads, capped at 65536 chains; 0 = 2048 (inputs below 4 M reads:
                                 up to 2048 chains of at least 1024 reads; low-coverage ones up to 4096 of at least 256).  Inputs that are
                                 already fragmented (one minimizer bucket of a multi-GPU shard) lose nothing with 1024 and run faster. */
    int32_t decode_memory_gb; /* -m of `./harc -d -p` (harc:225, MAX_BIN_SIZE of decoder_preserve.cpp:249-253): the original order is restored in bins of
                                 decode_memory_gb * 2e8 / 7 reads (0 = the driver's default 7; <= 3 counts as 3), and never more than fits in HBM */
    int32_t stream_digest;    /* 1: harc_amd_encode also folds every stage-II stream into four 64-bit words where it sits in HBM (harc_amd_stream_digest):
                                 two runs, two kernel variants or two GPUs are compared at full size without touching gigabytes of host memory */
    int32_t table_slots_per_read; /* 16-byte slots per read in each of the two stage-I dictionaries (the replacement of BooPHF.h's MPHF + startpos, reorder.cpp:277-394):
                                 4, 3 or 2; 0 = the library chooses -- 4 while one table stays below 15 % of the device's memory and a fifth of what is free, else 3,
                                 else 2.  A memory / speed choice, not visible in any output byte: configs[3] on one GPU 167 / 149 / 142 GB at 100 / 98.6 / 95.5 %
                                 of the speed (206 / 184 / 176 bytes per input read) */
} harc_amd_params;

/* Counters: the three numbers the reference prints (reorder.cpp:701, encoder.cpp:506-508) + kernel-side statistics. */
typedef struct harc_amd_counters {
    uint64_t n_clean, n_N;            /* inputs */
    uint64_t n_main, n_singleton;     /* stage I: reads in temp.dna / temp.dna.singleton */
    uint64_t unmatched;               /* "Reordering done, X were unmatched" */
    uint64_t aligned_singletons;      /* "X singleton reads were aligned" */
    uint64_t aligned_N;               /* "X reads with N were aligned" */
    uint64_t chains, rounds;          /* stage I schedule */
    uint64_t probes;                  /* hash-table slots inspected by the chain kernel */
    uint64_t candidates;              /* candidate reads fetched + Hamming-tested by the chain kernel */
    uint64_t conflicts;               /* proposals that lost arbitration */
    uint64_t propose_launches;        /* launches of the dominant kernel (k_propose) */
    double propose_ms;                /* sum of their HIP-event durations (params.profile=1), else 0 */
    double index_ms, chain_ms, encode_ms, total_ms;   /* host wall-clock of the phases, stream-synchronised */
    uint64_t contigs, seq_bases;      /* stage II */
    uint64_t bins_over_maxsearch;     /* stage-II dictionary bins larger than maxsearch: their probes are replayed sequentially
                                         (sliding window of encoder.cpp:293, exact) */
    uint64_t device_bytes_peak;
    uint64_t useful_probes;           /* dictionary keys a strictly sequential scan (reorder.cpp:517-649) would have looked up:
                                         priority index of the winning probe + 1, or all probes of the step on a miss */
    uint64_t candidates_seq;          /* candidates that sequential scan would have fetched and Hamming-tested: those of the probes up to and
                                         including the winning one (`candidates` also counts the speculative ones behind it) */
    /* the two kernels of the chain phase apart (round 6): the main kernel's launches are propose_launches / propose_ms; walks that reach a dictionary
       bin of more than 16 reads (repeat families: up to maxsearch Hamming tests per probe, reorder.cpp:540-556) go through the cooperative kernel,
       whose scans stream bin-ordered mirrors of the reads out of L2 -- another ceiling than the main kernel's random accesses */
    uint64_t coop_launches; double coop_ms;
    uint64_t coop_useful_probes, coop_candidates_seq, coop_candidates;   /* its share of useful_probes / candidates_seq / candidates */
    uint64_t coop_steps, dense_steps; /* chain steps WALKED by the launches of either kernel (kept or rolled back) */
} harc_amd_counters;

/* stream ids for harc_amd_get_stream; names are the reference's file names */
enum {
    /* stage I (reorder.cpp:722-830); shard must be 0 */
    HARC_AMD_S1_ORDER = 0,          /* read_order.bin            u32 per main read */
    HARC_AMD_S1_FLAG = 1,           /* tempflag.txt              '0'/'1' */
    HARC_AMD_S1_POS = 2,            /* temppos.txt               u8 */
    HARC_AMD_S1_RC = 3,             /* read_rev.txt              'd'/'r' */
    HARC_AMD_S1_ORDER_SINGLETON = 4,/* read_order.bin.singleton  u32 */
    HARC_AMD_S1_DNA = 5,            /* temp.dna                  text, produced on demand */
    HARC_AMD_S1_DNA_SINGLETON = 6,  /* temp.dna.singleton        text, produced on demand */
    /* stage II (encoder.cpp); per shard e in [0,num_thr) */
    HARC_AMD_S2_SEQ = 10,           /* read_seq.txt.<e>      2-bit packed */
    HARC_AMD_S2_SEQ_TAIL = 11,      /* read_seq.txt.<e>.tail */
    HARC_AMD_S2_POS = 12,           /* read_pos.txt.<e> */
    HARC_AMD_S2_NOISE = 13,         /* read_noise.txt.<e> */
    HARC_AMD_S2_NOISEPOS = 14,      /* read_noisepos.txt.<e> */
    HARC_AMD_S2_REV = 15,           /* read_rev.txt.<e>      1-bit packed */
    HARC_AMD_S2_REV_TAIL = 16,      /* read_rev.txt.<e>.tail */
    /* stage II, whole job; shard must be 0 */
    HARC_AMD_S2_ORDER = 20,         /* read_order.bin (post-encode; encoder.cpp:458-500) */
    HARC_AMD_S2_ORDER_N_PE = 21,    /* read_order_N_pe.bin */
    HARC_AMD_S2_SINGLETON = 22,     /* read_singleton.txt    2-bit packed */
    HARC_AMD_S2_SINGLETON_TAIL = 23,/* read_singleton.txt.tail */
    HARC_AMD_S2_INPUT_N = 24,       /* input_N.dna rewritten: unaligned N reads (encoder.cpp:493-499) */
    HARC_AMD_S2_META = 25,          /* read_meta.txt */
    /* -p (pack_order.cpp) */
    HARC_AMD_P_ORDER = 30,          /* read_order.bin packed */
    HARC_AMD_P_ORDER_TAIL = 31,     /* read_order.bin.tail */
    /* FASTQ ingest (preprocess.cpp:102) */
    HARC_AMD_IN_ORDER_N = 40        /* read_order_N.bin: original index (u32) of every read with N, after harc_amd_set_fastq_device */
};

typedef struct harc_amd_ctx harc_amd_ctx;

/* harc:52-60 formulae.  num_thr=8 (harc:195), num_chains=0 (auto). */
int harc_amd_default_params(int32_t readlen, harc_amd_params *out);

int harc_amd_create(const harc_amd_params *params, harc_amd_ctx **out);
void harc_amd_destroy(harc_amd_ctx *ctx);
const char *harc_amd_last_error(void);

/* ---- inputs.  Clean reads = the lines of input_clean.dna (alphabet ACGT, preprocess.cpp:104-108). */
/* host ASCII; read i starts at ascii + i*stride (stride = readlen+1 for the file layout, reorder.cpp:252) */
int harc_amd_set_reads_ascii(harc_amd_ctx *ctx, const char *ascii, uint32_t n_reads, uint32_t stride);
/* same, buffer already in device memory of params.device */
int harc_amd_set_reads_ascii_device(harc_amd_ctx *ctx, const char *d_ascii, uint32_t n_reads, uint32_t stride);
/* device buffer of n_reads * ceil(2*readlen/64) little-endian u64 words in std::bitset<2*readlen> layout
   (reorder.cpp:184-198: base i at bits 2i,2i+1, A=0 G=1 C=2 T=3); the buffer is copied */
int harc_amd_set_reads_packed_device(harc_amd_ctx *ctx, const uint64_t *d_packed, uint32_t n_reads);
/* reads containing N = the lines of input_N.dna (preprocess.cpp:98-103); host ASCII */
int harc_amd_set_nreads_ascii(harc_amd_ctx *ctx, const char *ascii, uint32_t n_reads, uint32_t stride);
int harc_amd_set_nreads_ascii_device(harc_amd_ctx *ctx, const char *d_ascii, uint32_t n_reads, uint32_t stride);
/* FASTQ text (4-line records) already in device memory: preprocess.cpp:81-121 + readDnaFile on the GPU -- takes line 2 of every
   record, checks the fixed read length, packs reads without N into the 2-bit store and reads with N into the 3-bit store; the
   original indices of the N reads become stream HARC_AMD_IN_ORDER_N.  Replaces harc_amd_set_reads_* + harc_amd_set_nreads_*. */
int harc_amd_set_fastq_device(harc_amd_ctx *ctx, const char *d_fastq, uint64_t n_bytes, uint64_t *n_records_out);
/* stage-II inputs when stage I ran elsewhere (the file family of reorder.cpp:722-830): host buffers.
   dna/dna_s are text with stride readlen+1 */
int harc_amd_set_stage1_streams(harc_amd_ctx *ctx, const char *temp_dna, const uint8_t *flag, const uint8_t *pos,
                                const uint32_t *order, const uint8_t *rc, uint32_t n_main,
                                const char *temp_dna_singleton, const uint32_t *order_singleton, uint32_t n_singleton);

/* ---- multi-GPU sharding helpers (BASELINE.json north_star: reads shard by k-mer bucket, one all-to-all, then every GPU
   runs the whole path on its shard).  Device buffers of params.device. */
/* 2-bit pack n ASCII reads into the caller's buffer of n*ceil(2*readlen/64) u64 (layout of harc_amd_set_reads_packed_device) */
int harc_amd_pack_reads_device(harc_amd_ctx *ctx, const char *d_ascii, uint32_t n_reads, uint32_t stride, uint64_t *d_packed_out);
/* bucket[i] = hash(canonical minimizer, k=15, of the whole read i) % n_buckets: reads that overlap by >= ~half a read
   share a minimizer with high probability and land on the same GPU */
int harc_amd_bucket_reads_device(harc_amd_ctx *ctx, const uint64_t *d_packed, uint32_t n_reads, uint32_t n_buckets, uint32_t *d_bucket_out);
/* the send buffer of the all-to-all in one call: reads grouped by bucket (bucket 0 first, original order inside a bucket) and the
   number of reads per bucket (n_buckets u64, device memory) */
int harc_amd_partition_reads_device(harc_amd_ctx *ctx, const uint64_t *d_packed, uint32_t n_reads, uint32_t n_buckets, uint64_t *d_packed_out, uint64_t *d_counts_out);

/* ---- multi-GPU run: one process (one context) per GPU, RCCL over xGMI.  The reference has no counterpart (its parallelism is OpenMP
   threads over one address space, reorder.cpp:455-457); what it fixes is the archive contract the merged result must meet: stream
   files read_*.txt.<shard> discovered by their count (harc:171, decoder.cpp:90-140), read_order.bin = original index of every
   decoded clean read in stream order, singletons last (decoder_preserve.cpp:246-290), read_order_N_pe.bin likewise for the reads
   with N (:212-244), read_order_N.bin = original record of every N read (merge_N.cpp:37-57).
   Bootstrap mirrors ncclGetUniqueId / ncclCommInitRank: rank 0 obtains an id, the caller ships the bytes to the other ranks (a
   file, MPI, torch.distributed, ...), every rank calls harc_amd_comm_init. */
#define HARC_AMD_COMM_ID_BYTES 128
int harc_amd_comm_get_id(uint8_t *id, size_t id_bytes);                                   /* id_bytes >= HARC_AMD_COMM_ID_BYTES */
int harc_amd_comm_init(harc_amd_ctx *ctx, const uint8_t *id, size_t id_bytes, int32_t world, int32_t rank);   /* ncclCommInitRank on params.device */
/* test transport: chunks travel through files of a shared directory, so that several ranks can share ONE GPU (RCCL refuses that) */
int harc_amd_comm_init_mailbox(harc_amd_ctx *ctx, const char *dir, int32_t world, int32_t rank);
int harc_amd_comm_barrier(harc_amd_ctx *ctx);
int harc_amd_comm_destroy(harc_amd_ctx *ctx);
/* The context holds this rank's slice of the job (harc_amd_set_reads_* / set_nreads_* / set_fastq_device).  Reads are bucketed by
   the hash of their canonical minimizer, grouped by destination and moved with ONE all-to-all(v) -- ncclGroupStart, ncclSend /
   ncclRecv per peer, ncclGroupEnd -- carrying 8W + 4 bytes per clean read (packed read + u32 global id) and 8 W3 + 4 per read with N.
   Afterwards harc_amd_reorder / harc_amd_encode work on the shard and HARC_AMD_S2_ORDER / HARC_AMD_S2_ORDER_N_PE hold GLOBAL ids
   (clean read i of rank r = sum of the clean reads of ranks < r, + i).  The slice itself stays with the context: calling the
   function again repeats the exchange.  info (8 u64, may be NULL): [0] clean reads of the whole job [1] reads with N [2] FASTQ
   records [3] this rank's first clean id [4] first N id [5] first record [6] clean reads received [7] N reads received. */
int harc_amd_shard_exchange(harc_amd_ctx *ctx, uint64_t *info);
/* Design (R): replicate the reads, partition the CHAINS.  One all-gather puts the reads of the whole job on every GPU in global id order;
   harc_amd_reorder then builds the whole index on every GPU, walks only the chains this rank owns and all-gathers the walked steps once per
   super-round; harc_amd_reorder / harc_amd_encode deliver, on EVERY rank, byte for byte what one GPU produces from the concatenated input
   (reorder.cpp:455-457 has one address space: this is its multi-GPU form that keeps the compression ratio; the bucket shard above trades
   ratio for memory and speed).  Every GPU must hold the whole job.  info as for harc_amd_shard_exchange. */
int harc_amd_replicate_exchange(harc_amd_ctx *ctx, uint64_t *info);
/* The stages read the context's own slice again (as before the first exchange); the results of the last run go, the communicator stays. */
int harc_amd_shard_reset(harc_amd_ctx *ctx);

/* ---- compute (all on params.device, asynchronous internally, synchronised before return) */
int harc_amd_reorder(harc_amd_ctx *ctx);      /* index build + chaining: reorder.cpp:277-703 */
int harc_amd_encode(harc_amd_ctx *ctx);       /* encoder.cpp:154-616 on the stage-I result held in HBM (or set_stage1_streams) */
int harc_amd_pack_order(harc_amd_ctx *ctx);   /* pack_order.cpp:20-77 on HARC_AMD_S2_ORDER */

/* ---- outputs */
int harc_amd_get_stream(harc_amd_ctx *ctx, int32_t stream_id, int32_t shard, const void **ptr, size_t *len);
int harc_amd_get_counters(harc_amd_ctx *ctx, harc_amd_counters *out);

/* ---- round-trip check at any size (decode side, decoder.cpp:90-169, restated on the GPU).
   Signature of a read multiset: sig[0] = number of reads, sig[1] = sum and sig[2] = xor of a 64-bit hash of every read
   (FNV-1a over the bases A0 C1 G2 T3 N4, then a 64-bit finaliser).  Order-independent, so the decoded output of any
   (num_chains, num_steps, num_thr) can be compared with the input reads without materialising either. */
int harc_amd_decode_signature(harc_amd_ctx *ctx, uint64_t sig[3]);         /* decodes the context's stage-II streams */
int harc_amd_reads_signature_device(harc_amd_ctx *ctx, const char *d_ascii, uint32_t n_reads, uint32_t stride, uint64_t sig[3]);
int harc_amd_input_signature(harc_amd_ctx *ctx, uint64_t sig[3]);           /* of the clean + N reads the context currently holds */

/* Digest of the stage-II streams of the last harc_amd_encode (needs params.stream_digest = 1), computed on the device from the very buffers the
   streams are copied out of: out[0] read_seq.txt.<e>(+.tail) of all shards and their cuts, out[1] read_noise / read_noisepos / read_pos and their
   cuts, out[2] read_rev.txt.<e>(+.tail), read_singleton.txt(+.tail), input_N.dna, out[3] read_order.bin and read_order_N_pe.bin.  Position-dependent
   64-bit sums: equal streams give equal words, a changed, moved, lost or added byte changes them (encoder.cpp:190-196,457-503 name the files). */
int harc_amd_stream_digest(harc_amd_ctx *ctx, uint64_t out[4]);
/* sha256 (hex) of the kernel sources this library was built from: ties a committed profile (profiles/k_steps_traffic.json) to a build */
const char *harc_amd_build_id(void);
/* 1 when this library was built with the named optional part: "grp" (make GRP=1: k_steps_grp, the walk with several chains per wave), "test_transport"
 * (the file-mailbox transport of the one-GPU multi-rank tests), "experiments" (schedule constants from the environment); 0 otherwise */
int harc_amd_build_has(const char *feature);
/* Self-test of the library's launch geometry (one thread per item over n items through harc_gid / harc_gid32, and four lanes per item in folded workgroups, n beyond 2^32 included: a one-dimensional grid of 2^32 and more
 * work-items is cut short without an error on this platform).  *visited == n and *index_sum == n (n - 1) / 2 mod 2^64 when every item was visited once. */
int harc_amd_selftest_launch(harc_amd_ctx *ctx, uint64_t n, uint64_t *visited, uint64_t *index_sum);

/* Where the wall time of this process's last harc_amd_compress_fastq_files_ex went, in seconds (the end-to-end leg of bench.py; preprocess.cpp:81-121 + harc:50-69):
 * out[0] context + device pool, [1] ingest (file -> HBM -> packed stores; reads, uploads and kernels overlapped), [2] of it the calling thread waiting for
 * the file reader threads, [3] of it the device's line index / classify / pack passes, [4] reorder, [5] encode (the D2H of the streams inside),
 * [6] stream files written, [7] total up to there (the -q files of a run without -p come after it).  n = how many of them the caller wants. */
int harc_amd_last_fastq_timing(double *out, int32_t n);

/* ---- file contract: drop-ins for the reference's stage programs.  basedir as argv[1] of those programs. */
int harc_amd_reorder_files(const harc_amd_params *params, const char *basedir);
int harc_amd_encoder_files(const harc_amd_params *params, const char *basedir);
int harc_amd_compress_files(const harc_amd_params *params, const char *basedir);
int harc_amd_pack_order_files(const harc_amd_params *params, const char *basedir);
/* harc:50-69 in one call: FASTQ file -> read_order_N.bin, numreads.bin and every stage-II file, parsed on the GPU, no
   input_clean.dna / temp.dna round trips */
int harc_amd_compress_fastq_files(const harc_amd_params *params, const char *fastq, const char *basedir);
/* the same with the reference's -p / -q switches (preprocess.out's 3rd and 4th argument, harc:50).  preserve_quality: also writes
   output/output.quality and output/output.id -- in file order with preserve_order (preprocess.cpp:64-69), otherwise gathered by the
   post-encoding orders exactly as reorder_quality.out does (reorder_quality.cpp:47-219, quirks included: see oracle/harc_oracle.c
   harc_oracle_quality).  preserve_order itself changes nothing else here: pack_order stays a separate call (harc:112).
   The file may be larger than HBM in every mode: it is ingested a piece at a time; without preserve_order the quality values and ids are
   then permuted by streaming the file again once per bin of output (reorder_quality.cpp:61-77 bins through host memory). */
int harc_amd_compress_fastq_files_ex(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order, int32_t preserve_quality);
/* == `preprocess.out <fastq> <basedir> <preserve_order> <preserve_quality> <readlen>` (src/preprocess.cpp:22-137, harc:50),
   the N split only; host code, feeds the boundary */
int harc_amd_preprocess_files(const char *fastq, const char *basedir, int32_t readlen);
/* == `decoder.out <basedir> <num_thr> <num_thr_e>` (src/decoder.cpp:44-172, harc:188; non -p): reads the stage-II stream files of
   num_thr_e shards under <basedir>/output and writes output/output.dna, byte-identical to the reference decoder */
int harc_amd_decoder_files(const harc_amd_params *params, const char *basedir, int32_t num_thr_e);
/* == `unpack_order.out` + `decoder_preserve.out` + `merge_N.out` (harc:183-185, -p): needs read_order.bin(+.tail) as written by
   pack_order, read_order_N_pe.bin and read_order_N.bin; writes output/output.dna = the reads in their original FASTQ order.
   The order is restored in bins of output lines (params.decode_memory_gb, the reference's -m; capped by what fits in HBM): every bin
   decodes the streams again and keeps its own lines, so neither HBM nor host memory grows with the archive (the reference bins
   through host memory and a temporary file, decoder_preserve.cpp:246-290) */
int harc_amd_decoder_preserve_files(const harc_amd_params *params, const char *basedir, int32_t num_thr_e);
/* One rank of a multi-GPU `./harc -c -g <world>`: reads its slice of the FASTQ file (cut at record boundaries near rank/world of the
   file), joins the communicator named by comm_spec -- "rccl:<file>" (rank 0 writes the ncclUniqueId there, the others wait for it) or
   "mailbox:<dir>" (test transport) --, exchanges, compresses its shard and writes read_*.txt.<rank*num_thr + e> into
   <basedir>/output plus its part of the whole-job files under <basedir>/output/.shard/.  preserve_quality needs preserve_order
   (quality values and ids then stay in file order, preprocess.cpp:64-69). */
int harc_amd_compress_fastq_shard_files(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                                        int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec);
/* After every rank has finished: the whole-job files of the archive (read_singleton.txt(+.tail), input_N.dna, read_order.bin,
   read_order_N_pe.bin, read_order_N.bin, numreads.bin, read_meta.txt, output.quality / output.id) from the parts under .shard/,
   laid out as encoder.cpp:457-503 writes them: aligned reads of all shards first, then the unaligned ones.  Host code. */
int harc_amd_merge_shard_files(const char *basedir, int32_t world);
/* One rank of `./harc -c -g N` in design-(R) mode (HARC_AMD_MG_MODE=replicate): as harc_amd_compress_fastq_shard_files, but the reads are
   all-gathered (harc_amd_replicate_exchange) and the chains partitioned; rank 0 writes the streams of the whole job -- byte for byte those
   of a single-GPU run, num_thr stream files -- the other ranks only their slice's parts; harc_amd_merge_shard_files finishes the archive. */
int harc_amd_compress_fastq_replicated_files(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                                             int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec);

#ifdef __cplusplus
}
#endif
#endif /* HARC_AMD_H */
