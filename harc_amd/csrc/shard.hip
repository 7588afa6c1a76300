// shard.hip -- the multi-GPU split of the hot path (BASELINE.json north_star; SURVEY.md 8e design B): every rank holds a slice of the
// job's reads; reads are bucketed by the hash of their canonical minimizer (k = 15), grouped by destination, and ONE all-to-all(v)
// moves each read (8W bytes) together with its u32 global id to the GPU that owns its bucket.  After that the ranks are independent:
// index build -> chains -> encode on the shard, with the global ids written into read_order.bin / read_order_N_pe.bin so that the
// merged archive decodes -- in the original order with -p -- like a single-GPU one (decoder_preserve.cpp:246-290, merge_N.cpp:37-57).
// Reads with N (3-bit store) travel the same way; their minimizer runs over the 15-mers without N.
//
// Global ids: clean read i of rank r is  sum_{s<r} N_s + i  (the line of a concatenated input_clean.dna, preprocess.cpp:104-108);
// N read i of rank r is  sum_{s<r} NN_s + i  (the line of a concatenated input_N.dna); read_order_N.bin entries are shifted by the
// records of the lower ranks.
#include "devutil.h"

// bucket of a read with N: canonical minimizer over the windows that hold no N; 3-bit codes A0 N1 G2 C4 T6 -> 2-bit A0 G1 C2 T3 (the
// packed code of the clean store, so that an N read goes where the clean reads around it go); no clean window: hash of the id
__global__ void k_bucket3(const uint64_t *reads3, uint32_t n, int L, int W3, uint32_t nb, uint32_t gid0, uint32_t *out)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t *r = reads3 + (size_t)i * W3;
    const int K = L < 15 ? L : 15;
    const uint64_t kmask = (K < 32) ? (((uint64_t)1 << (2 * K)) - 1) : ~(uint64_t)0;
    uint64_t fw = 0, rv = 0, best = ~(uint64_t)0; int valid = 0; bool any = false;
    for (int b = 0; b < L; b++) {
        const int off = 3 * b, wi = off >> 6, sh = off & 63;
        uint64_t v = r[wi] >> sh;
        if (sh > 61 && wi + 1 < W3) v |= r[wi + 1] << (64 - sh);
        const int c3 = (int)(v & 7);
        if (c3 == 1) { valid = 0; fw = 0; rv = 0; continue; }
        const uint64_t pc = (uint64_t)(c3 >> 1);
        fw = ((fw << 2) | pc) & kmask;
        rv = (rv >> 2) | ((3 - pc) << (2 * (K - 1)));
        if (++valid >= K) { const uint64_t h = mix64(fw < rv ? fw : rv); best = h < best ? h : best; any = true; }
    }
    if (!any) best = mix64((uint64_t)gid0 + i);
    out[i] = (uint32_t)(best % nb);
}
// (bucket, index) keys for the stable radix pass + reads per bucket (one atomic per distinct bucket per wave)
__global__ void k_shard_keys(const uint32_t *bucket, uint32_t n, uint64_t *keys, uint32_t *idx, unsigned long long *counts)
{
    const uint32_t i = harc_gid32();
    const bool in = i < n;
    const uint32_t b = in ? bucket[i] : 0xFFFFFFFFu;
    if (in) { keys[i] = b; idx[i] = i; }
    unsigned long long todo = __ballot(in);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lb = __shfl(b, leader, 64);
        const unsigned long long same = __ballot(in && b == lb);
        if ((threadIdx.x & 63) == leader) atomicAdd(&counts[lb], (unsigned long long)__popcll(same));
        todo &= ~same;
    }
}
__global__ void k_shard_gather(const uint64_t *words, const uint32_t *idx, uint32_t n, int nw, uint32_t gid0, uint64_t *out, uint32_t *gid_out)
{
    const uint64_t t = harc_gid();
    if (t >= (uint64_t)n * nw) return;
    const uint32_t i = (uint32_t)(t / nw); const int w = (int)(t % nw);
    const uint32_t src = idx[i];
    out[t] = words[(size_t)src * nw + w];
    if (gid_out && w == 0) gid_out[i] = gid0 + src;
}
__global__ void k_map_ids(uint32_t *v, uint64_t n, const uint32_t *map, uint32_t nmap, unsigned int *err)
{
    const uint64_t i = harc_gid();
    if (i >= n) return;
    const uint32_t x = v[i];
    if (x >= nmap) { atomicAdd(err, 1u); return; }
    v[i] = map[x];
}
// local ids -> global ids in an order stream of stage II (the ids of the shard this GPU received)
int shard_map_ids(harc_amd_ctx *c, uint32_t *d_v, uint64_t n, const uint32_t *d_map, uint32_t nmap, unsigned int *d_err)
{
    if (!n) return HARC_AMD_OK;
    hipLaunchKernelGGL(k_map_ids, harc_grid256(n), dim3(256), 0, c->stream, d_v, n, d_map, nmap, d_err);
    HIP_TRY(hipGetLastError());
    return HARC_AMD_OK;
}

int s1_bucket_reads(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint32_t *d_out);

int shard_partition(harc_amd_ctx *c, const uint64_t *d_words, uint32_t n, int nw, bool three_bit, uint32_t nb, uint32_t gid0,
                    uint64_t *d_out, uint32_t *d_gid_out, unsigned long long *d_counts)
{
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nb * 8, c->stream));
    if (!n) return HARC_AMD_OK;
    PoolScope scope(c);                                           // temporaries go on every way out
    uint32_t *b = nullptr, *i0 = nullptr, *i1 = nullptr; uint64_t *k0 = nullptr, *k1 = nullptr;
    RC_TRY(dalloc(c, &b, n)); RC_TRY(dalloc(c, &i0, n)); RC_TRY(dalloc(c, &i1, n)); RC_TRY(dalloc(c, &k0, n)); RC_TRY(dalloc(c, &k1, n));
    const dim3 g = harc_grid256(n), t(256);
    if (three_bit) hipLaunchKernelGGL(k_bucket3, g, t, 0, c->stream, d_words, n, c->P.readlen, nw, nb, gid0, b);
    else RC_TRY(s1_bucket_reads(c, d_words, n, nb, b));
    hipLaunchKernelGGL(k_shard_keys, g, t, 0, c->stream, (const uint32_t *)b, n, k0, i0, d_counts);
    unsigned bits = 1; while ((1u << bits) < nb) bits++;
    RC_TRY(prim_sort_pairs_u64_u32(c, k0, k1, i0, i1, n, bits));                 // stable: original order inside a bucket
    hipLaunchKernelGGL(k_shard_gather, harc_grid256((uint64_t)n * nw), t, 0, c->stream, d_words, (const uint32_t *)i1, n, nw, gid0, d_out, d_gid_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}

// The exchange.  In: the context's own inputs (harc_amd_set_reads_* / set_nreads_* / set_fastq_device) = this rank's slice of the job.
// Out: the context's reads are the shard of bucket `rank`, source-rank-major, original order inside a source; stage II will write
// global ids.  info (optional, 8 u64): [0] clean reads of the whole job [1] N reads [2] records [3..5] this rank's first clean id /
// N id / record [6] clean reads received [7] N reads received.
extern "C" int harc_amd_shard_exchange(harc_amd_ctx *c, uint64_t *info)
{
    if (!c) return HARC_AMD_EINVAL;
    if (!c->comm) { harc_set_error("harc_amd_shard_exchange: no communicator (harc_amd_comm_init)"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    HarcComm *cm = c->comm;
    const int world = cm->world, rank = cm->rank;
    const int W = c->W, W3 = c->W3;
    // results of an earlier run go (as every harc_amd_set_* does); the stages will read the shard from here on
    harc_drop_results(c);
    harc_reset_shard(c);
    const uint32_t N = c->N_own, NN = c->NN_own;
    const uint64_t *own2 = (const uint64_t *)c->own_reads.p, *own3 = (const uint64_t *)c->own_nreads3.p;

    // (1) sizes of every rank's slice -> global id offsets
    std::vector<uint64_t> all((size_t)world * 3);
    { const uint64_t mine[3] = { N, NN, c->nrec_own }; RC_TRY(cm->allgather_u64(c, mine, 3, all.data())); }
    uint64_t tot[3] = { 0, 0, 0 }, off[3] = { 0, 0, 0 };
    for (int r = 0; r < world; r++) for (int k = 0; k < 3; k++) { if (r < rank) off[k] += all[(size_t)r * 3 + k]; tot[k] += all[(size_t)r * 3 + k]; }
    if (tot[0] > 0xFFFFFFFFull || tot[1] > 0xFFFFFFFFull || tot[2] > 4294967290ull) {
        harc_set_error("Too many reads. HARC supports at most 4294967290 reads"); return HARC_AMD_EINVAL;      // preprocess.cpp:122-126
    }

    // (2) group by destination bucket.  Whatever goes wrong from here on, the send buffers go and the context keeps reading its own
    // slice (harc_reset_shard above): a failed exchange leaves the context usable
    PoolScope scope(c);
    uint64_t *s2 = nullptr, *s3 = nullptr; uint32_t *g2 = nullptr, *g3 = nullptr; unsigned long long *d_cnt = nullptr;
    RC_TRY(dalloc(c, &s2, (size_t)N * W + 1)); RC_TRY(dalloc(c, &g2, (size_t)N + 1));
    RC_TRY(dalloc(c, &s3, (size_t)NN * W3 + 1)); RC_TRY(dalloc(c, &g3, (size_t)NN + 1));
    RC_TRY(dalloc(c, &d_cnt, (size_t)2 * world));
    RC_TRY(shard_partition(c, own2, N, W, false, (uint32_t)world, (uint32_t)off[0], s2, g2, d_cnt));
    RC_TRY(shard_partition(c, own3, NN, W3, true, (uint32_t)world, (uint32_t)off[1], s3, g3, d_cnt + world));
    std::vector<uint64_t> row((size_t)2 * world), mat((size_t)2 * world * world);
    HIP_TRY(hipMemcpyAsync(row.data(), d_cnt, (size_t)2 * world * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (3) who sends how much to whom
    RC_TRY(cm->allgather_u64(c, row.data(), 2 * world, mat.data()));
    uint64_t rN = 0, rNN = 0;
    std::vector<size_t> so[4], sb[4], ro[4], rb[4];
    for (int a = 0; a < 4; a++) { so[a].resize(world); sb[a].resize(world); ro[a].resize(world); rb[a].resize(world); }
    {
        uint64_t sN = 0, sNN = 0;
        for (int p = 0; p < world; p++) {
            const uint64_t sc = row[p], sn = row[world + p];                                     // my reads for peer p
            const uint64_t rc = mat[(size_t)p * 2 * world + rank], rn = mat[(size_t)p * 2 * world + world + rank];   // peer p's reads for me
            so[0][p] = sN * W * 8;  sb[0][p] = sc * W * 8;  ro[0][p] = rN * W * 8;  rb[0][p] = rc * W * 8;
            so[1][p] = sN * 4;      sb[1][p] = sc * 4;      ro[1][p] = rN * 4;      rb[1][p] = rc * 4;
            so[2][p] = sNN * W3 * 8; sb[2][p] = sn * W3 * 8; ro[2][p] = rNN * W3 * 8; rb[2][p] = rn * W3 * 8;
            so[3][p] = sNN * 4;     sb[3][p] = sn * 4;      ro[3][p] = rNN * 4;     rb[3][p] = rn * 4;
            sN += sc; sNN += sn; rN += rc; rNN += rn;
        }
        if (sN != N || sNN != NN) { harc_set_error("shard exchange: bucket counts do not add up"); return HARC_AMD_EINTERNAL; }
    }
    if (rN > 0xFFFFFFFFull || rNN > 0xFFFFFFFFull) { harc_set_error("shard exchange: a bucket holds more than 2^32 reads"); return HARC_AMD_EINVAL; }
    RC_TRY(harc_in_reserve(c, &c->x_reads, ((size_t)rN * W + 1) * 8)); RC_TRY(harc_in_reserve(c, &c->x_gid, ((size_t)rN + 1) * 4));
    RC_TRY(harc_in_reserve(c, &c->x_nreads3, ((size_t)rNN * W3 + 1) * 8)); RC_TRY(harc_in_reserve(c, &c->x_ngid, ((size_t)rNN + 1) * 4));
    // (4) ONE all-to-all(v): packed reads + ids (8W + 4 bytes per clean read, 8 W3 + 4 per read with N)
    {
        const void *sp[4] = { s2, g2, s3, g3 };
        void *rp[4] = { c->x_reads.p, c->x_gid.p, c->x_nreads3.p, c->x_ngid.p };
        const size_t *sop[4], *sbp[4], *rop[4], *rbp[4];
        for (int a = 0; a < 4; a++) { sop[a] = so[a].data(); sbp[a] = sb[a].data(); rop[a] = ro[a].data(); rbp[a] = rb[a].data(); }
        RC_TRY(cm->alltoallv(c, 4, sp, sop, sbp, rp, rop, rbp));
        RC_TRY(cm->wait(c, "all-to-all of the reads"));              // the chunks have arrived (or the peers timed out)
    }
    scope.release_now();
    c->d_reads = (uint64_t *)c->x_reads.p; c->N = (uint32_t)rN;
    c->d_nreads3 = (uint64_t *)c->x_nreads3.p; c->NN = (uint32_t)rNN;
    c->d_gid = (uint32_t *)c->x_gid.p; c->d_ngid = (uint32_t *)c->x_ngid.p;
    c->C.n_clean = c->N; c->C.n_N = c->NN;
    c->shard_info[0] = tot[0]; c->shard_info[1] = tot[1]; c->shard_info[2] = tot[2];
    c->shard_info[3] = off[0]; c->shard_info[4] = off[1]; c->shard_info[5] = off[2]; c->shard_info[6] = rN; c->shard_info[7] = rNN;
    if (info) memcpy(info, c->shard_info, sizeof c->shard_info);
    return HARC_AMD_OK;
}

// The stages read the context's own inputs again (what harc_amd_set_* installed); results of the last run go.  The communicator stays.
extern "C" int harc_amd_shard_reset(harc_amd_ctx *c)
{
    if (!c) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    harc_drop_results(c);
    harc_reset_shard(c);
    return HARC_AMD_OK;
}

// Design (R) of the multi-GPU split: REPLICATE the reads, partition the CHAINS.  Every rank holds its slice of the job
// (as before an exchange); one all-gather puts the reads of the WHOLE job on every GPU, in global id order (rank-major = the order of a
// concatenated input_clean.dna / input_N.dna).  harc_amd_reorder then builds the full index on every GPU and walks only the chains it
// owns, with one all-gather of the walked steps per super-round (stage1.hip); what comes out of harc_amd_reorder / harc_amd_encode is, on
// every rank, byte for byte what ONE GPU produces from the concatenated input -- the archive keeps the single-GPU compression ratio,
// which the minimizer-bucket shard of harc_amd_shard_exchange does not (2.4-4.4 times the consensus bases).  The price is
// memory (every GPU holds everything) and the replicated parts (index build, arbitration, stage II).
// info as harc_amd_shard_exchange: [0..2] the whole job's clean reads / reads with N / records, [3..5] this rank's first, [6..7] = [0..1].
extern "C" int harc_amd_replicate_exchange(harc_amd_ctx *c, uint64_t *info)
{
    if (!c) return HARC_AMD_EINVAL;
    if (!c->comm) { harc_set_error("harc_amd_replicate_exchange: no communicator (harc_amd_comm_init)"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    HarcComm *cm = c->comm;
    const int world = cm->world, rank = cm->rank;
    if (world > 64) { harc_set_error("harc_amd_replicate_exchange: design (R) runs on at most 64 ranks (world %d)", world); return HARC_AMD_EINVAL; }     // before anything moves
    const int W = c->W, W3 = c->W3;
    harc_drop_results(c);
    harc_reset_shard(c);
    const uint32_t N = c->N_own, NN = c->NN_own;
    std::vector<uint64_t> all((size_t)world * 3);
    { const uint64_t mine[3] = { N, NN, c->nrec_own }; RC_TRY(cm->allgather_u64(c, mine, 3, all.data())); }
    uint64_t tot[3] = { 0, 0, 0 }, off[3] = { 0, 0, 0 };
    for (int r = 0; r < world; r++) for (int k = 0; k < 3; k++) { if (r < rank) off[k] += all[(size_t)r * 3 + k]; tot[k] += all[(size_t)r * 3 + k]; }
    if (tot[0] > 0xFFFFFFFFull || tot[1] > 0xFFFFFFFFull || tot[2] > 4294967290ull) {
        harc_set_error("Too many reads. HARC supports at most 4294967290 reads"); return HARC_AMD_EINVAL;      // preprocess.cpp:122-126
    }
    RC_TRY(harc_in_reserve(c, &c->x_reads, ((size_t)tot[0] * W + 1) * 8));
    RC_TRY(harc_in_reserve(c, &c->x_nreads3, ((size_t)tot[1] * W3 + 1) * 8));
    {   // the all-gather as ONE all-to-all(v) in which every peer gets the whole slice: chunks of (8W, 8 W3) bytes per read
        std::vector<size_t> so[2], sb[2], ro[2], rb[2];
        for (int a = 0; a < 2; a++) { so[a].assign(world, 0); sb[a].resize(world); ro[a].resize(world); rb[a].resize(world); }
        uint64_t accN = 0, accNN = 0;
        for (int p = 0; p < world; p++) {
            sb[0][p] = (size_t)N * W * 8; sb[1][p] = (size_t)NN * W3 * 8;
            ro[0][p] = (size_t)accN * W * 8; rb[0][p] = (size_t)all[(size_t)p * 3] * W * 8;
            ro[1][p] = (size_t)accNN * W3 * 8; rb[1][p] = (size_t)all[(size_t)p * 3 + 1] * W3 * 8;
            accN += all[(size_t)p * 3]; accNN += all[(size_t)p * 3 + 1];
        }
        const void *sp[2] = { c->own_reads.p, c->own_nreads3.p };
        void *rp[2] = { c->x_reads.p, c->x_nreads3.p };
        const size_t *sop[2] = { so[0].data(), so[1].data() }, *sbp[2] = { sb[0].data(), sb[1].data() }, *rop[2] = { ro[0].data(), ro[1].data() }, *rbp[2] = { rb[0].data(), rb[1].data() };
        RC_TRY(cm->alltoallv(c, 2, sp, sop, sbp, rp, rop, rbp));
        RC_TRY(cm->wait(c, "all-gather of the reads"));
    }
    c->d_reads = (uint64_t *)c->x_reads.p; c->N = (uint32_t)tot[0];
    c->d_nreads3 = (uint64_t *)c->x_nreads3.p; c->NN = (uint32_t)tot[1];
    c->d_gid = c->d_ngid = nullptr;                              // the ids of the replicated store ARE the global ids
    c->replicated = true;                                        // world 1 included: the per-round all-gather then runs over one rank (tests)
    c->C.n_clean = c->N; c->C.n_N = c->NN;
    c->shard_info[0] = tot[0]; c->shard_info[1] = tot[1]; c->shard_info[2] = tot[2];
    c->shard_info[3] = off[0]; c->shard_info[4] = off[1]; c->shard_info[5] = off[2]; c->shard_info[6] = tot[0]; c->shard_info[7] = tot[1];
    if (info) memcpy(info, c->shard_info, sizeof c->shard_info);
    return HARC_AMD_OK;
}
