// verify.hip -- decode side on the GPU, used as the size-independent round-trip check.
//
// decoder.cpp:90-169 restated data-parallel: the start column of read i in its shard's consensus stream is (sum of the pos bytes
// up to i) - readlen, its noise entries sit between the (i-1)-th and i-th '\n' of read_noise, the mismatch positions are the
// running sum of its read_noisepos bytes (decoder.cpp:100-108), the read is reverse-complemented when its read_rev bit is set
// (:109-129).  Instead of writing 100 B per read back to the host, every decoded read is hashed and the hashes are summed and
// xored: an order-independent signature of the decoded multiset that is compared with the same signature of the input reads.
#include "devutil.h"
#include <string>
#include <stdlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

__device__ __forceinline__ uint64_t read_hash_step(uint64_t h, int code) { return (h ^ (uint64_t)code) * 1099511628211ULL; }   // FNV-1a over codes A0 C1 G2 T3 N4
#define READ_HASH_INIT 1469598103934665603ULL

__device__ __forceinline__ void sig_accumulate(uint64_t h, bool valid, unsigned long long *sig)
{
    // sig[0] count, sig[1] sum, sig[2] xor ; wave-level reduction first
    unsigned long long s = valid ? h : 0, x = valid ? h : 0, n = valid ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += shfl_u64_any(s, o); x ^= shfl_u64_any(x, o); n += shfl_u64_any(n, o);
    }
    if ((threadIdx.x & 63) == 0 && n) { atomicAdd(&sig[0], n); atomicAdd(&sig[1], s); atomicXor(&sig[2], x); }
}

__global__ void k_sig_ascii(const char *ascii, uint32_t n, uint32_t stride, int L, unsigned long long *sig)
{
    const uint32_t i = harc_gid32();
    uint64_t h = READ_HASH_INIT;
    if (i < n) {
        const char *s = ascii + (size_t)i * stride;
        for (int j = 0; j < L; j++) { const char ch = s[j]; h = read_hash_step(h, ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4); }
        h = mix64(h);
    }
    sig_accumulate(h, i < n, sig);
}
// 2-bit packed stream (A0 C1 G2 T3, 4 per byte, encoder.cpp:540-541) + ASCII tail -> one code per byte
__global__ void k_unpack_seq(const uint8_t *packed, uint64_t nb, const uint8_t *tail, uint64_t ntail, uint8_t *out)
{
    const uint64_t i = harc_gid();
    if (i < 4 * nb) out[i] = (packed[i >> 2] >> (2 * (i & 3))) & 3;
    else if (i < 4 * nb + ntail) { const uint8_t ch = tail[i - 4 * nb]; out[i] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }
}
__global__ void k_pos_to_u64(const uint8_t *pos, uint32_t n, uint64_t *out)
{
    const uint32_t i = harc_gid32();
    if (i < n) out[i] = pos[i];
}
__global__ void k_nl_flags(const uint8_t *noise, uint64_t n, uint32_t *flag)
{
    const uint64_t i = harc_gid();
    if (i < n) flag[i] = noise[i] == '\n' ? 1u : 0u;
}
__global__ void k_nl_positions(const uint8_t *noise, const uint32_t *rank, uint64_t n, uint64_t *nlpos)
{
    const uint64_t i = harc_gid();
    if (i < n && noise[i] == '\n') nlpos[rank[i]] = i;
}
// one thread per read of a shard: decode, hash
__global__ void k_decode_sig(const uint8_t *seqb, uint64_t seqlen, const uint64_t *possum, const uint8_t *noise, const uint8_t *noisepos,
                             const uint64_t *nlpos, const uint8_t *revb, uint64_t nrevb, const uint8_t *revtail, uint32_t n, int L,
                             unsigned long long *sig, unsigned int *err)
{
    const uint32_t i = harc_gid32();
    uint64_t h = READ_HASH_INIT;
    bool ok = i < n;
    if (ok) {
        const uint64_t start = possum[i] - (uint64_t)L;                       // decoder.cpp:93-98
        if (possum[i] < (uint64_t)L || start + L > seqlen) { atomicAdd(err, 1u); ok = false; }
        else {
            uint8_t buf[256];
            for (int j = 0; j < L; j++) buf[j] = seqb[start + j];
            const uint64_t n0 = i ? nlpos[i - 1] + 1 : 0, n1 = nlpos[i];
            uint64_t np = n0 - i;                                             // noisepos bytes consumed by earlier reads
            int p = 0;
            for (uint64_t k = n0; k < n1; k++) {                              // decoder.cpp:100-108
                p += noisepos[np++];
                const int ref = buf[p] & 3, code = noise[k] - '0';
                // dec_noise (decoder.cpp setglobalarrays): A:{C,G,T,N} C:{A,G,T,N} G:{T,A,C,N} T:{G,C,A,N}
                const unsigned tab = ref == 0 ? 0x4321u : ref == 1 ? 0x4320u : ref == 2 ? 0x4103u : 0x4012u;
                if (p < L) buf[p] = (uint8_t)((tab >> (4 * code)) & 0xF); else atomicAdd(err, 1u);
            }
            const bool rev = i < 8 * nrevb ? ((revb[i >> 3] >> (i & 7)) & 1) : (revtail[i - 8 * nrevb] == 'r');
            if (!rev) for (int j = 0; j < L; j++) h = read_hash_step(h, buf[j]);
            else for (int j = L - 1; j >= 0; j--) h = read_hash_step(h, buf[j] == 4 ? 4 : 3 - buf[j]);   // decoder.cpp:109-129
            h = mix64(h);
        }
    }
    sig_accumulate(h, ok, sig);
}
// unaligned singletons: 2-bit packed, L bases each, back to back (encoder.cpp:484-491)
__global__ void k_sig_codes(const uint8_t *codes, uint32_t n, int L, unsigned long long *sig)
{
    const uint32_t i = harc_gid32();
    uint64_t h = READ_HASH_INIT;
    if (i < n) { for (int j = 0; j < L; j++) h = read_hash_step(h, codes[(size_t)i * L + j]); h = mix64(h); }
    sig_accumulate(h, i < n, sig);
}

#define G256(n) harc_grid256((uint64_t)(n)), dim3(256), 0, c->stream

static bool get_out(harc_amd_ctx *c, int id, int shard, const uint8_t **p, size_t *n)
{
    auto it = c->out.find(std::make_pair(id, shard));
    if (it == c->out.end()) return false;
    if (it->second.ptr) { *p = it->second.ptr; *n = it->second.len; } else { *p = it->second.own.data(); *n = it->second.own.size(); }
    return true;
}
static int up(harc_amd_ctx *c, const uint8_t *h, size_t n, uint8_t **d)
{
    RC_TRY(dalloc(c, d, n + 16));
    if (n) HIP_TRY(hipMemcpyAsync(*d, h, n, hipMemcpyHostToDevice, c->stream));
    return HARC_AMD_OK;
}

extern "C" int harc_amd_reads_signature_device(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *sig3)
{
    if (!c || !sig3 || (n && !d_ascii) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    const harc_mark_t mk = harc_pool_mark(c);
    unsigned long long *d_sig = nullptr; RC_TRY(dalloc(c, &d_sig, 4));
    HIP_TRY(hipMemsetAsync(d_sig, 0, 32, c->stream));
    if (n) hipLaunchKernelGGL(k_sig_ascii, G256(n), d_ascii, n, stride, c->P.readlen, d_sig);
    HIP_TRY(hipMemcpyAsync(sig3, d_sig, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    harc_pool_release(c, mk);
    return HARC_AMD_OK;
}

extern "C" int harc_amd_decode_signature(harc_amd_ctx *c, uint64_t *sig3)
{
    if (!c || !sig3) return HARC_AMD_EINVAL;
    if (!c->have_s2) { harc_set_error("harc_amd_decode_signature: encode first"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    const int L = c->P.readlen;
    const harc_mark_t mk0 = harc_pool_mark(c);
    unsigned long long *d_sig = nullptr; unsigned int *d_err = nullptr;
    RC_TRY(dalloc(c, &d_sig, 4)); RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_sig, 0, 32, c->stream)); HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    for (int e = 0; e < c->P.num_thr; e++) {
        const uint8_t *seq, *seqt, *pos, *noise, *npz, *rev, *revt; size_t nseq, nseqt, npos, nnoise, nnp, nrev, nrevt;
        if (!get_out(c, HARC_AMD_S2_SEQ, e, &seq, &nseq) || !get_out(c, HARC_AMD_S2_SEQ_TAIL, e, &seqt, &nseqt) || !get_out(c, HARC_AMD_S2_POS, e, &pos, &npos) ||
            !get_out(c, HARC_AMD_S2_NOISE, e, &noise, &nnoise) || !get_out(c, HARC_AMD_S2_NOISEPOS, e, &npz, &nnp) ||
            !get_out(c, HARC_AMD_S2_REV, e, &rev, &nrev) || !get_out(c, HARC_AMD_S2_REV_TAIL, e, &revt, &nrevt)) { harc_set_error("shard %d streams missing", e); return HARC_AMD_ESTATE; }
        if (npos == 0) continue;
        if (npos > 0xFFFFFFFFull || 8 * nrev + nrevt != npos) { harc_set_error("shard %d: rev stream does not match pos stream", e); return HARC_AMD_EIO; }
        const harc_mark_t mk = harc_pool_mark(c);
        const uint32_t n = (uint32_t)npos;
        uint8_t *d_seq, *d_seqt, *d_pos, *d_noise, *d_np, *d_rev, *d_revt, *seqb; uint64_t *p64, *possum, *nlpos; uint32_t *fl, *rk;
        RC_TRY(up(c, seq, nseq, &d_seq)); RC_TRY(up(c, seqt, nseqt, &d_seqt)); RC_TRY(up(c, pos, npos, &d_pos)); RC_TRY(up(c, noise, nnoise, &d_noise));
        RC_TRY(up(c, npz, nnp, &d_np)); RC_TRY(up(c, rev, nrev, &d_rev)); RC_TRY(up(c, revt, nrevt, &d_revt));
        const uint64_t seqlen = 4 * (uint64_t)nseq + nseqt;
        RC_TRY(dalloc(c, &seqb, (size_t)seqlen + 16)); RC_TRY(dalloc(c, &p64, (size_t)n + 1)); RC_TRY(dalloc(c, &possum, (size_t)n + 1));
        RC_TRY(dalloc(c, &nlpos, (size_t)n + 1)); RC_TRY(dalloc(c, &fl, nnoise + 1)); RC_TRY(dalloc(c, &rk, nnoise + 1));
        if (seqlen) hipLaunchKernelGGL(k_unpack_seq, G256(seqlen), d_seq, (uint64_t)nseq, d_seqt, (uint64_t)nseqt, seqb);
        hipLaunchKernelGGL(k_pos_to_u64, G256(n), d_pos, n, p64);
        RC_TRY(prim_incl_scan_u64(c, p64, possum, n));
        if (nnoise) {
            hipLaunchKernelGGL(k_nl_flags, G256(nnoise), d_noise, (uint64_t)nnoise, fl);
            RC_TRY(prim_excl_scan_u32(c, fl, rk, nnoise));
            hipLaunchKernelGGL(k_nl_positions, G256(nnoise), d_noise, rk, (uint64_t)nnoise, nlpos);
        }
        hipLaunchKernelGGL(k_decode_sig, G256(n), seqb, seqlen, possum, d_noise, d_np, nlpos, d_rev, (uint64_t)nrev, d_revt, n, L, d_sig, d_err);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        harc_pool_release(c, mk);
    }
    {   // unaligned singletons and unaligned N reads (decoder.cpp:148-169)
        const uint8_t *sg, *sgt, *nt; size_t nsg, nsgt, nnt;
        if (!get_out(c, HARC_AMD_S2_SINGLETON, 0, &sg, &nsg) || !get_out(c, HARC_AMD_S2_SINGLETON_TAIL, 0, &sgt, &nsgt) || !get_out(c, HARC_AMD_S2_INPUT_N, 0, &nt, &nnt)) {
            harc_set_error("singleton streams missing"); return HARC_AMD_ESTATE;
        }
        const uint64_t nb = 4 * (uint64_t)nsg + nsgt;
        uint8_t *d_sg, *d_sgt, *codes, *d_nt;
        RC_TRY(up(c, sg, nsg, &d_sg)); RC_TRY(up(c, sgt, nsgt, &d_sgt)); RC_TRY(up(c, nt, nnt, &d_nt)); RC_TRY(dalloc(c, &codes, (size_t)nb + 16));
        if (nb) hipLaunchKernelGGL(k_unpack_seq, G256(nb), d_sg, (uint64_t)nsg, d_sgt, (uint64_t)nsgt, codes);
        const uint32_t ns = (uint32_t)(nb / L), nn = (uint32_t)(nnt / (L + 1));
        if (ns) hipLaunchKernelGGL(k_sig_codes, G256(ns), codes, ns, L, d_sig);
        if (nn) hipLaunchKernelGGL(k_sig_ascii, G256(nn), (const char *)d_nt, nn, (uint32_t)L + 1, L, d_sig);
    }
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(sig3, d_sig, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    harc_pool_release(c, mk0);
    if (err) { harc_set_error("decode: %u reads with inconsistent pos/noise streams", err); return HARC_AMD_EIO; }
    return HARC_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ decoder.out drop-in (non -p)
// Same decode as k_decode_sig, but the read is written as a text line into tmp[i] and flagged when it contains N
// (decoder.cpp:110-129 routes those to a separate file that is appended after the singletons, :141-169).
__global__ void k_decode_text(const uint8_t *seqb, uint64_t seqlen, const uint64_t *possum, const uint8_t *noise, const uint8_t *noisepos,
                              const uint64_t *nlpos, const uint8_t *revb, uint64_t nrevb, const uint8_t *revtail, uint32_t n, int L,
                              char *tmp, uint32_t *isN, unsigned int *err)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint64_t start = possum[i] - (uint64_t)L;
    char *o = tmp + (size_t)i * (L + 1);
    if (possum[i] < (uint64_t)L || start + L > seqlen) { atomicAdd(err, 1u); isN[i] = 0; for (int j = 0; j < L; j++) o[j] = 'A'; o[L] = '\n'; return; }
    uint8_t buf[256];
    for (int j = 0; j < L; j++) buf[j] = seqb[start + j];
    const uint64_t n0 = i ? nlpos[i - 1] + 1 : 0, n1 = nlpos[i];
    uint64_t np = n0 - i;
    int p = 0; bool hasN = false;
    for (uint64_t k = n0; k < n1; k++) {
        p += noisepos[np++];
        const int ref = buf[p < L ? p : 0] & 3, code = noise[k] - '0';
        const unsigned tab = ref == 0 ? 0x4321u : ref == 1 ? 0x4320u : ref == 2 ? 0x4103u : 0x4012u;
        if (p < L) { buf[p] = (uint8_t)((tab >> (4 * code)) & 0xF); hasN |= buf[p] == 4; } else atomicAdd(err, 1u);
    }
    const bool rev = i < 8 * nrevb ? ((revb[i >> 3] >> (i & 7)) & 1) : (revtail[i - 8 * nrevb] == 'r');
    if (!rev) for (int j = 0; j < L; j++) o[j] = "ACGTN"[buf[j]];
    else for (int j = 0; j < L; j++) { const int b = buf[L - 1 - j]; o[j] = "ACGTN"[b == 4 ? 4 : 3 - b]; }
    o[L] = '\n';
    isN[i] = hasN ? 1u : 0u;
}
// stable split of the lines: reads without N to outA[rankA], reads with N to outN[i - rankA]
__global__ void k_split_lines(const char *tmp, const uint32_t *isN, const uint32_t *rankN, uint32_t n, int L, char *outA, char *outN)
{
    const uint64_t gid = harc_gid();
    const uint64_t LL = (uint64_t)L + 1;
    if (gid >= (uint64_t)n * LL) return;
    const uint32_t i = (uint32_t)(gid / LL); const uint32_t j = (uint32_t)(gid % LL);
    const char ch = tmp[gid];
    if (isN[i]) outN[(uint64_t)rankN[i] * LL + j] = ch; else outA[(uint64_t)(i - rankN[i]) * LL + j] = ch;
}
__global__ void k_codes_to_lines(const uint8_t *codes, uint32_t n, int L, char *out)
{
    const uint64_t gid = harc_gid();
    const uint64_t LL = (uint64_t)L + 1;
    if (gid >= (uint64_t)n * LL) return;
    const uint32_t i = (uint32_t)(gid / LL); const uint32_t j = (uint32_t)(gid % LL);
    out[gid] = j == (uint32_t)L ? '\n' : "ACGT"[codes[(uint64_t)i * L + j] & 3];
}

static bool slurp_file(const std::string &path, std::vector<uint8_t> &out)
{
    out.clear();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// HBM -> file at the rate of the host's memory system (round 6: decoder.out's replacement spent 4.4 of its 4.7 s on 100 M reads in D2H copies into pageable
// vectors and one thread's fwrite): the output file is sized and mapped first (its length is known from the stream files' sizes), the calling thread sends
// pieces of device memory through a ring of pinned slices (hipMemcpyAsync on the context's stream), writer threads copy every slice that has arrived into the
// mapping.  write() calls to ONE tmpfs file serialise on its inode (tools/micro/feed_rate.cpp: 6 GB/s with 1 or 16 threads); page faults of a shared mapping do not.
struct FileDrain {
    struct Job { int sl; size_t len; uint64_t off; };
    harc_amd_ctx *c; int fd = -1; char *map = nullptr; size_t fsize = 0;
    size_t SL = 0; int NS = 0;
    std::vector<hipEvent_t> ev;
    std::mutex mu; std::condition_variable cv_free, cv_job, cv_idle;
    std::deque<int> free_slices; std::deque<Job> jobs; int busy = 0; bool stop = false;
    std::vector<std::thread> th;
    // the file's blocks are ALLOCATED ahead of the writers by a thread of its own (posix_fallocate, 64 MB at a time): a store into a mapping of a sparse file on a full
    // file system is a SIGBUS, not an error code -- this way "no space left" is an error of the call, as it was with fwrite
    std::thread alloc_th; std::condition_variable cv_alloc; uint64_t alloc_upto = 0; int alloc_err = 0;
    std::string fname;
    explicit FileDrain(harc_amd_ctx *c_) : c(c_) {}
    ~FileDrain() { (void)finish(); }
    int start(const std::string &path, size_t bytes)
    {
        fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (fd < 0) { harc_set_error("cannot create %s", path.c_str()); return HARC_AMD_EIO; }
        fsize = bytes; fname = path;
        if (bytes) {
            if (ftruncate(fd, (off_t)bytes) != 0) { harc_set_error("cannot size %s to %zu bytes", path.c_str(), bytes); return HARC_AMD_EIO; }
            map = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { map = nullptr; harc_set_error("cannot map %s", path.c_str()); return HARC_AMD_EIO; }
        }
        SL = (size_t)64 << 20; NS = 16;
        int nthr = 16;
        if (const char *e = getenv("HARC_AMD_FEED_THREADS")) { const int x = atoi(e); if (x >= 1 && x <= 64) nthr = x; }
        if (const char *e = getenv("HARC_AMD_FEED_SLICE")) { const long long x = atoll(e); if (x >= 16 && x <= ((long long)1 << 30)) SL = (size_t)x; }      // tests: slices of a few reads
        if (c->feed_ring_bytes < SL * (size_t)NS) {
            if (c->feed_ring) { (void)hipHostFree(c->feed_ring); c->feed_ring = nullptr; c->feed_ring_bytes = 0; }
            if (hipHostMalloc((void **)&c->feed_ring, SL * (size_t)NS) != hipSuccess) { harc_set_error("hipHostMalloc of the output ring (%zu bytes) failed", SL * (size_t)NS); return HARC_AMD_ENOMEM; }
            c->feed_ring_bytes = SL * (size_t)NS;
        }
        ev.assign(NS, nullptr);
        for (int k = 0; k < NS; k++) { if (hipEventCreate(&ev[k]) != hipSuccess) { harc_set_error("hipEventCreate failed"); return HARC_AMD_ENODEVICE; } free_slices.push_back(k); }
        alloc_th = std::thread([this] {
            const uint64_t STEP = (uint64_t)64 << 20;
            for (uint64_t a = 0; a < (uint64_t)fsize; a += STEP) {
                const uint64_t len = (uint64_t)fsize - a < STEP ? (uint64_t)fsize - a : STEP;
                const int e = posix_fallocate(fd, (off_t)a, (off_t)len);
                std::lock_guard<std::mutex> lk(mu);
                if (e == EOPNOTSUPP || e == EINVAL) { alloc_upto = (uint64_t)fsize; break; }      // a file system without preallocation: as before this round
                if (e) { alloc_err = e; break; }
                alloc_upto = a + len;
                cv_alloc.notify_all();
                if (stop) break;
            }
            cv_alloc.notify_all();
        });
        const int dev = c->P.device;
        for (int t = 0; t < nthr; t++) th.emplace_back([this, dev] {
            (void)hipSetDevice(dev);
            for (;;) {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
                    if (jobs.empty()) return;
                    j = jobs.front(); jobs.pop_front(); busy++;
                }
                (void)hipEventSynchronize(ev[j.sl]);              // the slice has arrived
                bool space;
                { std::unique_lock<std::mutex> lk(mu); cv_alloc.wait(lk, [&] { return alloc_err != 0 || alloc_upto >= j.off + j.len; }); space = alloc_err == 0; }
                if (space) memcpy(map + j.off, c->feed_ring + (size_t)j.sl * SL, j.len);
                { std::lock_guard<std::mutex> lk(mu); busy--; free_slices.push_back(j.sl); }
                cv_free.notify_one(); cv_idle.notify_all();
            }
        });
        return HARC_AMD_OK;
    }
    // n bytes of device memory -> bytes [off, off + n) of the file.  The copies are on the context's stream: what is enqueued behind them may reuse d_src
    int put(const void *d_src, size_t n, uint64_t off)
    {
        if (off + n > fsize) { harc_set_error("output file: %zu bytes at %llu do not fit its %zu bytes", n, (unsigned long long)off, fsize); return HARC_AMD_EINTERNAL; }
        for (size_t a = 0; a < n; a += SL) {
            const size_t len = n - a < SL ? n - a : SL;
            int sl;
            { std::unique_lock<std::mutex> lk(mu); cv_free.wait(lk, [&] { return !free_slices.empty(); }); sl = free_slices.front(); free_slices.pop_front(); }
            if (hipMemcpyAsync(c->feed_ring + (size_t)sl * SL, (const char *)d_src + a, len, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipEventRecord(ev[sl], c->stream) != hipSuccess) {
                harc_set_error("device -> host copy of the output failed"); return HARC_AMD_ENODEVICE; }
            { std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{ sl, len, off + a }); }
            cv_job.notify_one();
        }
        return HARC_AMD_OK;
    }
    int put_host(const void *h, size_t n, uint64_t off)
    {
        if (off + n > fsize) { harc_set_error("output file: %zu bytes at %llu do not fit its %zu bytes", n, (unsigned long long)off, fsize); return HARC_AMD_EINTERNAL; }
        if (n) {
            { std::unique_lock<std::mutex> lk(mu); cv_alloc.wait(lk, [&] { return alloc_err != 0 || alloc_upto >= off + n; }); if (alloc_err) return HARC_AMD_OK; }      // (finish() reports it)
            memcpy(map + off, h, n);
        }
        return HARC_AMD_OK;
    }
    int finish()
    {
        if (!th.empty()) {
            { std::unique_lock<std::mutex> lk(mu); cv_idle.wait(lk, [&] { return jobs.empty() && busy == 0; }); stop = true; }
            cv_job.notify_all();
            for (auto &t : th) t.join();
            th.clear();
        }
        if (alloc_th.joinable()) alloc_th.join();
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        ev.clear();
        if (map) { munmap(map, fsize); map = nullptr; }
        if (fd >= 0) { close(fd); fd = -1; }
        if (alloc_err) { harc_set_error("cannot allocate %zu bytes for %s: %s", fsize, fname.c_str(), strerror(alloc_err)); const int e = alloc_err; alloc_err = 0; (void)e; return HARC_AMD_EIO; }
        return HARC_AMD_OK;
    }
};
static size_t file_size_or_zero(const std::string &path) { struct stat st; return stat(path.c_str(), &st) == 0 ? (size_t)st.st_size : 0; }

// decoder.out <basedir> <num_thr> <num_thr_e>  (src/decoder.cpp:44-172, harc:188): writes output/output.dna
extern "C" int harc_amd_decoder_files(const harc_amd_params *params, const char *basedir, int32_t num_thr_e)
{
    if (!params || !basedir || num_thr_e < 1) return HARC_AMD_EINVAL;
    const std::string od = std::string(basedir) + "/output/";
    std::vector<uint8_t> meta;
    if (!slurp_file(od + "read_meta.txt", meta)) { harc_set_error("cannot read %sread_meta.txt", od.c_str()); return HARC_AMD_EIO; }
    meta.push_back(0);
    const int L = atoi((const char *)meta.data());                              // getDataParams, decoder.cpp:324-333
    harc_amd_params P = *params;
    if (harc_amd_default_params(L, &P) != HARC_AMD_OK) return HARC_AMD_EINVAL;
    P.device = params->device; P.num_thr = num_thr_e;
    harc_amd_ctx *c = nullptr;
    RC_TRY(harc_amd_create(&P, &c));
    struct Guard { harc_amd_ctx *c; ~Guard() { harc_amd_destroy(c); } } guard{ c };
    // output.dna holds one line per read: its length is known before a byte is decoded -- a read per byte of read_pos.txt.<e> (decoder.cpp:96), the singletons
    // (4 bases per byte + tail, :148-158), input_N.dna as it is (:166-168)
    const size_t LLo = (size_t)L + 1;
    size_t total_out = file_size_or_zero(od + "input_N.dna");
    for (int e = 0; e < num_thr_e; e++) total_out += file_size_or_zero(od + "read_pos.txt." + std::to_string(e)) * LLo;
    total_out += ((4 * file_size_or_zero(od + "read_singleton.txt") + file_size_or_zero(od + "read_singleton.txt.tail")) / (size_t)L) * LLo;
    // the N reads of every shard come behind the singletons (decoder.cpp:159-165): they wait in device memory of their own (declared in front of the
    // drain: it goes after the drain has written what it still holds)
    struct NParts { harc_amd_ctx *c; std::vector<std::pair<char *, size_t>> v; ~NParts() { for (auto &x : v) if (x.first) harc_raw_free(c, x.first); } } nparts{ c, {} };
    auto wall = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double td0 = wall(); double t_slurp = 0, t_put = 0;
    FileDrain drain(c);
    RC_TRY(drain.start(od + "output.dna", total_out));
    const double td1 = wall();
    uint64_t out_at = 0;
    unsigned int *d_err = nullptr; RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    for (int e = 0; e < num_thr_e; e++) {
        const std::string sfx = "." + std::to_string(e);
        std::vector<uint8_t> seq, seqt, pos, noise, npz, rev, revt;
        const double ts0 = wall();
        if (!slurp_file(od + "read_seq.txt" + sfx, seq) || !slurp_file(od + "read_seq.txt" + sfx + ".tail", seqt) || !slurp_file(od + "read_pos.txt" + sfx, pos) ||
            !slurp_file(od + "read_noise.txt" + sfx, noise) || !slurp_file(od + "read_noisepos.txt" + sfx, npz) ||
            !slurp_file(od + "read_rev.txt" + sfx, rev) || !slurp_file(od + "read_rev.txt" + sfx + ".tail", revt)) { harc_set_error("shard %d: stream files missing", e); return HARC_AMD_EIO; }
        t_slurp += wall() - ts0;
        if (pos.empty()) continue;
        if (pos.size() > 0xFFFFFFFFull || 8 * rev.size() + revt.size() != pos.size()) { harc_set_error("shard %d: rev stream does not match pos stream", e); return HARC_AMD_EIO; }
        const harc_mark_t mk = harc_pool_mark(c);
        const uint32_t n = (uint32_t)pos.size();
        uint8_t *d_seq, *d_seqt, *d_pos, *d_noise, *d_np, *d_rev, *d_revt, *seqb; uint64_t *p64, *possum, *nlpos; uint32_t *fl, *rk, *isN, *rkN; char *tmp, *outA, *outN;
        RC_TRY(up(c, seq.data(), seq.size(), &d_seq)); RC_TRY(up(c, seqt.data(), seqt.size(), &d_seqt)); RC_TRY(up(c, pos.data(), pos.size(), &d_pos));
        RC_TRY(up(c, noise.data(), noise.size(), &d_noise)); RC_TRY(up(c, npz.data(), npz.size(), &d_np)); RC_TRY(up(c, rev.data(), rev.size(), &d_rev)); RC_TRY(up(c, revt.data(), revt.size(), &d_revt));
        const uint64_t seqlen = 4 * (uint64_t)seq.size() + seqt.size();
        const size_t nnoise = noise.size(), LL = (size_t)L + 1;
        RC_TRY(dalloc(c, &seqb, (size_t)seqlen + 16)); RC_TRY(dalloc(c, &p64, (size_t)n + 1)); RC_TRY(dalloc(c, &possum, (size_t)n + 1)); RC_TRY(dalloc(c, &nlpos, (size_t)n + 1));
        RC_TRY(dalloc(c, &fl, nnoise + 1)); RC_TRY(dalloc(c, &rk, nnoise + 1)); RC_TRY(dalloc(c, &isN, (size_t)n + 1)); RC_TRY(dalloc(c, &rkN, (size_t)n + 1));
        RC_TRY(dalloc(c, &tmp, (size_t)n * LL + 16)); RC_TRY(dalloc(c, &outA, (size_t)n * LL + 16)); RC_TRY(dalloc(c, &outN, (size_t)n * LL + 16));
        if (seqlen) hipLaunchKernelGGL(k_unpack_seq, G256(seqlen), d_seq, (uint64_t)seq.size(), d_seqt, (uint64_t)seqt.size(), seqb);
        hipLaunchKernelGGL(k_pos_to_u64, G256(n), d_pos, n, p64);
        RC_TRY(prim_incl_scan_u64(c, p64, possum, n));
        if (nnoise) {
            hipLaunchKernelGGL(k_nl_flags, G256(nnoise), d_noise, (uint64_t)nnoise, fl);
            RC_TRY(prim_excl_scan_u32(c, fl, rk, nnoise));
            hipLaunchKernelGGL(k_nl_positions, G256(nnoise), d_noise, rk, (uint64_t)nnoise, nlpos);
        }
        HIP_TRY(hipMemsetAsync(isN, 0, ((size_t)n + 1) * 4, c->stream));
        hipLaunchKernelGGL(k_decode_text, G256(n), seqb, seqlen, possum, d_noise, d_np, nlpos, d_rev, (uint64_t)rev.size(), d_revt, n, L, tmp, isN, d_err);
        RC_TRY(prim_excl_scan_u32(c, isN, rkN, (size_t)n + 1));
        hipLaunchKernelGGL(k_split_lines, G256((uint64_t)n * LL), tmp, isN, rkN, n, L, outA, outN);
        uint32_t nN = 0;
        HIP_TRY(hipMemcpyAsync(&nN, rkN + n, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const size_t bytesA = (size_t)(n - nN) * LL, bytesN = (size_t)nN * LL;
        if (bytesN) {
            char *keep = nullptr;
            RC_TRY(harc_raw_alloc(c, (void **)&keep, bytesN + 16));
            nparts.v.emplace_back(keep, bytesN);
            HIP_TRY(hipMemcpyAsync(keep, outN, bytesN, hipMemcpyDeviceToDevice, c->stream));
        }
        { const double tp0 = wall(); if (bytesA) { RC_TRY(drain.put(outA, bytesA, out_at)); out_at += bytesA; } t_put += wall() - tp0; }
        harc_pool_release(c, mk);                                  // (the copies out of outA / outN are on the stream in front of whatever takes their place)
    }
    {   // singletons (decoder.cpp:148-158), then the N reads of every shard (:159-165), then input_N.dna (:166-168)
        std::vector<uint8_t> sg, sgt, nt;
        if (!slurp_file(od + "read_singleton.txt", sg) || !slurp_file(od + "read_singleton.txt.tail", sgt)) { harc_set_error("singleton files missing"); return HARC_AMD_EIO; }
        slurp_file(od + "input_N.dna", nt);
        const uint64_t nb = 4 * (uint64_t)sg.size() + sgt.size();
        const uint32_t ns = (uint32_t)(nb / L);
        if (ns) {
            const harc_mark_t mk = harc_pool_mark(c);
            uint8_t *d_sg, *d_sgt, *codes; char *lines;
            RC_TRY(up(c, sg.data(), sg.size(), &d_sg)); RC_TRY(up(c, sgt.data(), sgt.size(), &d_sgt)); RC_TRY(dalloc(c, &codes, (size_t)nb + 16));
            RC_TRY(dalloc(c, &lines, (size_t)ns * (L + 1) + 16));
            hipLaunchKernelGGL(k_unpack_seq, G256(nb), d_sg, (uint64_t)sg.size(), d_sgt, (uint64_t)sgt.size(), codes);
            hipLaunchKernelGGL(k_codes_to_lines, G256((uint64_t)ns * (L + 1)), codes, ns, L, lines);
            RC_TRY(drain.put(lines, (size_t)ns * (L + 1), out_at)); out_at += (size_t)ns * (L + 1);
            harc_pool_release(c, mk);
        }
        for (auto &p : nparts.v) { RC_TRY(drain.put(p.first, p.second, out_at)); out_at += p.second; }
        RC_TRY(drain.put_host(nt.data(), nt.size(), out_at)); out_at += nt.size();
    }
    if (out_at != total_out) { harc_set_error("decoder: %llu bytes decoded, the stream files announce %zu", (unsigned long long)out_at, total_out); return HARC_AMD_EIO; }
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double tf0 = wall();
    RC_TRY(drain.finish());
    if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[decoder] %.3f s: output mapped and ring ready %.3f, stream files read %.3f, waiting for ring slices %.3f, last slices written %.3f\n", wall() - td0, td1 - td0, t_slurp, t_put, wall() - tf0);
    if (err) { harc_set_error("decoder: %u reads with inconsistent pos/noise streams", err); return HARC_AMD_EIO; }
    printf("Decoding done\n");                                                 // decoder.cpp:170
    return HARC_AMD_OK;
}

// signature of the reads the context currently holds (2-bit clean reads + 3-bit N reads): the other side of the round-trip check
// when the inputs never existed as ASCII on this GPU (the shard received through the all-to-all)
__global__ void k_sig_packed2(const uint64_t *reads, uint32_t n, int L, int W, unsigned long long *sig)
{
    const uint32_t i = harc_gid32();
    uint64_t h = READ_HASH_INIT;
    if (i < n) {
        const uint64_t *r = reads + (size_t)i * W;
        for (int j = 0; j < L; j++) { const int pc = (int)((r[j >> 5] >> (2 * (j & 31))) & 3); h = read_hash_step(h, ((pc & 1) << 1) | (pc >> 1)); }
        h = mix64(h);
    }
    sig_accumulate(h, i < n, sig);
}
__global__ void k_sig_packed3(const uint64_t *reads, uint32_t n, int L, int W3, unsigned long long *sig)
{
    const uint32_t i = harc_gid32();
    uint64_t h = READ_HASH_INIT;
    if (i < n) {
        const uint64_t *r = reads + (size_t)i * W3;
        for (int j = 0; j < L; j++) {
            const int off = 3 * j, wi = off >> 6, sh = off & 63;
            uint64_t v = r[wi] >> sh;
            if (sh > 61 && wi + 1 < W3) v |= r[wi + 1] << (64 - sh);
            const int c3 = (int)(v & 7);
            h = read_hash_step(h, c3 == 0 ? 0 : c3 == 4 ? 1 : c3 == 2 ? 2 : c3 == 6 ? 3 : 4);
        }
        h = mix64(h);
    }
    sig_accumulate(h, i < n, sig);
}
extern "C" int harc_amd_input_signature(harc_amd_ctx *c, uint64_t *sig3)
{
    if (!c || !sig3) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    const harc_mark_t mk = harc_pool_mark(c);
    unsigned long long *d_sig = nullptr; RC_TRY(dalloc(c, &d_sig, 4));
    HIP_TRY(hipMemsetAsync(d_sig, 0, 32, c->stream));
    if (c->N && c->d_reads) hipLaunchKernelGGL(k_sig_packed2, G256(c->N), c->d_reads, c->N, c->P.readlen, c->W, d_sig);
    if (c->NN && c->d_nreads3) hipLaunchKernelGGL(k_sig_packed3, G256(c->NN), c->d_nreads3, c->NN, c->P.readlen, c->W3, d_sig);
    HIP_TRY(hipMemcpyAsync(sig3, d_sig, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    harc_pool_release(c, mk);
    return HARC_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ -p decode chain
// unpack_order.out + decoder_preserve.out + merge_N.out (harc:183-185) in one call: every read goes back to its line of the original
// FASTQ.  read_order.bin (packed, pack_order.cpp) tells the original clean-read index of every decoded clean read in stream order,
// read_order_N_pe.bin the index among the N reads of every decoded / left-over N read, read_order_N.bin the original line of every
// N read (merge_N.cpp:37-57).
__global__ void k_unpack_order(const uint32_t *packed, uint32_t ngroups, int numbits, uint32_t *out)      // unpack_order.cpp:34-62
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)ngroups * 32) return;
    const uint32_t g = (uint32_t)(gid >> 5); const int k = (int)(gid & 31);
    const uint32_t *w = packed + (size_t)g * numbits;
    const int bit = k * numbits, wi = bit >> 5, sh = bit & 31;
    uint64_t v = w[wi];
    if (sh + numbits > 32) v |= (uint64_t)w[wi + 1] << 32;
    out[gid] = (uint32_t)((v >> sh) & (numbits == 32 ? 0xFFFFFFFFull : ((1ull << numbits) - 1)));
}
// line i of src goes to line order[i] - lo of dst when lo <= order[i] < hi (one bin of restore_order, decoder_preserve.cpp:246-290);
// order[i] >= ndst is an inconsistent archive
__global__ void k_permute_lines(const char *src, const uint32_t *order, uint32_t n, int L, char *dst, uint32_t lo, uint32_t hi, uint32_t ndst, unsigned int *err)
{
    const uint64_t gid = harc_gid();
    const uint64_t LL = (uint64_t)L + 1;
    if (gid >= (uint64_t)n * LL) return;
    const uint32_t i = (uint32_t)(gid / LL);
    const uint32_t o = order[i];
    if (o >= ndst) { if (gid % LL == 0) atomicAdd(err, 1u); return; }
    if (o < lo || o >= hi) return;
    dst[(uint64_t)(o - lo) * LL + gid % LL] = src[gid];
}
__global__ void k_mark_N(const uint32_t *orderN, uint32_t nN, uint32_t total, uint32_t *flag, unsigned int *err)
{
    const uint32_t m = harc_gid32();
    if (m >= nN) return;
    if (orderN[m] >= total) { atomicAdd(err, 1u); return; }
    flag[orderN[m]] = 1u;
}
// output lines [p0, p0 + n): line p is the next read with N (flag) or the next clean read; cl / nl hold the clean reads from index clo on and
// the N reads from index nlo on (merge_N.cpp:37-57, one bin of it)
__global__ void k_merge_lines(const char *clean, const char *withN, const uint32_t *flag, const uint32_t *rankN, uint32_t p0, uint32_t n, uint32_t clo, uint32_t nlo, int L, char *out)
{
    const uint64_t gid = harc_gid();
    const uint64_t LL = (uint64_t)L + 1;
    if (gid >= (uint64_t)n * LL) return;
    const uint32_t p = p0 + (uint32_t)(gid / LL); const uint64_t j = gid % LL;
    out[gid] = flag[p] ? withN[(uint64_t)(rankN[p] - nlo) * LL + j] : clean[(uint64_t)(p - rankN[p] - clo) * LL + j];
}

extern "C" int harc_amd_decoder_preserve_files(const harc_amd_params *params, const char *basedir, int32_t num_thr_e)
{
    if (!params || !basedir || num_thr_e < 1) return HARC_AMD_EINVAL;
    const std::string od = std::string(basedir) + "/output/";
    std::vector<uint8_t> meta, pord, ptail, ordNpe, ordN;
    if (!slurp_file(od + "read_meta.txt", meta)) { harc_set_error("cannot read %sread_meta.txt", od.c_str()); return HARC_AMD_EIO; }
    meta.push_back(0);
    const int L = atoi((const char *)meta.data());
    harc_amd_params P = *params;
    if (harc_amd_default_params(L, &P) != HARC_AMD_OK) return HARC_AMD_EINVAL;
    P.device = params->device; P.num_thr = num_thr_e;
    if (!slurp_file(od + "read_order.bin", pord) || !slurp_file(od + "read_order.bin.tail", ptail) || !slurp_file(od + "read_order_N_pe.bin", ordNpe) ||
        !slurp_file(od + "read_order_N.bin", ordN)) { harc_set_error("order files missing: was the archive made with -p?"); return HARC_AMD_EIO; }
    harc_amd_ctx *c = nullptr;
    RC_TRY(harc_amd_create(&P, &c));
    struct Guard { harc_amd_ctx *c; ~Guard() { harc_amd_destroy(c); } } guard{ c };
    const size_t LL = (size_t)L + 1;
    // ---- unpack_order
    uint32_t nC = 0; int numbits = 0;
    if (pord.size() >= 8) { memcpy(&numbits, pord.data(), 4); memcpy(&nC, pord.data() + 4, 4); }
    const uint32_t ng = nC / 32, ntail = nC % 32;
    if (nC && (numbits < 1 || numbits > 32 || pord.size() != 8 + (size_t)ng * numbits * 4 || ptail.size() != (size_t)ntail * 4)) { harc_set_error("read_order.bin is not a pack_order file"); return HARC_AMD_EIO; }
    const uint32_t nN = (uint32_t)(ordN.size() / 4);
    if (ordNpe.size() != ordN.size()) { harc_set_error("read_order_N_pe.bin and read_order_N.bin disagree"); return HARC_AMD_EIO; }
    const uint64_t total64 = (uint64_t)nC + nN;
    if (total64 > 4294967290ull) { harc_set_error("more than 4294967290 reads"); return HARC_AMD_EINVAL; }
    const uint32_t total = (uint32_t)total64;
    uint32_t *d_order = nullptr, *d_ordNpe = nullptr, *d_ordN = nullptr, *flag = nullptr, *rankN = nullptr; unsigned int *d_err = nullptr;
    RC_TRY(dalloc(c, &d_order, (size_t)nC + 32)); RC_TRY(dalloc(c, &d_ordNpe, (size_t)nN + 1)); RC_TRY(dalloc(c, &d_ordN, (size_t)nN + 1)); RC_TRY(dalloc(c, &d_err, 4));
    RC_TRY(dalloc(c, &flag, (size_t)total + 1)); RC_TRY(dalloc(c, &rankN, (size_t)total + 1));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    if (ng) {
        const harc_mark_t mk = harc_pool_mark(c);
        uint8_t *d_p = nullptr; RC_TRY(up(c, pord.data() + 8, (size_t)ng * numbits * 4, &d_p));
        hipLaunchKernelGGL(k_unpack_order, G256((uint64_t)ng * 32), (const uint32_t *)d_p, ng, numbits, d_order);
        HIP_TRY(hipStreamSynchronize(c->stream));
        harc_pool_release(c, mk);
    }
    if (ntail) HIP_TRY(hipMemcpyAsync(d_order + (size_t)ng * 32, ptail.data(), (size_t)ntail * 4, hipMemcpyHostToDevice, c->stream));
    if (nN) { HIP_TRY(hipMemcpyAsync(d_ordNpe, ordNpe.data(), (size_t)nN * 4, hipMemcpyHostToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(d_ordN, ordN.data(), (size_t)nN * 4, hipMemcpyHostToDevice, c->stream)); }
    // which output lines are reads with N (merge_N.cpp:37-57), and how many of those come before each line
    HIP_TRY(hipMemsetAsync(flag, 0, ((size_t)total + 1) * 4, c->stream));
    if (nN) hipLaunchKernelGGL(k_mark_N, G256(nN), d_ordN, nN, total, flag, d_err);
    RC_TRY(prim_excl_scan_u32(c, flag, rankN, (size_t)total + 1));
    HIP_TRY(hipStreamSynchronize(c->stream));
    pord.clear(); pord.shrink_to_fit();

    // ---- bins of output lines.  The reference restores the order through host memory in bins of MAX_BIN_SIZE * 2e8 / 7 reads (-m,
    // decoder_preserve.cpp:249-253), reading the decoded reads of every bin back from a temporary file; here a bin is what fits in HBM next
    // to the scratch of one shard's decode (and no more than -m asks for), and every bin decodes the streams again and keeps the lines that
    // fall into it -- decoding is cheap, no temporary file, host memory bounded by one output chunk.
    uint64_t bin_lines;
    {
        const int mgb = P.decode_memory_gb > 3 ? P.decode_memory_gb : (P.decode_memory_gb == 0 ? 7 : 3);     // harc:225 default 7; decoder_preserve.cpp:249-252
        bin_lines = (uint64_t)mgb * 200000000ull / 7ull;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            fr += c->pool_total;
            // every bin decodes every shard in full: the scratch of the LARGEST shard (three line buffers + streams + scans, ~3 (L+1) + 64
            // bytes per read of the shard) is needed whatever the bin size, and is taken off before the bin gets its quarter
            uint64_t nmax = 0;
            for (int e = 0; e < num_thr_e; e++) {
                FILE *fp = fopen((od + "read_pos.txt." + std::to_string(e)).c_str(), "rb");
                if (fp) { if (fseek(fp, 0, SEEK_END) == 0) { const long sz = ftell(fp); if (sz > 0 && (uint64_t)sz > nmax) nmax = (uint64_t)sz; } fclose(fp); }
            }
            const double shard_scratch = (double)nmax * (3.0 * (double)LL + 64.0);
            if (shard_scratch > 0.9 * (double)fr) {
                harc_set_error("decoder_preserve: the largest shard (%llu reads) needs %.1f GB of decode scratch, %.1f GB are free; compress with more threads (-t) for smaller shards",
                               (unsigned long long)nmax, shard_scratch / 1e9, (double)fr / 1e9);
                return HARC_AMD_ENOMEM;
            }
            const uint64_t cap = (uint64_t)(0.25 * ((double)fr - shard_scratch) / (double)(3 * LL));     // clean + N + merged lines of the bin: a quarter of what is left
            if (bin_lines > (cap ? cap : 1)) bin_lines = cap ? cap : 1;
        }
        if (const char *e = getenv("HARC_AMD_BIN_READS")) bin_lines = strtoull(e, nullptr, 10);      // tests: tiny bins
        if (bin_lines < 1) bin_lines = 1;
    }
    FILE *fo = fopen((od + "output.dna").c_str(), "wb");
    if (!fo) { harc_set_error("cannot create %soutput.dna", od.c_str()); return HARC_AMD_EIO; }
    struct FClose { FILE *f; ~FClose() { if (f) fclose(f); } } fcl{ fo };
    std::vector<uint8_t> sg, sgt, nt;                             // singletons and unaligned N reads: read once
    if (!slurp_file(od + "read_singleton.txt", sg) || !slurp_file(od + "read_singleton.txt.tail", sgt)) { harc_set_error("singleton files missing"); return HARC_AMD_EIO; }
    slurp_file(od + "input_N.dna", nt);
    const harc_mark_t mark_bins = harc_pool_mark(c);
    for (uint64_t p0 = 0; p0 < total || (total == 0 && p0 == 0); p0 += bin_lines) {
        harc_pool_release(c, mark_bins);
        const uint32_t pn = (uint32_t)(total - p0 < bin_lines ? total - p0 : bin_lines);
        uint32_t r0 = 0, r1 = 0;
        HIP_TRY(hipMemcpyAsync(&r0, rankN + p0, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&r1, rankN + p0 + pn, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint32_t nlo = r0, nhi = r1, clo = (uint32_t)p0 - r0, chi = (uint32_t)(p0 + pn) - r1;     // the clean reads and the N reads of this bin
        char *clp = nullptr, *nlp = nullptr;
        RC_TRY(dalloc(c, &clp, (size_t)(chi - clo) * LL + 16)); RC_TRY(dalloc(c, &nlp, (size_t)(nhi - nlo) * LL + 16));
        // ---- decode every shard (stream order) and keep the lines of this bin: restore_order (decoder_preserve.cpp:246-290) and
        //      restore_order_N (:212-244)
        uint64_t cA = 0, cN = 0;
        for (int e = 0; e < num_thr_e; e++) {
            const std::string sfx = "." + std::to_string(e);
            std::vector<uint8_t> seq, seqt, pos, noise, npz, rev, revt;
            if (!slurp_file(od + "read_seq.txt" + sfx, seq) || !slurp_file(od + "read_seq.txt" + sfx + ".tail", seqt) || !slurp_file(od + "read_pos.txt" + sfx, pos) ||
                !slurp_file(od + "read_noise.txt" + sfx, noise) || !slurp_file(od + "read_noisepos.txt" + sfx, npz) ||
                !slurp_file(od + "read_rev.txt" + sfx, rev) || !slurp_file(od + "read_rev.txt" + sfx + ".tail", revt)) { harc_set_error("shard %d: stream files missing", e); return HARC_AMD_EIO; }
            if (pos.empty()) continue;
            if (pos.size() > 0xFFFFFFFFull || 8 * rev.size() + revt.size() != pos.size()) { harc_set_error("shard %d: rev stream does not match pos stream", e); return HARC_AMD_EIO; }
            const harc_mark_t mk = harc_pool_mark(c);
            const uint32_t n = (uint32_t)pos.size();
            uint8_t *d_seq, *d_seqt, *d_pos, *d_noise, *d_np, *d_rev, *d_revt, *seqb; uint64_t *p64, *possum, *nlpos; uint32_t *fl, *rk, *isN, *rkN; char *tmp, *outA, *outN;
            RC_TRY(up(c, seq.data(), seq.size(), &d_seq)); RC_TRY(up(c, seqt.data(), seqt.size(), &d_seqt)); RC_TRY(up(c, pos.data(), pos.size(), &d_pos));
            RC_TRY(up(c, noise.data(), noise.size(), &d_noise)); RC_TRY(up(c, npz.data(), npz.size(), &d_np)); RC_TRY(up(c, rev.data(), rev.size(), &d_rev)); RC_TRY(up(c, revt.data(), revt.size(), &d_revt));
            const uint64_t seqlen = 4 * (uint64_t)seq.size() + seqt.size();
            const size_t nnoise = noise.size();
            RC_TRY(dalloc(c, &seqb, (size_t)seqlen + 16)); RC_TRY(dalloc(c, &p64, (size_t)n + 1)); RC_TRY(dalloc(c, &possum, (size_t)n + 1)); RC_TRY(dalloc(c, &nlpos, (size_t)n + 1));
            RC_TRY(dalloc(c, &fl, nnoise + 1)); RC_TRY(dalloc(c, &rk, nnoise + 1)); RC_TRY(dalloc(c, &isN, (size_t)n + 1)); RC_TRY(dalloc(c, &rkN, (size_t)n + 1));
            RC_TRY(dalloc(c, &tmp, (size_t)n * LL + 16)); RC_TRY(dalloc(c, &outA, (size_t)n * LL + 16)); RC_TRY(dalloc(c, &outN, (size_t)n * LL + 16));
            if (seqlen) hipLaunchKernelGGL(k_unpack_seq, G256(seqlen), d_seq, (uint64_t)seq.size(), d_seqt, (uint64_t)seqt.size(), seqb);
            hipLaunchKernelGGL(k_pos_to_u64, G256(n), d_pos, n, p64);
            RC_TRY(prim_incl_scan_u64(c, p64, possum, n));
            if (nnoise) {
                hipLaunchKernelGGL(k_nl_flags, G256(nnoise), d_noise, (uint64_t)nnoise, fl);
                RC_TRY(prim_excl_scan_u32(c, fl, rk, nnoise));
                hipLaunchKernelGGL(k_nl_positions, G256(nnoise), d_noise, rk, (uint64_t)nnoise, nlpos);
            }
            HIP_TRY(hipMemsetAsync(isN, 0, ((size_t)n + 1) * 4, c->stream));
            hipLaunchKernelGGL(k_decode_text, G256(n), seqb, seqlen, possum, d_noise, d_np, nlpos, d_rev, (uint64_t)rev.size(), d_revt, n, L, tmp, isN, d_err);
            RC_TRY(prim_excl_scan_u32(c, isN, rkN, (size_t)n + 1));
            hipLaunchKernelGGL(k_split_lines, G256((uint64_t)n * LL), tmp, isN, rkN, n, L, outA, outN);
            uint32_t nNs = 0;
            HIP_TRY(hipMemcpyAsync(&nNs, rkN + n, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const uint32_t nAs = n - nNs;
            if (cA + nAs > nC || cN + nNs > nN) { harc_set_error("streams hold more reads than the order files"); return HARC_AMD_EIO; }
            if (nAs) hipLaunchKernelGGL(k_permute_lines, G256((uint64_t)nAs * LL), outA, d_order + cA, nAs, L, clp, clo, chi, nC, d_err);
            if (nNs) hipLaunchKernelGGL(k_permute_lines, G256((uint64_t)nNs * LL), outN, d_ordNpe + cN, nNs, L, nlp, nlo, nhi, nN, d_err);
            HIP_TRY(hipStreamSynchronize(c->stream));
            cA += nAs; cN += nNs;
            harc_pool_release(c, mk);
        }
        {   // singletons, then the N reads that were not aligned (decoder_preserve.cpp:160-197)
            const harc_mark_t mk = harc_pool_mark(c);
            const uint64_t nb = 4 * (uint64_t)sg.size() + sgt.size();
            const uint32_t ns = (uint32_t)(nb / L), nu = (uint32_t)(nt.size() / LL);
            if (cA + ns != nC || cN + nu != nN) { harc_set_error("read counts do not add up: clean %llu+%u vs %u, N %llu+%u vs %u", (unsigned long long)cA, ns, nC, (unsigned long long)cN, nu, nN); return HARC_AMD_EIO; }
            if (ns) {
                uint8_t *d_sg, *d_sgt, *codes; char *sl;
                RC_TRY(up(c, sg.data(), sg.size(), &d_sg)); RC_TRY(up(c, sgt.data(), sgt.size(), &d_sgt)); RC_TRY(dalloc(c, &codes, (size_t)nb + 16)); RC_TRY(dalloc(c, &sl, (size_t)ns * LL + 16));
                hipLaunchKernelGGL(k_unpack_seq, G256(nb), d_sg, (uint64_t)sg.size(), d_sgt, (uint64_t)sgt.size(), codes);
                hipLaunchKernelGGL(k_codes_to_lines, G256((uint64_t)ns * LL), codes, ns, L, sl);
                hipLaunchKernelGGL(k_permute_lines, G256((uint64_t)ns * LL), sl, d_order + cA, ns, L, clp, clo, chi, nC, d_err);
            }
            if (nu) {
                uint8_t *d_nt; RC_TRY(up(c, nt.data(), (size_t)nu * LL, &d_nt));
                hipLaunchKernelGGL(k_permute_lines, G256((uint64_t)nu * LL), (const char *)d_nt, d_ordNpe + cN, nu, L, nlp, nlo, nhi, nN, d_err);
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            harc_pool_release(c, mk);
        }
        // ---- merge_N for the lines of the bin, written in pieces of at most 256 MiB
        unsigned int err = 0;
        HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (err) { harc_set_error("decoder -p: %u inconsistent order / stream entries", err); return HARC_AMD_EIO; }
        const uint32_t piece = (uint32_t)(((size_t)256 << 20) / LL);
        char *outl = nullptr; RC_TRY(dalloc(c, &outl, (size_t)piece * LL + 16));
        std::vector<uint8_t> host;
        for (uint32_t q = 0; q < pn; q += piece) {
            const uint32_t m = pn - q < piece ? pn - q : piece;
            hipLaunchKernelGGL(k_merge_lines, G256((uint64_t)m * LL), clp, nlp, flag, rankN, (uint32_t)p0 + q, m, clo, nlo, L, outl);
            host.resize((size_t)m * LL);
            HIP_TRY(hipMemcpyAsync(host.data(), outl, host.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (fwrite(host.data(), 1, host.size(), fo) != host.size()) { harc_set_error("short write on output.dna"); return HARC_AMD_EIO; }
        }
        if (total == 0) break;
    }
    printf("Decoding done\n");
    return HARC_AMD_OK;
}
