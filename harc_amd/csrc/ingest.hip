// ingest.hip -- FASTQ ingest on the GPU: the work of src/preprocess.cpp:81-121 (take line 2 of every
// 4-line record, check the fixed read length, split reads with and without N, remember the original index of every N read) and of
// readDnaFile / stringtobitset (reorder.cpp:240-263,203-209) without the single-threaded getline loop and without the
// input_clean.dna round trip: the whole file is staged in HBM, a prefix sum over the newline flags gives the line index, one thread
// per record classifies, two compactions pack the reads straight into the 2-bit / 3-bit stores of the context.
#include "devutil.h"
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

// line index in two levels: newlines per 4096-byte tile -> scan of the tile counts -> positions written tile by tile
#define NL_TILE 4096
__device__ __forceinline__ uint32_t nl_count16(const char *txt, uint64_t n, uint64_t at, uint32_t *mask)
{
    uint32_t m = 0;
    for (int k = 0; k < 16; k++) if (at + k < n && txt[at + k] == '\n') m |= 1u << k;
    *mask = m;
    return (uint32_t)__popc(m);
}
__global__ __launch_bounds__(256) void k_nl_count(const char *txt, uint64_t n, uint32_t *tilecnt)
{
    __shared__ uint32_t sm[8];
    const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t m; uint32_t cnt = nl_count16(txt, n, at, &m);
    uint32_t tot; (void)block_excl_scan_u32<256>(cnt, sm, &tot);
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_nl_write(const char *txt, uint64_t n, const uint64_t *tilebase, uint64_t *nl)
{
    __shared__ uint32_t sm[8];
    const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t m; uint32_t cnt = nl_count16(txt, n, at, &m);
    uint32_t tot; const uint32_t off = block_excl_scan_u32<256>(cnt, sm, &tot);
    uint64_t o = tilebase[blockIdx.x] + off;
    while (m) { const int k = __ffs((int)m) - 1; m &= m - 1; nl[o++] = at + k; }
}
// record r = lines 4r .. 4r+3 ; sequence = line 4r+1 = bytes (nl[4r], nl[4r+1]).  flags: bit0 = has N, err counts bad lengths
__global__ void k_classify(const char *txt, const uint64_t *nl, uint32_t nrec, int L, uint32_t *isN, uint32_t *isClean, unsigned int *err)
{
    const uint32_t r = harc_gid32();
    if (r >= nrec) return;
    const uint64_t s = nl[4ull * r] + 1, e = nl[4ull * r + 1];
    uint64_t len = e - s;
    if (len && txt[e - 1] == '\r') len--;                         // tolerate CRLF
    if (len != (uint64_t)L) { atomicAdd(err, 1u); isN[r] = 0; isClean[r] = 0; return; }    // preprocess.cpp:92-97
    bool hasN = false;
    for (int j = 0; j < L; j++) hasN |= txt[s + j] == 'N';        // preprocess.cpp:98
    isN[r] = hasN ? 1u : 0u; isClean[r] = hasN ? 0u : 1u;
}
// clean reads -> 2-bit words (one thread per (record, word)); N reads -> 3-bit words; original index of N reads (read_order_N.bin)
__global__ void k_ingest_pack2(const char *txt, const uint64_t *nl, const uint32_t *isClean, const uint32_t *rankC, uint32_t nrec, int L, int W, uint64_t *out)
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)nrec * W) return;
    const uint32_t r = (uint32_t)(gid / W); const int w = (int)(gid % W);
    if (!isClean[r]) return;
    const char *s = txt + nl[4ull * r] + 1;
    uint64_t v = 0;
    for (int k = 0; k < 32; k++) {
        const int b = 32 * w + k;
        if (b < L) { const char ch = s[b]; v |= (uint64_t)(ch == 'A' ? 0 : ch == 'G' ? 1 : ch == 'C' ? 2 : 3) << (2 * k); }
    }
    out[(size_t)rankC[r] * W + w] = v;
}
__global__ void k_ingest_pack3(const char *txt, const uint64_t *nl, const uint32_t *isN, const uint32_t *rankN, uint32_t nrec, int L, int W3, uint64_t *out, uint32_t *orderN)
{
    const uint64_t gid = harc_gid();
    if (gid >= (uint64_t)nrec * W3) return;
    const uint32_t r = (uint32_t)(gid / W3); const int w = (int)(gid % W3);
    if (!isN[r]) return;
    const char *s = txt + nl[4ull * r] + 1;
    uint64_t v = 0;
    const int b0 = (64 * w) / 3, b1 = (64 * w + 63) / 3;
    for (int b = b0; b <= b1 && b < L; b++) {
        const char ch = s[b];
        const uint64_t c3 = ch == 'A' ? 0 : ch == 'N' ? 1 : ch == 'G' ? 2 : ch == 'C' ? 4 : 6;
        const int sh = 3 * b - 64 * w;
        v |= sh >= 0 ? (c3 << sh) : (c3 >> (-sh));
    }
    out[(size_t)rankN[r] * W3 + w] = v;
    if (w == 0) orderN[rankN[r]] = r;                             // preprocess.cpp:102
}

#define G256(n) harc_grid256((uint64_t)(n)), dim3(256), 0, c->stream

// nls[k] = byte position of the newline that ends line k (a last line without one ends at nbytes); nls[-1] = (u64)-1 so that line k
// starts at nls[k-1]+1 for every k.  Pool memory: the caller brackets it with harc_pool_mark / release.
static int build_line_index(harc_amd_ctx *c, const char *d_txt, uint64_t nbytes, const uint64_t **nls_out, uint64_t *total_lines_out)
{
    const uint64_t ntiles = (nbytes + NL_TILE - 1) / NL_TILE;
    uint32_t *tilecnt = nullptr; uint64_t *tilebase = nullptr;
    RC_TRY(dalloc(c, &tilecnt, (size_t)ntiles + 1)); RC_TRY(dalloc(c, &tilebase, (size_t)ntiles + 1));
    HIP_TRY(hipMemsetAsync(tilecnt + ntiles, 0, 4, c->stream));
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)ntiles), dim3(256), 0, c->stream, d_txt, nbytes, tilecnt);
    RC_TRY(prim_excl_scan_u32_to_u64(c, tilecnt, tilebase, (size_t)ntiles + 1));
    uint64_t nlines = 0; char lastch = 0;
    HIP_TRY(hipMemcpyAsync(&nlines, tilebase + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&lastch, d_txt + nbytes - 1, 1, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t *nl = nullptr; RC_TRY(dalloc(c, &nl, (size_t)nlines + 8));
    HIP_TRY(hipMemsetAsync(nl, 0xFF, 8, c->stream));
    hipLaunchKernelGGL(k_nl_write, dim3((unsigned)ntiles), dim3(256), 0, c->stream, d_txt, nbytes, (const uint64_t *)tilebase, nl + 1);
    uint64_t total_lines = nlines;
    if (lastch != '\n') {                                         // last line without a newline: it ends at nbytes
        HIP_TRY(hipMemcpyAsync(nl + 1 + nlines, &nbytes, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        total_lines++;
    }
    *nls_out = nl + 1; *total_lines_out = total_lines;
    return HARC_AMD_OK;
}

// FASTQ text in device memory -> the context's clean reads (2-bit) and N reads (3-bit), a chunk of whole records at a time: a file
// larger than HBM is ingested in pieces (harc_amd_compress_fastq_files_ex), the packed stores grow as the pieces arrive.
struct IngestState {
    uint64_t nC = 0, nN = 0, nrec = 0, nfull = 0;                 // clean reads, reads with N, reads, complete records so far
    std::vector<uint32_t> orderN;                                 // read_order_N.bin: record number of every read with N (preprocess.cpp:102)
    // -q without -p on a file that does not stay in HBM (emit_quality_and_ids_streamed): the length of every id line, the lines the file holds
    bool want_idlen = false;
    std::vector<uint32_t> idlen;
    uint64_t total_lines = 0;
};
__global__ void k_q_idlen(const uint64_t *nls, uint32_t nid, uint32_t *len)
{
    const uint32_t r = harc_gid32();
    if (r < nid) len[r] = (uint32_t)(nls[4ll * r] - (nls[4ll * r - 1] + 1));
}
static int ingest_begin(harc_amd_ctx *c, IngestState &st)
{
    { const bool w = st.want_idlen; st = IngestState(); st.want_idlen = w; }
    // same effect as harc_amd_set_reads_* on an empty set: previous inputs and results go
    RC_TRY(harc_amd_set_reads_packed_device(c, nullptr, 0));
    RC_TRY(harc_amd_set_nreads_ascii_device(c, nullptr, 0, (uint32_t)c->P.readlen));
    return HARC_AMD_OK;
}
// grows *b to `need` bytes keeping its first `keep` bytes
static int in_grow(harc_amd_ctx *c, harc_amd_ctx::InBuf *b, size_t need, size_t keep)
{
    if (b->p && b->cap >= need) return HARC_AMD_OK;
    harc_amd_ctx::InBuf nb;
    RC_TRY(harc_in_reserve(c, &nb, need + (keep ? need / 4 : 0)));
    if (keep) HIP_TRY(hipMemcpyAsync(nb.p, b->p, keep, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (b->p) harc_raw_free(c, b->p);
    *b = nb;
    return HARC_AMD_OK;
}
// one chunk of whole 4-line records (the last chunk of the file may end inside a record).  expect_*: the caller's estimate of the final
// counts (0 = unknown), so that the stores are sized once
static int ingest_append(harc_amd_ctx *c, IngestState &st, const char *d_txt, uint64_t nbytes, bool last, uint64_t expect_clean, uint64_t expect_N)
{
    if (nbytes == 0) return HARC_AMD_OK;
    const int L = c->P.readlen;
    const harc_mark_t mk = harc_pool_mark(c);
    struct Rel { harc_amd_ctx *c; harc_mark_t mk; ~Rel() { harc_pool_release(c, mk); } } rel{ c, mk };       // scratch goes on every way out
    const uint64_t *nls = nullptr; uint64_t total_lines = 0;
    RC_TRY(build_line_index(c, d_txt, nbytes, &nls, &total_lines));
    if (!last && (total_lines % 4)) { harc_set_error("FASTQ: a piece of the file does not hold whole 4-line records (%llu lines)", (unsigned long long)total_lines); return HARC_AMD_EINVAL; }
    // a truncated last record still counts as a read when its sequence line is there: the getline loop handles line 2 before it meets
    // the end of the file (preprocess.cpp:90-111); only `readnum` (case 3, :118) misses it
    const uint64_t nfull = total_lines / 4;
    const uint64_t nrec64 = nfull + ((total_lines % 4) >= 2 ? 1 : 0);
    if (st.nrec + nrec64 > 4294967290ull) { harc_set_error("Too many reads. HARC supports at most 4294967290 reads"); return HARC_AMD_EINVAL; }   // preprocess.cpp:122-126
    const uint32_t nrec = (uint32_t)nrec64;
    uint32_t *isN = nullptr, *isC = nullptr, *rkN = nullptr, *rkC = nullptr; unsigned int *d_err = nullptr;
    RC_TRY(dalloc(c, &isN, (size_t)nrec + 1)); RC_TRY(dalloc(c, &isC, (size_t)nrec + 1)); RC_TRY(dalloc(c, &rkN, (size_t)nrec + 1)); RC_TRY(dalloc(c, &rkC, (size_t)nrec + 1));
    RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    HIP_TRY(hipMemsetAsync(isN, 0, ((size_t)nrec + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(isC, 0, ((size_t)nrec + 1) * 4, c->stream));
    if (nrec) hipLaunchKernelGGL(k_classify, G256(nrec), d_txt, nls, nrec, L, isN, isC, d_err);
    RC_TRY(prim_excl_scan_u32(c, isN, rkN, (size_t)nrec + 1)); RC_TRY(prim_excl_scan_u32(c, isC, rkC, (size_t)nrec + 1));
    uint32_t nN = 0, nC = 0; unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(&nN, rkN + nrec, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&nC, rkC + nrec, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err) {
        printf("Read length not fixed. Found reads whose length is not %d\n", L);
        harc_set_error("read length not fixed (%u records differ from %d)", err, L); return HARC_AMD_EINVAL;
    }
    // the packed stores outlive the scratch: raw allocations (as harc_amd_set_reads_*), grown when a piece does not fit
    const uint64_t wantC = std::max<uint64_t>(st.nC + nC, expect_clean), wantN = std::max<uint64_t>(st.nN + nN, expect_N);
    RC_TRY(in_grow(c, &c->own_reads, ((size_t)wantC * c->W + 1) * 8, (size_t)st.nC * c->W * 8));
    RC_TRY(in_grow(c, &c->own_nreads3, ((size_t)wantN * c->W3 + 1) * 8, (size_t)st.nN * c->W3 * 8));
    uint32_t *orderN = nullptr; RC_TRY(dalloc(c, &orderN, (size_t)nN + 1));
    if (nrec) {
        hipLaunchKernelGGL(k_ingest_pack2, G256((uint64_t)nrec * c->W), d_txt, nls, isC, rkC, nrec, L, c->W, (uint64_t *)c->own_reads.p + (size_t)st.nC * c->W);
        hipLaunchKernelGGL(k_ingest_pack3, G256((uint64_t)nrec * c->W3), d_txt, nls, isN, rkN, nrec, L, c->W3, (uint64_t *)c->own_nreads3.p + (size_t)st.nN * c->W3, orderN);
    }
    HIP_TRY(hipGetLastError());
    const size_t at = st.orderN.size();
    st.orderN.resize(at + nN);
    if (nN) HIP_TRY(hipMemcpyAsync(st.orderN.data() + at, orderN, (size_t)nN * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = at; i < st.orderN.size(); i++) st.orderN[i] += (uint32_t)st.nrec;       // records of the earlier pieces come first
    if (st.want_idlen) {                                          // one id line per complete record, plus the one of a truncated last record
        const uint32_t nid = (uint32_t)nfull + (total_lines % 4 ? 1u : 0u);
        uint32_t *dl = nullptr; RC_TRY(dalloc(c, &dl, (size_t)nid + 1));
        if (nid) hipLaunchKernelGGL(k_q_idlen, G256(nid), nls, nid, dl);
        const size_t a0 = st.idlen.size();
        st.idlen.resize(a0 + nid);
        if (nid) HIP_TRY(hipMemcpyAsync(st.idlen.data() + a0, dl, (size_t)nid * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    st.total_lines += total_lines;
    st.nC += nC; st.nN += nN; st.nrec += nrec64; st.nfull += nfull;
    return HARC_AMD_OK;
}
static int ingest_finish(harc_amd_ctx *c, IngestState &st)
{
    c->N_own = (uint32_t)st.nC; c->NN_own = (uint32_t)st.nN;
    harc_reset_shard(c);
    c->nrec_own = st.nrec;
    std::vector<uint8_t> &ob = out_buf(c, HARC_AMD_IN_ORDER_N, 0);
    ob.resize(st.orderN.size() * 4);
    if (!st.orderN.empty()) memcpy(ob.data(), st.orderN.data(), ob.size());
    return HARC_AMD_OK;
}

// the whole FASTQ text at once; the original indices of the N reads (read_order_N.bin, u32 each) are returned through
// harc_amd_get_stream(HARC_AMD_IN_ORDER_N)
extern "C" int harc_amd_set_fastq_device(harc_amd_ctx *c, const char *d_txt, uint64_t nbytes, uint64_t *n_records_out)
{
    if (!c || (nbytes && !d_txt)) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    IngestState st;
    RC_TRY(ingest_begin(c, st));
    RC_TRY(ingest_append(c, st, d_txt, nbytes, true, 0, 0));
    RC_TRY(ingest_finish(c, st));
    if (n_records_out) *n_records_out = st.nfull;                 // what preprocess.cpp:134 prints: complete records
    return HARC_AMD_OK;
}


// ------------------------------------------------------------------------------------------------ -q: ids and quality values
// preprocess.cpp:61-118 + reorder_quality.cpp as gathers over the line index.  `rec[p]` names the record whose
// line `k` (0 = id, 3 = quality) becomes output line p.
__global__ void k_q_idflags(const uint32_t *isN, uint32_t nid, uint32_t *idC, uint32_t *idN)
{
    const uint32_t r = harc_gid32();
    if (r >= nid) return;
    const uint32_t prevN = r ? isN[r - 1] : 0u;                    // preprocess.cpp:83-88 runs before :98-110 of the same record
    idC[r] = prevN ? 0u : 1u; idN[r] = prevN;
}
__global__ void k_q_compact(const uint32_t *flag, const uint32_t *rank, uint32_t n, uint32_t *out)
{
    const uint32_t r = harc_gid32();
    if (r < n && flag[r]) out[rank[r]] = r;
}
__global__ void k_q_gather(const uint32_t *src, uint32_t nsrc, const uint32_t *order, uint32_t n, uint32_t *out, unsigned int *err)
{
    const uint32_t p = harc_gid32();
    if (p >= n) return;
    const uint32_t o = order[p];
    if (o >= nsrc) { atomicAdd(err, 1u); out[p] = 0; return; }
    out[p] = src[o];
}
__global__ void k_q_iota(uint32_t *out, uint32_t n) { const uint32_t p = harc_gid32(); if (p < n) out[p] = p; }
__global__ void k_q_linelen(const uint64_t *nls, const uint32_t *rec, uint32_t n, int k, int want, uint32_t *len, unsigned int *err)
{
    const uint32_t p = harc_gid32();
    if (p >= n) return;
    const int64_t li = 4ll * rec[p] + k;
    const uint64_t l = nls[li] - (nls[li - 1] + 1);
    if (want >= 0 && l != (uint64_t)want) atomicAdd(err, 1u);      // reorder_quality.cpp:78-79 reads quality values at a (readlen+1) stride
    len[p] = (uint32_t)l + 1;
}
// one wave per output line
__global__ __launch_bounds__(256) void k_q_copy(const char *txt, const uint64_t *nls, const uint32_t *rec, const uint64_t *off, uint32_t n, int k, char *out)
{
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t li = 4ll * rec[p] + k;
    const uint64_t s = nls[li - 1] + 1, l = nls[li] - s;
    char *o = out + off[p];
    for (uint64_t j = lane; j < l; j += 64) o[j] = txt[s + j];
    if (lane == 0) o[l] = '\n';
}

static int emit_lines(harc_amd_ctx *c, const char *d_txt, const uint64_t *nls, const uint32_t *rec, uint64_t n, int k, int want, FILE *fo)
{
    const uint32_t CH = 1u << 23;
    unsigned int *d_err = nullptr; RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    std::vector<uint8_t> host;
    for (uint64_t at = 0; at < n; at += CH) {
        const harc_mark_t mk = harc_pool_mark(c);
        struct Rel { harc_amd_ctx *c; harc_mark_t mk; ~Rel() { harc_pool_release(c, mk); } } rel{ c, mk };
        const uint32_t m = (uint32_t)(n - at < CH ? n - at : CH);
        uint32_t *len = nullptr; uint64_t *off = nullptr;
        RC_TRY(dalloc(c, &len, (size_t)m + 1)); RC_TRY(dalloc(c, &off, (size_t)m + 1));
        HIP_TRY(hipMemsetAsync(len + m, 0, 4, c->stream));
        hipLaunchKernelGGL(k_q_linelen, G256(m), nls, rec + at, m, k, want, len, d_err);
        RC_TRY(prim_excl_scan_u32_to_u64(c, len, off, (size_t)m + 1));
        uint64_t total = 0; unsigned int err = 0;
        HIP_TRY(hipMemcpyAsync(&total, off + m, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (err) { harc_set_error("-q without -p needs quality lines of exactly readlen characters (%u differ)", err); return HARC_AMD_EINVAL; }
        char *out = nullptr; RC_TRY(dalloc(c, &out, (size_t)total + 16));
        hipLaunchKernelGGL(k_q_copy, dim3((m + 3) / 4), dim3(256), 0, c->stream, d_txt, nls, rec + at, (const uint64_t *)off, m, k, out);
        RC_TRY(harc_d2h(c, host, out, (size_t)total));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (total && fwrite(host.data(), 1, (size_t)total, fo) != (size_t)total) { harc_set_error("short write"); return HARC_AMD_EIO; }
    }
    return HARC_AMD_OK;
}

// -q -p (preprocess.cpp:64-69): quality values and ids in file order; one piece of whole records at a time, appended to the open files
static int emit_q_fileorder(harc_amd_ctx *c, const char *d_txt, uint64_t nbytes, FILE *fq, FILE *fi)
{
    if (nbytes == 0) return HARC_AMD_OK;
    const harc_mark_t mk = harc_pool_mark(c);
    struct Rel { harc_amd_ctx *c; harc_mark_t mk; ~Rel() { harc_pool_release(c, mk); } } rel{ c, mk };
    const uint64_t *nls = nullptr; uint64_t total_lines = 0;
    RC_TRY(build_line_index(c, d_txt, nbytes, &nls, &total_lines));
    const uint32_t nrec = (uint32_t)(total_lines / 4);
    const uint32_t nid = nrec + (total_lines % 4 ? 1u : 0u);       // the id line of a truncated last record is still written (case 0 of the getline loop)
    uint32_t *rec = nullptr; RC_TRY(dalloc(c, &rec, (size_t)nid + 1));
    hipLaunchKernelGGL(k_q_iota, G256((size_t)nid + 1), rec, nid + 1);
    RC_TRY(emit_lines(c, d_txt, nls, rec, nrec, 3, -1, fq));
    RC_TRY(emit_lines(c, d_txt, nls, rec, nid, 0, -1, fi));
    return HARC_AMD_OK;
}

// Which record's quality line / id line becomes output line p (reorder_quality.cpp:47-133, :140-208), from the per-record flags: qrec[nC + nN],
// irec[nC + nidN].  Pool memory of the caller's bracket.
static int q_routes(harc_amd_ctx *c, const uint32_t *isN, const uint32_t *isC, uint32_t nrec, uint32_t nid, unsigned int *d_err, uint32_t **qrec_out, uint32_t **irec_out, uint32_t *nidN_out)
{
    const void *ho = nullptr, *hn = nullptr; size_t ho_len = 0, hn_len = 0;
    RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_ORDER, 0, &ho, &ho_len)); RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_ORDER_N_PE, 0, &hn, &hn_len));
    const uint32_t nC = c->N, nN = c->NN;
    if (ho_len != (size_t)nC * 4 || hn_len != (size_t)nN * 4) { harc_set_error("-q: order streams do not match the read counts"); return HARC_AMD_ESTATE; }
    uint32_t *rk, *idC, *idN, *cleanrec, *nrecs, *idcrec, *idnrec, *d_ord, *d_ordn, *qrec, *irec;
    RC_TRY(dalloc(c, &rk, (size_t)nid + 2));
    RC_TRY(dalloc(c, &idC, (size_t)nid + 1)); RC_TRY(dalloc(c, &idN, (size_t)nid + 1));
    RC_TRY(dalloc(c, &cleanrec, (size_t)nC + 1)); RC_TRY(dalloc(c, &nrecs, (size_t)nN + 1)); RC_TRY(dalloc(c, &idcrec, (size_t)nid + 1)); RC_TRY(dalloc(c, &idnrec, (size_t)nid + 1));
    RC_TRY(dalloc(c, &d_ord, (size_t)nC + 1)); RC_TRY(dalloc(c, &d_ordn, (size_t)nN + 1)); RC_TRY(dalloc(c, &qrec, (size_t)nC + nN + 1)); RC_TRY(dalloc(c, &irec, (size_t)nC + nid + 1));
    if (nC) HIP_TRY(hipMemcpyAsync(d_ord, ho, ho_len, hipMemcpyHostToDevice, c->stream));
    if (nN) HIP_TRY(hipMemcpyAsync(d_ordn, hn, hn_len, hipMemcpyHostToDevice, c->stream));
    // quality: clean records gathered by read_order.bin, then N records gathered by read_order_N_pe.bin (reorder_quality.cpp:47-133)
    RC_TRY(prim_excl_scan_u32(c, isC, rk, (size_t)nrec + 1));
    if (nrec) hipLaunchKernelGGL(k_q_compact, G256(nrec), isC, rk, nrec, cleanrec);
    RC_TRY(prim_excl_scan_u32(c, isN, rk, (size_t)nrec + 1));
    if (nrec) hipLaunchKernelGGL(k_q_compact, G256(nrec), isN, rk, nrec, nrecs);
    if (nC) hipLaunchKernelGGL(k_q_gather, G256(nC), cleanrec, nC, d_ord, nC, qrec, d_err);
    if (nN) hipLaunchKernelGGL(k_q_gather, G256(nN), nrecs, nN, d_ordn, nN, qrec + nC, d_err);
    // ids: routed by the PREVIOUS record's N flag; the clean list gathered by read_order.bin, the N list appended as it is
    // (reorder_id_N writes its result over input_N.quality, reorder_quality.cpp:208, so reorder_id appends input_N.id untouched, :181)
    uint32_t nidC = 0, nidN = 0;
    if (nid) {
        hipLaunchKernelGGL(k_q_idflags, G256(nid), isN, nid, idC, idN);
        HIP_TRY(hipMemsetAsync(idC + nid, 0, 4, c->stream)); HIP_TRY(hipMemsetAsync(idN + nid, 0, 4, c->stream));
        RC_TRY(prim_excl_scan_u32(c, idC, rk, (size_t)nid + 1));
        hipLaunchKernelGGL(k_q_compact, G256(nid), idC, rk, nid, idcrec);
        HIP_TRY(hipMemcpyAsync(&nidC, rk + nid, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        RC_TRY(prim_excl_scan_u32(c, idN, rk, (size_t)nid + 1));
        hipLaunchKernelGGL(k_q_compact, G256(nid), idN, rk, nid, idnrec);
        nidN = nid - nidC;
    }
    if (nC) hipLaunchKernelGGL(k_q_gather, G256(nC), idcrec, nidC, d_ord, nC, irec, d_err);
    if (nidN) HIP_TRY(hipMemcpyAsync(irec + nC, idnrec, (size_t)nidN * 4, hipMemcpyDeviceToDevice, c->stream));
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err) { harc_set_error("-q: order files and FASTQ disagree (%u entries)", err); return HARC_AMD_ESTATE; }
    *qrec_out = qrec; *irec_out = irec; *nidN_out = nidN;
    return HARC_AMD_OK;
}

static int emit_quality_and_ids(harc_amd_ctx *c, const char *d_txt, uint64_t nbytes, bool preserve_order, const std::string &od, const char *qname, const char *iname)
{
    FILE *fq = fopen((od + qname).c_str(), "wb"), *fi = fopen((od + iname).c_str(), "wb");
    struct Closer { FILE *a, *b; ~Closer() { if (a) fclose(a); if (b) fclose(b); } } closer{ fq, fi };
    if (!fq || !fi) { harc_set_error("cannot create %soutput.quality / output.id", od.c_str()); return HARC_AMD_EIO; }
    if (nbytes == 0) return HARC_AMD_OK;
    const int L = c->P.readlen;
    const harc_mark_t mk = harc_pool_mark(c);
    struct Rel { harc_amd_ctx *c; harc_mark_t mk; ~Rel() { harc_pool_release(c, mk); } } rel{ c, mk };
    const uint64_t *nls = nullptr; uint64_t total_lines = 0;
    RC_TRY(build_line_index(c, d_txt, nbytes, &nls, &total_lines));
    const uint32_t nrec = (uint32_t)(total_lines / 4);
    const uint32_t nid = nrec + (total_lines % 4 ? 1u : 0u);       // the id line of a truncated last record is still written (case 0 of the getline loop)
    if (!preserve_order && (total_lines % 4) >= 2) {               // that read has no quality line: reorder_quality.cpp would walk off its arrays
        harc_set_error("-q without -p: the last FASTQ record is truncated (its read is kept, its quality line is missing)"); return HARC_AMD_EINVAL;
    }
    if (preserve_order) {                                          // preprocess.cpp:64-69: both files in file order
        uint32_t *rec = nullptr; RC_TRY(dalloc(c, &rec, (size_t)nid + 1));
        hipLaunchKernelGGL(k_q_iota, G256((size_t)nid + 1), rec, nid + 1);
        RC_TRY(emit_lines(c, d_txt, nls, rec, nrec, 3, -1, fq));
        RC_TRY(emit_lines(c, d_txt, nls, rec, nid, 0, -1, fi));
        return HARC_AMD_OK;
    }
    uint32_t *isN, *isC; unsigned int *d_err;
    RC_TRY(dalloc(c, &isN, (size_t)nid + 1)); RC_TRY(dalloc(c, &isC, (size_t)nid + 1)); RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    HIP_TRY(hipMemsetAsync(isN, 0, ((size_t)nid + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(isC, 0, ((size_t)nid + 1) * 4, c->stream));
    if (nrec) hipLaunchKernelGGL(k_classify, G256(nrec), d_txt, nls, nrec, L, isN, isC, d_err);
    uint32_t *qrec = nullptr, *irec = nullptr, nidN = 0;
    RC_TRY(q_routes(c, isN, isC, nrec, nid, d_err, &qrec, &irec, &nidN));
    const uint32_t nC = c->N, nN = c->NN;
    RC_TRY(emit_lines(c, d_txt, nls, qrec, (uint64_t)nC + nN, 3, L, fq));
    RC_TRY(emit_lines(c, d_txt, nls, irec, (uint64_t)nC + nidN, 0, -1, fi));
    return HARC_AMD_OK;
}

// ---- -q without -p when the FASTQ text does not stay in HBM: the reference permutes quality values and ids in 4-8 bins of host memory, reading
// the files once per bin (reorder_quality.cpp:61-77).  Here: the routes (which record's line becomes output line p) come from the per-record N
// flags and the id-line lengths kept by the ingest; the OUTPUT is cut into bins that fit HBM, and for every bin the FASTQ file is streamed
// through the GPU once more, piece by piece, every line that belongs to the bin copied to its place (quality lines have a fixed stride -- they
// must be readlen long, reorder_quality.cpp:78-79 -- id lines go by a prefix sum of their lengths).  Quality and id bins share a pass.
__global__ void k_q_scatter_flag(const uint32_t *idx, uint32_t n, uint32_t lim, uint32_t *flag, unsigned int *err)
{
    const uint32_t i = harc_gid32();
    if (i >= n) return;
    const uint32_t r = idx[i];
    if (r >= lim) { atomicAdd(err, 1u); return; }
    flag[r] = 1u;
}
__global__ void k_q_not(const uint32_t *isN, uint32_t n, uint32_t *isC) { const uint32_t r = harc_gid32(); if (r < n) isC[r] = isN[r] ? 0u : 1u; }
__global__ void k_q_invert(const uint32_t *rec, uint32_t n, uint32_t *pos) { const uint32_t p = harc_gid32(); if (p < n) pos[rec[p]] = p; }
__global__ void k_q_outlen(const uint32_t *idlen, const uint32_t *rec, uint32_t n, uint32_t *len) { const uint32_t p = harc_gid32(); if (p < n) len[p] = idlen[rec[p]] + 1u; }
// first p in [lo, n] with off[p] - off[lo] > budget, minus one (at least lo + 1 when lo < n): the end of the bin that starts at lo
__global__ void k_q_bin_end(const uint64_t *off, uint32_t lo, uint32_t n, uint64_t budget, uint32_t *out)
{
    if (blockIdx.x || threadIdx.x) return;
    const uint64_t base = off[lo];
    uint32_t a = lo, b = n;                                       // largest p with off[p] - base <= budget
    while (a < b) { const uint32_t mid = a + (b - a + 1) / 2; if (off[mid] - base <= budget) a = mid; else b = mid - 1; }
    if (a == lo && lo < n) a = lo + 1;                            // a single line longer than the budget still goes out
    out[0] = a;
}
// one wave per record of the piece: its quality line and its id line, each if its output line lies in the bin being filled
__global__ __launch_bounds__(256) void k_q_place(const char *txt, const uint64_t *nls, uint32_t nrec, uint32_t nid, uint32_t base, int L,
                                                 const uint32_t *posQ, uint32_t q0, uint32_t q1, char *outQ,
                                                 const uint32_t *posI, const uint64_t *offI, uint32_t i0, uint32_t i1, char *outI, unsigned int *err)
{
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r < nrec) {
        const uint32_t p = posQ[base + r];
        if (p >= q0 && p < q1) {
            const uint64_t s = nls[4ll * r + 2] + 1, l = nls[4ll * r + 3] - s;
            if (l != (uint64_t)L) { if (lane == 0) atomicAdd(err, 1u); }
            else {
                char *o = outQ + (size_t)(p - q0) * (size_t)(L + 1);
                for (uint64_t j = lane; j < l; j += 64) o[j] = txt[s + j];
                if (lane == 0) o[l] = '\n';
            }
        }
    }
    if (r < nid) {
        const uint32_t p = posI[base + r];
        if (p >= i0 && p < i1) {
            const uint64_t s = nls[4ll * r - 1] + 1, l = nls[4ll * r] - s;
            char *o = outI + (offI[p] - offI[i0]);
            for (uint64_t j = lane; j < l; j += 64) o[j] = txt[s + j];
            if (lane == 0) o[l] = '\n';
        }
    }
}
static int load_file_range(harc_amd_ctx *c, FILE *f, const char *name, uint64_t lo, uint64_t hi, char **d_txt);
static int record_start_at_or_after(FILE *f, uint64_t pos, uint64_t fsz, uint64_t *out);
static int write_device_range(harc_amd_ctx *c, const char *d, size_t n, FILE *fo)
{
    std::vector<uint8_t> host;
    const size_t CH = (size_t)256 << 20;
    for (size_t at = 0; at < n; at += CH) {
        const size_t m = n - at < CH ? n - at : CH;
        RC_TRY(harc_d2h(c, host, d + at, m));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (fwrite(host.data(), 1, m, fo) != m) { harc_set_error("short write"); return HARC_AMD_EIO; }
    }
    return HARC_AMD_OK;
}
static int emit_quality_and_ids_streamed(harc_amd_ctx *c, FILE *f, const char *name, uint64_t fsz, const IngestState &st, const std::string &od, const char *qname, const char *iname)
{
    FILE *fq = fopen((od + qname).c_str(), "wb"), *fi = fopen((od + iname).c_str(), "wb");
    struct Closer { FILE *a, *b; ~Closer() { if (a) fclose(a); if (b) fclose(b); } } closer{ fq, fi };
    if (!fq || !fi) { harc_set_error("cannot create %soutput.quality / output.id", od.c_str()); return HARC_AMD_EIO; }
    if (fsz == 0) return HARC_AMD_OK;
    const int L = c->P.readlen;
    if ((st.total_lines % 4) >= 2) { harc_set_error("-q without -p: the last FASTQ record is truncated (its read is kept, its quality line is missing)"); return HARC_AMD_EINVAL; }
    const uint32_t nrec = (uint32_t)st.nfull, nid = nrec + (st.total_lines % 4 ? 1u : 0u);
    if (st.idlen.size() != (size_t)nid) { harc_set_error("-q: %zu id lines recorded, %u expected", st.idlen.size(), nid); return HARC_AMD_EINTERNAL; }
    PoolScope scope(c);
    uint32_t *isN, *isC; unsigned int *d_err;
    RC_TRY(dalloc(c, &isN, (size_t)nid + 1)); RC_TRY(dalloc(c, &isC, (size_t)nid + 1)); RC_TRY(dalloc(c, &d_err, 4));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, c->stream));
    HIP_TRY(hipMemsetAsync(isN, 0, ((size_t)nid + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(isC, 0, ((size_t)nid + 1) * 4, c->stream));
    {   // the N flags of k_classify, from read_order_N.bin
        PoolScope tmp(c);
        const uint32_t nN = (uint32_t)st.orderN.size();
        uint32_t *d_on = nullptr; RC_TRY(dalloc(c, &d_on, (size_t)nN + 1));
        if (nN) { HIP_TRY(hipMemcpyAsync(d_on, st.orderN.data(), (size_t)nN * 4, hipMemcpyHostToDevice, c->stream)); hipLaunchKernelGGL(k_q_scatter_flag, G256(nN), d_on, nN, nrec, isN, d_err); }
        if (nrec) hipLaunchKernelGGL(k_q_not, G256(nrec), isN, nrec, isC);
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    uint32_t *qrec = nullptr, *irec = nullptr, nidN = 0;
    RC_TRY(q_routes(c, isN, isC, nrec, nid, d_err, &qrec, &irec, &nidN));
    const uint32_t nC = c->N, nN = c->NN, nQ = nC + nN, nI = nC + nidN;
    uint32_t *posQ, *posI, *d_idlen, *lenI, *d_end; uint64_t *offI;
    RC_TRY(dalloc(c, &posQ, (size_t)nrec + 1)); RC_TRY(dalloc(c, &posI, (size_t)nid + 1)); RC_TRY(dalloc(c, &d_idlen, (size_t)nid + 1));
    RC_TRY(dalloc(c, &lenI, (size_t)nI + 1)); RC_TRY(dalloc(c, &offI, (size_t)nI + 1)); RC_TRY(dalloc(c, &d_end, 4));
    HIP_TRY(hipMemsetAsync(posQ, 0xFF, ((size_t)nrec + 1) * 4, c->stream)); HIP_TRY(hipMemsetAsync(posI, 0xFF, ((size_t)nid + 1) * 4, c->stream));
    if (nQ) hipLaunchKernelGGL(k_q_invert, G256(nQ), qrec, nQ, posQ);
    if (nI) hipLaunchKernelGGL(k_q_invert, G256(nI), irec, nI, posI);
    if (nid) HIP_TRY(hipMemcpyAsync(d_idlen, st.idlen.data(), (size_t)nid * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(lenI + nI, 0, 4, c->stream));
    if (nI) hipLaunchKernelGGL(k_q_outlen, G256(nI), d_idlen, irec, nI, lenI);
    RC_TRY(prim_excl_scan_u32_to_u64(c, lenI, offI, (size_t)nI + 1));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // bytes of output one pass holds, per file
    size_t budget;
    {
        size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot);
        budget = (size_t)(0.25 * (double)fr);
        if (budget > ((size_t)32 << 30)) budget = (size_t)32 << 30;
        if (budget < ((size_t)64 << 20)) budget = (size_t)64 << 20;
        if (const char *e = getenv("HARC_AMD_Q_BIN")) { const unsigned long long v = strtoull(e, nullptr, 10); if (v >= 1) budget = (size_t)v; }       // tests: bins of a few lines
    }
    uint64_t piece = (uint64_t)1 << 30;
    if (const char *e = getenv("HARC_AMD_INGEST_CHUNK")) { piece = strtoull(e, nullptr, 10); if (piece < 16) piece = 16; }
    uint32_t q0 = 0, i0 = 0; int passes = 0;
    while (q0 < nQ || i0 < nI) {
        PoolScope pass_scope(c);
        uint64_t perq = budget / (uint64_t)(L + 1); if (perq < 1) perq = 1;
        const uint32_t q1 = (uint64_t)nQ - q0 > perq ? q0 + (uint32_t)perq : nQ;
        uint32_t i1 = i0;
        uint64_t ob[2] = { 0, 0 };
        if (i0 < nI) {
            hipLaunchKernelGGL(k_q_bin_end, dim3(1), dim3(1), 0, c->stream, (const uint64_t *)offI, i0, nI, (uint64_t)budget, d_end);
            HIP_TRY(hipMemcpyAsync(&i1, d_end, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpyAsync(&ob[0], offI + i0, 8, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(&ob[1], offI + i1, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        const size_t bytesQ = (size_t)(q1 - q0) * (size_t)(L + 1), bytesI = (size_t)(ob[1] - ob[0]);
        char *outQ = nullptr, *outI = nullptr;
        RC_TRY(dalloc(c, &outQ, bytesQ + 16)); RC_TRY(dalloc(c, &outI, bytesI + 16));
        uint64_t lo = 0; uint32_t base = 0;
        while (lo < fsz) {
            uint64_t hi = fsz;
            if (fsz - lo > piece) { RC_TRY(record_start_at_or_after(f, lo + piece, fsz, &hi)); if (hi <= lo) hi = fsz; }
            char *d_txt = nullptr;
            RC_TRY(load_file_range(c, f, name, lo, hi, &d_txt));
            struct Free { harc_amd_ctx *c; char *p; ~Free() { harc_raw_free(c, p); } } fr{ c, d_txt };
            PoolScope piece_scope(c);
            const uint64_t *nls = nullptr; uint64_t tl = 0;
            RC_TRY(build_line_index(c, d_txt, hi - lo, &nls, &tl));
            const uint32_t pr = (uint32_t)(tl / 4), pi = pr + (tl % 4 ? 1u : 0u);
            if ((uint64_t)base + pi > (uint64_t)nid) { harc_set_error("-q: the FASTQ file changed between the passes"); return HARC_AMD_EIO; }
            if (pi) hipLaunchKernelGGL(k_q_place, dim3((pi + 3) / 4), dim3(256), 0, c->stream, (const char *)d_txt, nls, pr, pi, base, L, (const uint32_t *)posQ, q0, q1, outQ,
                                       (const uint32_t *)posI, (const uint64_t *)offI, i0, i1, outI, d_err);
            HIP_TRY(hipStreamSynchronize(c->stream));
            base += pr; lo = hi;
        }
        unsigned int err = 0;
        HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (err) { harc_set_error("-q without -p needs quality lines of exactly readlen characters (%u differ)", err); return HARC_AMD_EINVAL; }
        RC_TRY(write_device_range(c, outQ, bytesQ, fq)); RC_TRY(write_device_range(c, outI, bytesI, fi));
        q0 = q1; i0 = i1; passes++;
    }
    if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[-q] quality values and ids permuted in %d passes over the FASTQ file (%zu bytes of each per pass)\n", passes, budget);
    return HARC_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ file drivers
// Where the wall time of the last harc_amd_compress_fastq_files_ex of this process went (seconds; harc_amd_last_fastq_timing): [0] context + device pool,
// [1] ingest = file -> HBM -> packed stores, reads / uploads / kernels overlapped, [2] of it the calling thread waiting for the reader threads (file-read bound),
// [3] of it the device's line index / classify / pack kernels and their syncs, [4] reorder, [5] encode (the D2H of the streams inside), [6] stream files written,
// [7] total
static double g_fastq_timing[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
static inline double mono_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
extern "C" int harc_amd_last_fastq_timing(double *out, int32_t n)
{
    if (!out || n < 1) return HARC_AMD_EINVAL;
    for (int i = 0; i < n; i++) out[i] = i < 8 ? g_fastq_timing[i] : 0.0;
    return HARC_AMD_OK;
}
// File -> HBM at the rate of the host's memory system instead of one core's (round 6; round 3's ./harc -c spent most of its 6 s on 100 M reads in a
// single-threaded fread into one pinned buffer): reader threads pread() slices of the file into a ring of pinned slices kept by the context, the calling
// thread uploads every filled slice (hipMemcpyAsync on the context's stream) and hands the slice back when its copy has finished.  The slices of ALL the
// pieces of a range are read ahead in file order as far as the ring goes: while the device indexes and packs piece i the readers already hold the first
// slices of piece i + 1.  A slice is taken from the ring BEFORE its chunk number, under one lock: the chunks that hold slices are always the lowest
// unfinished ones, so a piece being waited for can never starve behind read-ahead that cannot be uploaded yet.
struct FileFeeder {
    struct Chunk { uint64_t off; uint32_t len; uint32_t piece; uint64_t at; };     // file offset, bytes, piece, byte offset inside the piece
    harc_amd_ctx *c; int fd = -1; const char *name;
    bool use_mmap = true;                                         // the readers copy out of a mapping of their slice instead of calling pread (HARC_AMD_FEED_MMAP=0: pread)
    std::vector<Chunk> chunks; std::vector<size_t> piece_left;                     // chunks of each piece not uploaded yet
    size_t SL = 0; int NS = 0;
    std::vector<hipEvent_t> ev;
    std::mutex mu; std::condition_variable cv_free, cv_filled;
    std::deque<int> free_slices; std::deque<std::pair<int, size_t>> filled, held;  // (slice, chunk)
    std::deque<int> inflight;                                                      // slices whose upload is on the stream, oldest first
    size_t next_chunk = 0; bool stop = false; int err = 0;
    double t_ring = 0, t_pread = 0, t_wait_free = 0, t_wait_filled = 0, t_wait_copy = 0, t_enqueue = 0;      // HARC_AMD_TRACE: summed over the readers / of the calling thread
    std::vector<std::thread> th;
    FileFeeder(harc_amd_ctx *c_, const char *name_) : c(c_), name(name_) {}
    ~FileFeeder()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_free.notify_all();
        for (auto &t : th) t.join();
        (void)hipStreamSynchronize(c->stream);                    // before the ring is used again
        if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[file feeder] %zu slices of %zu MB through %d pinned slices by %zu readers (pinned ring allocated in %.3f s): readers in pread %.2f s, waiting for a free slice %.2f s (summed); uploader enqueueing %.2f s, waiting for a filled slice %.2f s, for a copy %.2f s\n",
                                              chunks.size(), SL >> 20, NS, th.size(), t_ring, t_pread, t_wait_free, t_enqueue, t_wait_filled, t_wait_copy);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (fd >= 0) close(fd);
    }
    // pieces: [lo, hi) byte ranges of the file, in file order
    int start(const std::vector<std::pair<uint64_t, uint64_t>> &pieces)
    {
        fd = open(name, O_RDONLY);
        if (fd < 0) { harc_set_error("cannot open %s", name); return HARC_AMD_EIO; }
        // 16 slices of 64 MB, 16 readers (tools/micro/feed_rate.cpp, profiles/r06/feed_rate.txt: a file that has been read before reaches HBM at 54 GB/s this way, the
        // PCIe rate is 57; 16-MB slices 42; the FIRST read of a freshly written tmpfs file runs at 24 GB/s whatever is done here -- the kernel's own first touch)
        SL = (size_t)64 << 20; NS = 16;
        int nthr = 16;
        // a read() of page-cache pages that nobody has read yet marks every one of them accessed (LRU lists, under a lock the readers share): the first read of
        // a freshly written 21.7-GB file ran at 14-24 GB/s with 16 readers, the second at 54.  Copies out of a shared mapping do not go that way -- but ONE
        // mapping of the whole file took 0.8 s to take down again (the same marking, at unmap, by one thread): every reader maps its own slice, tells the kernel
        // that it reads it once from front to back (no recency kept for such a mapping), copies and unmaps.  HARC_AMD_FEED_MMAP=0: pread -- the way to read a file
        // that somebody may TRUNCATE meanwhile: a copy out of a mapping beyond the new end of the file is a SIGBUS, not a short read.
        use_mmap = !(getenv("HARC_AMD_FEED_MMAP") && atoi(getenv("HARC_AMD_FEED_MMAP")) == 0);
        const double t_ring0 = mono_now();
        if (c->feed_ring_bytes < SL * (size_t)NS) {
            if (c->feed_ring) { (void)hipHostFree(c->feed_ring); c->feed_ring = nullptr; c->feed_ring_bytes = 0; }
            if (hipHostMalloc((void **)&c->feed_ring, SL * (size_t)NS) != hipSuccess) { harc_set_error("hipHostMalloc of the ingest ring (%zu bytes) failed", SL * (size_t)NS); return HARC_AMD_ENOMEM; }
            c->feed_ring_bytes = SL * (size_t)NS;
        }
        t_ring = mono_now() - t_ring0;
        ev.assign(NS, nullptr);
        for (int k = 0; k < NS; k++) { if (hipEventCreate(&ev[k]) != hipSuccess) { harc_set_error("hipEventCreate failed"); return HARC_AMD_ENODEVICE; } free_slices.push_back(k); }
        piece_left.assign(pieces.size(), 0);
        for (size_t p = 0; p < pieces.size(); p++)
            for (uint64_t a = pieces[p].first; a < pieces[p].second; a += SL) {
                const uint64_t b = pieces[p].second - a < SL ? pieces[p].second : a + SL;
                chunks.push_back(Chunk{ a, (uint32_t)(b - a), (uint32_t)p, a - pieces[p].first });
                piece_left[p]++;
            }
        if ((size_t)nthr > chunks.size()) nthr = (int)chunks.size();
        for (int t = 0; t < nthr; t++) th.emplace_back([this] { reader(); });
        return HARC_AMD_OK;
    }
    void reader()
    {
        for (;;) {
            int sl; size_t k;
            const double tw0 = mono_now();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_free.wait(lk, [&] { return stop || next_chunk >= chunks.size() || !free_slices.empty(); });
                if (stop || next_chunk >= chunks.size()) return;
                sl = free_slices.front(); free_slices.pop_front(); k = next_chunk++;
            }
            const Chunk &ch = chunks[k];
            const double tr0 = mono_now();
            char *dst = c->feed_ring + (size_t)sl * SL; size_t got = 0; int e = 0;
            if (use_mmap) {
                const uint64_t a0 = ch.off & ~(uint64_t)4095; const size_t mlen = (size_t)(ch.off + ch.len - a0);
                void *m = mmap(nullptr, mlen, PROT_READ, MAP_SHARED, fd, (off_t)a0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, mlen, MADV_SEQUENTIAL);
                    memcpy(dst, (const char *)m + (ch.off - a0), ch.len); got = ch.len;
                    munmap(m, mlen);
                }
            }
            while (got < ch.len) {
                const ssize_t r = pread(fd, dst + got, ch.len - got, (off_t)(ch.off + got));
                if (r <= 0) { e = 1; break; }
                got += (size_t)r;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (e) { err = 1; stop = true; }
                filled.emplace_back(sl, k);
                t_wait_free += tr0 - tw0; t_pread += mono_now() - tr0;
            }
            cv_filled.notify_one();
            if (e) { cv_free.notify_all(); return; }
        }
    }
    void give_back(int sl) { { std::lock_guard<std::mutex> lk(mu); free_slices.push_back(sl); } cv_free.notify_one(); }
    // every chunk of piece p on the stream towards d_txt; chunks of piece p + 1 that are ready meanwhile go to d_next (may be null: they wait)
    int upload_piece(size_t p, char *d_txt, char *d_next)
    {
        auto put = [&](int sl, size_t k) -> int {
            const Chunk &ch = chunks[k];
            char *base = ch.piece == p ? d_txt : d_next;
            const double te0 = mono_now();
            if (hipMemcpyAsync(base + ch.at, c->feed_ring + (size_t)sl * SL, ch.len, hipMemcpyHostToDevice, c->stream) != hipSuccess) { harc_set_error("upload of %s failed", name); return HARC_AMD_ENODEVICE; }
            (void)hipEventRecord(ev[sl], c->stream);
            inflight.push_back(sl); piece_left[ch.piece]--;
            t_enqueue += mono_now() - te0;
            return HARC_AMD_OK;
        };
        // what was read ahead for this piece while the last one was uploaded
        for (size_t i = 0; i < held.size();) {
            const Chunk &ch = chunks[held[i].second];
            if (ch.piece == p || (ch.piece == p + 1 && d_next)) { RC_TRY(put(held[i].first, held[i].second)); held.erase(held.begin() + (long)i); } else i++;
        }
        while (piece_left[p] > 0) {
            while (!inflight.empty() && hipEventQuery(ev[inflight.front()]) == hipSuccess) { give_back(inflight.front()); inflight.pop_front(); }
            std::pair<int, size_t> it(-1, 0);
            {
                std::unique_lock<std::mutex> lk(mu);
                if (filled.empty() && !err) {
                    if (!inflight.empty()) { lk.unlock(); const double t0 = mono_now(); (void)hipEventSynchronize(ev[inflight.front()]); t_wait_copy += mono_now() - t0; give_back(inflight.front()); inflight.pop_front(); continue; }
                    const double t0 = mono_now();
                    cv_filled.wait(lk, [&] { return !filled.empty() || err; });
                    t_wait_filled += mono_now() - t0;
                }
                if (err) { harc_set_error("short read on %s", name); return HARC_AMD_EIO; }
                it = filled.front(); filled.pop_front();
            }
            const Chunk &ch = chunks[it.second];
            if (ch.piece == p || (ch.piece == p + 1 && d_next)) RC_TRY(put(it.first, it.second));
            else held.push_back(it);
        }
        return HARC_AMD_OK;
    }
};
// bytes [lo, hi) of the file -> device memory; *d_txt is a raw allocation of the context
static int load_file_range(harc_amd_ctx *c, FILE *f, const char *name, uint64_t lo, uint64_t hi, char **d_txt)
{
    (void)f;
    *d_txt = nullptr;
    const uint64_t n = hi - lo;
    RC_TRY(harc_raw_alloc(c, (void **)d_txt, (size_t)n + 16));
    int rc = HARC_AMD_OK;
    if (n) {
        FileFeeder fd(c, name);
        rc = fd.start({ { lo, hi } });
        if (rc == HARC_AMD_OK) rc = fd.upload_piece(0, *d_txt, nullptr);
    }
    if (rc == HARC_AMD_OK && hipStreamSynchronize(c->stream) != hipSuccess) { harc_set_error("upload of %s failed", name); rc = HARC_AMD_ENODEVICE; }
    if (rc != HARC_AMD_OK) { harc_raw_free(c, *d_txt); *d_txt = nullptr; }
    return rc;
}
static int spit_file(const std::string &path, const void *p, size_t n)
{
    FILE *o = fopen(path.c_str(), "wb");
    if (!o) { harc_set_error("cannot create %s", path.c_str()); return HARC_AMD_EIO; }
    if (n && fwrite(p, 1, n, o) != n) { fclose(o); harc_set_error("short write on %s", path.c_str()); return HARC_AMD_EIO; }
    fclose(o);
    return HARC_AMD_OK;
}
static int spit_stream_to(harc_amd_ctx *c, int id, int shard, const std::string &path)
{
    const void *p; size_t n;
    RC_TRY(harc_amd_get_stream(c, id, shard, &p, &n));
    return spit_file(path, p, n);
}
// read_{seq,pos,noise,noisepos,rev}.txt.<first_shard + e> (+ .tail): the per-shard family of encoder.cpp:190-196
// Stream by stream, the largest first: when HARC_AMD_READY_FD names an open descriptor (./harc passes a pipe), the name of every stream is
// written to it as soon as all its shard files are closed, and ./harc starts that stream's stage-III coder (harc:102-109) while the next
// stream is still being written.
static void announce_stream(const char *stem)
{
    const char *e = getenv("HARC_AMD_READY_FD");
    if (!e) return;
    const int fd = atoi(e);
    if (fd < 3) return;
    const std::string line = std::string(stem) + "\n";
    (void)!write(fd, line.data(), line.size());
}
// the stream files of the context's encoder shards [e_lo, e_hi) (default: all) as read_*.txt.<first_shard + e>
static int write_shard_family(harc_amd_ctx *c, const std::string &od, int first_shard, int e_lo = 0, int e_hi = -1)
{
    if (e_hi < 0) e_hi = c->P.num_thr;
    static const struct { int id; const char *name; int tail; } files[] = {
        { HARC_AMD_S2_SEQ, "read_seq", HARC_AMD_S2_SEQ_TAIL }, { HARC_AMD_S2_POS, "read_pos", -1 }, { HARC_AMD_S2_NOISE, "read_noise", -1 },
        { HARC_AMD_S2_NOISEPOS, "read_noisepos", -1 }, { HARC_AMD_S2_REV, "read_rev", HARC_AMD_S2_REV_TAIL } };
    for (auto &fd : files) {
        for (int e = e_lo; e < e_hi; e++) {
            const std::string path = od + fd.name + ".txt." + std::to_string(first_shard + e);
            RC_TRY(spit_stream_to(c, fd.id, e, path));
            if (fd.tail >= 0) RC_TRY(spit_stream_to(c, fd.tail, e, path + ".tail"));
        }
        if (first_shard == 0) announce_stream(fd.name);
    }
    return HARC_AMD_OK;
}

// first byte >= pos at which a FASTQ record starts: a line that begins with '@' whose second successor begins with '+'.  (A quality
// line may begin with '@' too, but then the line two further on is a sequence line, which never begins with '+'.)
static int record_start_at_or_after(FILE *f, uint64_t pos, uint64_t fsz, uint64_t *out)
{
    if (pos == 0) { *out = 0; return HARC_AMD_OK; }
    if (pos >= fsz) { *out = fsz; return HARC_AMD_OK; }
    std::vector<char> buf;
    for (size_t win = (size_t)1 << 16;; win <<= 1) {
        const uint64_t lo = pos - 1;                              // the byte before pos tells whether pos starts a line
        const uint64_t n = fsz - lo < win ? fsz - lo : win;
        buf.resize((size_t)n);
        if (fseeko(f, (off_t)lo, SEEK_SET) != 0 || fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { harc_set_error("cannot read the FASTQ file around byte %llu", (unsigned long long)pos); return HARC_AMD_EIO; }
        std::vector<size_t> ls;                                   // line starts inside the window (offsets into buf)
        for (size_t i = 0; i + 1 < (size_t)n; i++) if (buf[i] == '\n') ls.push_back(i + 1);
        for (size_t k = 0; k + 2 < ls.size(); k++)
            if (buf[ls[k]] == '@' && buf[ls[k + 2]] == '+') { *out = lo + ls[k]; return HARC_AMD_OK; }
        if (lo + n >= fsz) { *out = fsz; return HARC_AMD_OK; }   // no further record
    }
}
// The records of bytes [lo, end) of the file, a piece at a time: every piece goes to HBM, is indexed, classified and packed, then makes room
// for the next -- the file never has to fit next to the dictionaries.  With fq / fi (-q -p) the quality and id lines of every piece
// are written out in file order on the way.
static int ingest_file_range(harc_amd_ctx *c, FILE *f, const char *name, uint64_t lo, uint64_t end, uint64_t fsz, IngestState &st, FILE *fq, FILE *fi)
{
    uint64_t piece = (uint64_t)1 << 30;
    if (const char *e = getenv("HARC_AMD_INGEST_CHUNK")) { piece = strtoull(e, nullptr, 10); if (piece < 16) piece = 16; }    // tests: pieces of a few records
    const bool tlog = getenv("HARC_AMD_TRACE") != nullptr;
    double tl = mono_now();
    auto lap = [&](const char *what) { if (tlog) { const double t = mono_now(); fprintf(stderr, "[ingest] %s: %.3f s\n", what, t - tl); tl = t; } };
    RC_TRY(ingest_begin(c, st));
    lap("previous inputs dropped");
    // the pieces first (a piece ends where a record starts: a 64-KB look at the file per boundary), so that the readers can run ahead over all of them
    std::vector<std::pair<uint64_t, uint64_t>> pieces;
    uint64_t maxlen = 0;
    for (uint64_t a = lo; a < end;) {
        uint64_t hi = end;
        if (end - a > piece) { RC_TRY(record_start_at_or_after(f, a + piece, fsz, &hi)); if (hi > end) hi = end; if (hi <= a) hi = end; }
        pieces.emplace_back(a, hi); if (hi - a > maxlen) maxlen = hi - a;
        a = hi;
    }
    if (pieces.empty()) return ingest_finish(c, st);
    lap("piece boundaries");
    // two device buffers: piece i + 1 is uploaded (behind piece i's kernels on the stream) while the host still waits for piece i's counts
    struct Bufs { harc_amd_ctx *c; char *p[2] = { nullptr, nullptr }; ~Bufs() { for (char *x : p) if (x) harc_raw_free(c, x); } } db{ c };
    RC_TRY(harc_raw_alloc(c, (void **)&db.p[0], (size_t)maxlen + 16));
    if (pieces.size() > 1) RC_TRY(harc_raw_alloc(c, (void **)&db.p[1], (size_t)maxlen + 16));
    lap("two device buffers");
    {
    FileFeeder feed(c, name);
    RC_TRY(feed.start(pieces));
    lap("feeder started (file mapped, ring pinned, readers running)");
    const uint64_t start = lo;
    for (size_t p = 0; p < pieces.size(); p++) {
        const uint64_t a = pieces[p].first, hi = pieces[p].second;
        char *d_txt = db.p[p & 1];
        const double tu = mono_now();
        RC_TRY(feed.upload_piece(p, d_txt, p + 1 < pieces.size() ? db.p[(p + 1) & 1] : nullptr));
        const double ta = mono_now(); g_fastq_timing[2] += ta - tu;
        // the first piece tells how many reads the whole range will hold, give or take: the stores are sized once
        uint64_t expC = 0, expN = 0;
        if (a > start) { const double scale = 1.03 * (double)(end - start) / (double)(a - start); expC = (uint64_t)(scale * (double)st.nC); expN = (uint64_t)(scale * (double)st.nN); }
        RC_TRY(ingest_append(c, st, d_txt, hi - a, hi == fsz, expC, expN));
        g_fastq_timing[3] += mono_now() - ta;
        if (fq) RC_TRY(emit_q_fileorder(c, d_txt, hi - a, fq, fi));
    }
    lap("pieces uploaded, indexed, classified, packed");
    }
    lap("feeder gone (readers joined, file unmapped)");
    const int rc = ingest_finish(c, st);
    lap("ingest_finish");
    return rc;
}

// FASTQ file -> every stage-II file under <basedir>/output (+ read_order_N.bin, numreads.bin): harc:50-69 without input_clean.dna
extern "C" int harc_amd_compress_fastq_files_ex(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order, int32_t preserve_quality)
{
    if (!params || !fastq || !basedir) return HARC_AMD_EINVAL;
    for (double &x : g_fastq_timing) x = 0;
    const double t_begin = mono_now();
    harc_amd_ctx *c = nullptr;
    RC_TRY(harc_amd_create(params, &c));
    struct Guard { harc_amd_ctx *c; ~Guard() { harc_amd_destroy(c); } } guard{ c };
    g_fastq_timing[0] = mono_now() - t_begin;
    const double t_ingest = mono_now();
    FILE *f = fopen(fastq, "rb");
    if (!f) { harc_set_error("cannot open %s", fastq); return HARC_AMD_EIO; }
    struct FClose { FILE *f; ~FClose() { fclose(f); } } fcl{ f };
    fseeko(f, 0, SEEK_END); const uint64_t fsz = (uint64_t)ftello(f);
    const std::string od = std::string(basedir) + "/output/";
    char *d_txt = nullptr;                                        // -q without -p: the whole text stays in HBM until the orders are known
    struct FreeTxt { harc_amd_ctx *c; char **p; ~FreeTxt() { if (*p) harc_raw_free(c, *p); } } freetxt{ c, &d_txt };
    IngestState st;
    // -q without -p: the text stays in HBM until the orders are known when it fits next to everything else; a larger file is ingested in pieces
    // like any other and streamed again, once per bin of output, when the orders are there (emit_quality_and_ids_streamed)
    bool stream_q = false;
    if (preserve_quality && !preserve_order) {
        size_t fr = 0, tot = 0;
        stream_q = (hipMemGetInfo(&fr, &tot) == hipSuccess && (double)fsz > 0.45 * (double)fr);
        if (const char *e = getenv("HARC_AMD_Q_STREAM")) stream_q = atoi(e) != 0;       // tests: either way on a small file
    }
    if (preserve_quality && !preserve_order && !stream_q) {
        RC_TRY(load_file_range(c, f, fastq, 0, fsz, &d_txt));
        RC_TRY(ingest_begin(c, st));
        RC_TRY(ingest_append(c, st, d_txt, fsz, true, 0, 0));
        RC_TRY(ingest_finish(c, st));
    } else {
        FILE *fq = nullptr, *fi = nullptr;
        struct Closer { FILE **a, **b; ~Closer() { if (*a) fclose(*a); if (*b) fclose(*b); } } closer{ &fq, &fi };
        if (preserve_quality && preserve_order) {                  // file order: written while the pieces pass through
            fq = fopen((od + "output.quality").c_str(), "wb"); fi = fopen((od + "output.id").c_str(), "wb");
            if (!fq || !fi) { harc_set_error("cannot create %soutput.quality / output.id", od.c_str()); return HARC_AMD_EIO; }
        }
        st.want_idlen = stream_q;
        RC_TRY(ingest_file_range(c, f, fastq, 0, fsz, fsz, st, fq, fi));
    }
    g_fastq_timing[1] = mono_now() - t_ingest;
    printf("Read length: %d\nTotal number of reads: %llu\nTotal number of reads without N: %llu\nPreprocessing Done!\n", params->readlen,
           (unsigned long long)st.nfull, (unsigned long long)c->N);                                       // preprocess.cpp:133-136
    const bool tlog = getenv("HARC_AMD_TRACE") != nullptr;
    auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double tl0 = now();
    auto lap = [&](const char *what) { if (tlog) { const double t = now(); fprintf(stderr, "[compress_fastq] %s: %.3f s\n", what, t - tl0); tl0 = t; } };
    RC_TRY(spit_stream_to(c, HARC_AMD_IN_ORDER_N, 0, od + "read_order_N.bin"));
    { const uint32_t n32 = c->N; RC_TRY(spit_file(od + "numreads.bin", &n32, 4)); }
    double t_ph = mono_now();
    RC_TRY(harc_amd_reorder(c));
    lap("reorder");
    g_fastq_timing[4] = mono_now() - t_ph; t_ph = mono_now();
    RC_TRY(harc_amd_encode(c));
    lap("encode");
    g_fastq_timing[5] = mono_now() - t_ph; t_ph = mono_now();
    harc_amd_counters C; harc_amd_get_counters(c, &C);
    printf("Reordering done, %llu were unmatched\n", (unsigned long long)C.unmatched);
    printf("Encoding done:\n%llu singleton reads were aligned\n%llu reads with N were aligned\n", (unsigned long long)C.aligned_singletons, (unsigned long long)C.aligned_N);
    RC_TRY(write_shard_family(c, od, 0));
    lap("stream files");
    static const struct { int id; const char *name; } whole[] = {
        { HARC_AMD_S2_ORDER, "read_order.bin" }, { HARC_AMD_S2_ORDER_N_PE, "read_order_N_pe.bin" }, { HARC_AMD_S2_INPUT_N, "input_N.dna" },
        { HARC_AMD_S2_META, "read_meta.txt" }, { HARC_AMD_S2_SINGLETON, "read_singleton.txt" }, { HARC_AMD_S2_SINGLETON_TAIL, "read_singleton.txt.tail" } };
    for (auto &fd : whole) RC_TRY(spit_stream_to(c, fd.id, 0, od + fd.name));
    g_fastq_timing[6] = mono_now() - t_ph; g_fastq_timing[7] = mono_now() - t_begin;
    if (preserve_quality && !preserve_order) {
        printf("Reordering quality values and ids\n");                                                      // harc:122
        lap("whole-job files");
        if (stream_q) RC_TRY(emit_quality_and_ids_streamed(c, f, fastq, fsz, st, od, "output.quality", "output.id"));
        else RC_TRY(emit_quality_and_ids(c, d_txt, fsz, false, od, "output.quality", "output.id"));
        lap("quality values and ids");
    }
    return HARC_AMD_OK;
}
extern "C" int harc_amd_compress_fastq_files(const harc_amd_params *params, const char *fastq, const char *basedir)
{
    return harc_amd_compress_fastq_files_ex(params, fastq, basedir, 0, 0);
}

// ------------------------------------------------------------------------------------------------ one rank of a multi-GPU run
static int rendezvous(harc_amd_ctx *c, const char *spec, int world, int rank)
{
    if (!spec) { harc_set_error("comm_spec missing"); return HARC_AMD_EINVAL; }
    if (!strncmp(spec, "mailbox:", 8)) return harc_amd_comm_init_mailbox(c, spec + 8, world, rank);
    if (strncmp(spec, "rccl:", 5)) { harc_set_error("comm_spec must be rccl:<file> or mailbox:<dir>"); return HARC_AMD_EINVAL; }
    const std::string path = spec + 5;
    uint8_t id[HARC_AMD_COMM_ID_BYTES];
    if (rank == 0) {
        RC_TRY(harc_amd_comm_get_id(id, sizeof id));
        RC_TRY(spit_file(path + ".tmp", id, sizeof id));
        if (rename((path + ".tmp").c_str(), path.c_str()) != 0) { harc_set_error("cannot publish %s", path.c_str()); return HARC_AMD_EIO; }
    } else {
        FILE *f = nullptr;
        for (int tries = 0; !(f = fopen(path.c_str(), "rb")); tries++) {
            if (tries > 150000) { harc_set_error("rank %d: no communicator id at %s after 300 s", rank, path.c_str()); return HARC_AMD_EIO; }
            usleep(2000);
        }
        const bool ok = fread(id, 1, sizeof id, f) == sizeof id;
        fclose(f);
        if (!ok) { harc_set_error("short communicator id in %s", path.c_str()); return HARC_AMD_EIO; }
    }
    return harc_amd_comm_init(c, id, sizeof id, world, rank);
}

static int compress_fastq_rank(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                               int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec, bool replicate);
extern "C" int harc_amd_compress_fastq_shard_files(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                                                   int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec)
{
    return compress_fastq_rank(params, fastq, basedir, preserve_order, preserve_quality, world, rank, comm_spec, false);
}
// design (R): every rank ingests its slice, the reads are all-gathered, the chains partitioned; every rank ends with the streams of the whole
// job -- ONE GPU's streams, byte for byte -- and rank 0 writes them (as shard family 0 .. num_thr - 1, like a single-GPU run); what depends on
// the slice (read_order_N.bin, quality values and ids in file order) is left as rank parts.  harc_amd_merge_shard_files finishes the archive.
extern "C" int harc_amd_compress_fastq_replicated_files(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                                                        int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec)
{
    return compress_fastq_rank(params, fastq, basedir, preserve_order, preserve_quality, world, rank, comm_spec, true);
}
static int compress_fastq_rank(const harc_amd_params *params, const char *fastq, const char *basedir, int32_t preserve_order,
                               int32_t preserve_quality, int32_t world, int32_t rank, const char *comm_spec, bool replicate)
{
    if (!params || !fastq || !basedir || world < 1 || rank < 0 || rank >= world) { harc_set_error("compress_fastq_shard: bad arguments"); return HARC_AMD_EINVAL; }
    if (preserve_quality && !preserve_order) { harc_set_error("multi-GPU -q needs -p (quality values and ids stay in file order)"); return HARC_AMD_EINVAL; }
    harc_amd_ctx *c = nullptr;
    RC_TRY(harc_amd_create(params, &c));
    struct Guard { harc_amd_ctx *c; ~Guard() { harc_amd_destroy(c); } } guard{ c };
    FILE *f = fopen(fastq, "rb");
    if (!f) { harc_set_error("cannot open %s", fastq); return HARC_AMD_EIO; }
    struct FClose { FILE *f; ~FClose() { fclose(f); } } fcl{ f };
    fseeko(f, 0, SEEK_END); const uint64_t fsz = (uint64_t)ftello(f);
    uint64_t lo = 0, hi = fsz;
    RC_TRY(record_start_at_or_after(f, fsz / (uint64_t)world * (uint64_t)rank, fsz, &lo));
    if (rank + 1 < world) RC_TRY(record_start_at_or_after(f, fsz / (uint64_t)world * (uint64_t)(rank + 1), fsz, &hi));
    const std::string od = std::string(basedir) + "/output/", sd = od + ".shard/", r = "." + std::to_string(rank);
    (void)mkdir(sd.c_str(), 0777);                                // every rank tries; the first one wins
    IngestState st;
    {
        FILE *fq = nullptr, *fi = nullptr;
        struct Closer { FILE **a, **b; ~Closer() { if (*a) fclose(*a); if (*b) fclose(*b); } } closer{ &fq, &fi };
        if (preserve_quality) {                                   // file order (preprocess.cpp:64-69): the slices are concatenated by the merge
            fq = fopen((sd + "quality" + r).c_str(), "wb"); fi = fopen((sd + "id" + r).c_str(), "wb");
            if (!fq || !fi) { harc_set_error("cannot create the quality / id parts under %s", sd.c_str()); return HARC_AMD_EIO; }
        }
        RC_TRY(ingest_file_range(c, f, fastq, lo, hi, fsz, st, fq, fi));
    }
    const uint64_t nrec_full = st.nfull;
    const uint64_t n_clean_own = c->N_own;
    std::vector<uint32_t> orderN = st.orderN;                     // read_order_N.bin of this slice, local record numbers
    RC_TRY(rendezvous(c, comm_spec, world, rank));
    uint64_t info[8];
    if (replicate) RC_TRY(harc_amd_replicate_exchange(c, info)); else RC_TRY(harc_amd_shard_exchange(c, info));
    for (auto &x : orderN) x += (uint32_t)info[5];                // records of the lower ranks come first in the file
    RC_TRY(spit_file(sd + "order_N" + r, orderN.data(), orderN.size() * 4));
    RC_TRY(harc_amd_reorder(c));
    RC_TRY(harc_amd_encode(c));
    harc_amd_counters C; harc_amd_get_counters(c, &C);
    if (replicate && rank != 0 && !c->s2_part) {                  // (HARC_AMD_S2_PART=0) the same streams as rank 0 holds: nothing to write but empty parts
        static const char *stems[] = { "order_a", "order_u", "orderN_a", "orderN_u", "singleton", "singleton_tail", "input_N" };
        for (const char *st0 : stems) RC_TRY(spit_file(sd + st0 + r, "", 0));
        char line[256];
        const int n = snprintf(line, sizeof line, "%d %llu %llu %llu 0 0 0 %llu %llu\n", params->readlen, (unsigned long long)nrec_full,
                               (unsigned long long)n_clean_own, (unsigned long long)c->NN_own, (unsigned long long)c->N, (unsigned long long)c->NN);
        RC_TRY(spit_file(sd + "stats" + r, line, (size_t)n));
        RC_TRY(harc_amd_comm_barrier(c));
        return HARC_AMD_OK;
    }
    // design (R) with stage II partitioned: every rank writes the stream files of ITS encoder shards under their single-GPU names, and its parts of
    // the whole-job files (aligned part: its shards; unaligned part: its share of the candidates) -- the merge below is the bucket mode's
    if (replicate && c->s2_part) {
        RC_TRY(write_shard_family(c, od, 0, c->s2_e0, c->s2_e1));
        if (rank != 0) C.unmatched = 0;                           // stage I is the whole job's on every rank: counted once
    } else
    RC_TRY(write_shard_family(c, od, replicate ? 0 : rank * params->num_thr));
    {   // whole-job files: this rank's part, aligned and unaligned halves apart (the merge interleaves them as encoder.cpp:457-503 does)
        const void *po, *pn, *ps, *pt, *pi; size_t no, nn, ns, nt, ni;
        RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_ORDER, 0, &po, &no)); RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_ORDER_N_PE, 0, &pn, &nn));
        RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_SINGLETON, 0, &ps, &ns)); RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_SINGLETON_TAIL, 0, &pt, &nt));
        RC_TRY(harc_amd_get_stream(c, HARC_AMD_S2_INPUT_N, 0, &pi, &ni));
        const int L = params->readlen;
        const size_t US = (4 * ns + nt) / (size_t)L, UN = ni / (size_t)(L + 1);
        if (no < US * 4 || nn < UN * 4) { harc_set_error("shard %d: order streams shorter than the unaligned reads", rank); return HARC_AMD_ESTATE; }
        RC_TRY(spit_file(sd + "order_a" + r, po, no - US * 4)); RC_TRY(spit_file(sd + "order_u" + r, (const char *)po + (no - US * 4), US * 4));
        RC_TRY(spit_file(sd + "orderN_a" + r, pn, nn - UN * 4)); RC_TRY(spit_file(sd + "orderN_u" + r, (const char *)pn + (nn - UN * 4), UN * 4));
        RC_TRY(spit_file(sd + "singleton" + r, ps, ns)); RC_TRY(spit_file(sd + "singleton_tail" + r, pt, nt));
        RC_TRY(spit_file(sd + "input_N" + r, pi, ni));
    }
    {
        char line[512];
        const int n = snprintf(line, sizeof line, "%d %llu %llu %llu %llu %llu %llu %llu %llu\n", params->readlen, (unsigned long long)nrec_full,
                               (unsigned long long)n_clean_own, (unsigned long long)c->NN_own, (unsigned long long)C.unmatched,
                               (unsigned long long)C.aligned_singletons, (unsigned long long)C.aligned_N, (unsigned long long)c->N, (unsigned long long)c->NN);
        RC_TRY(spit_file(sd + "stats" + r, line, (size_t)n));       // written last: its presence says that the rank is done
    }
    RC_TRY(harc_amd_comm_barrier(c));                             // nobody leaves (and tears the communicator down) while a peer is still exchanging
    return HARC_AMD_OK;
}
