// api.cpp -- the C-ABI of include/harc_amd.h: context, inputs, outputs.  Compiled with hipcc (host code only).
#include "internal.h"
#include <stdarg.h>
#include <chrono>
#include <stdlib.h>

static thread_local char g_err[512] = "";
void harc_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char *harc_amd_last_error(void) { return g_err; }

// The empty chunks of the pool behind chunk `from` go back to the device (stack discipline: every chunk behind the one in use is empty).  Their places
// in the list stay, as chunks of no size, so that marks keep their meaning.  For the case that the device is full of chunks too small for what is
// asked for now (a first pass over an input whose later buffers are larger than its earlier ones: configs[3] with repeats asked for 16 GB with 119 GB
// of the pool's 267 GB idle in 8.6-GB chunks).  Returns the bytes handed back.
static size_t pool_hand_back(harc_amd_ctx *c, size_t from, size_t asked)
{
    size_t given = 0;
    for (size_t i = 0; i < c->pool.size(); i++) {                  // (a chunk below the one in use that holds nothing was too small for what came after it)
        harc_amd_ctx::PoolChunk &k = c->pool[i];
        if (i == from || k.size == 0 || k.used != 0) continue;
        (void)hipFree(k.base); c->pool_total -= k.size; given += k.size; k.base = nullptr; k.size = 0;
    }
    if (given) {
        (void)hipGetLastError();
        if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[pool] %.2f GB of empty chunks handed back for a request of %.2f GB\n", given / 1e9, asked / 1e9);
    }
    return given;
}
int harc_raw_alloc(harc_amd_ctx *c, void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess && !c->pool.empty() && pool_hand_back(c, c->pool_cur, bytes)) e = hipMalloc(p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); harc_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return HARC_AMD_ENOMEM; }
    c->owned.push_back(*p);
    c->sizes[*p] = bytes;
    c->dev_bytes += bytes;
    if (c->dev_bytes > c->dev_peak) c->dev_peak = c->dev_bytes;
    return HARC_AMD_OK;
}
void harc_raw_free(harc_amd_ctx *c, void *p)
{
    if (!p) return;
    for (size_t i = 0; i < c->owned.size(); i++)
        if (c->owned[i] == p) { c->owned[i] = c->owned.back(); c->owned.pop_back(); break; }
    auto it = c->sizes.find(p);
    if (it != c->sizes.end()) { c->dev_bytes -= it->second; c->sizes.erase(it); }
    (void)hipFree(p);
}
static size_t pool_in_use(const harc_amd_ctx *c)
{
    size_t u = 0;
    for (size_t i = 0; i <= c->pool_cur && i < c->pool.size(); i++) u += c->pool[i].used;
    return u;
}
int harc_dev_alloc(harc_amd_ctx *c, void **p, size_t bytes)
{
    *p = nullptr;
    if (const char *e = getenv("HARC_AMD_FAIL_ALLOC")) {           // tests: the n-th pool allocation of the process fails (error paths must leave the context usable)
        static long long countdown = -1;
        if (countdown < 0) countdown = atoll(e);
        if (countdown > 0 && --countdown == 0) { harc_set_error("pool allocation refused (HARC_AMD_FAIL_ALLOC)"); return HARC_AMD_ENOMEM; }
    }
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    const size_t cur_in = c->pool_cur;                             // stack discipline: every chunk behind this one is empty
    for (;;) {
        if (c->pool_cur < c->pool.size()) {
            harc_amd_ctx::PoolChunk &k = c->pool[c->pool_cur];
            if (k.used + bytes <= k.size) { *p = k.base + k.used; k.used += bytes; break; }
            if (c->pool_cur + 1 < c->pool.size()) { c->pool_cur++; c->pool[c->pool_cur].used = 0; continue; }
        }
        size_t want = bytes;
        const size_t grow = c->pool_total < ((size_t)8 << 30) ? c->pool_total : ((size_t)8 << 30);
        // small requests share large chunks; a request of more than half a chunk gets one of its own size -- an 8.6-GB chunk per 5-GB buffer left 40 % of
        // the pool idle behind the buffers (configs[3] with repeats: 267 GB of chunks for 148 GB in use, and the next 16 GB did not exist)
        if (want < grow / 2) want = grow;
        if (want < ((size_t)64 << 20)) want = (size_t)64 << 20;
        void *base = nullptr;
        hipError_t e = hipMalloc(&base, want);
        if (e != hipSuccess) (void)hipGetLastError();              // (a failed attempt that is recovered from below must not be what a later hipGetLastError() reports)
        if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc(&base, want); if (e != hipSuccess) (void)hipGetLastError(); }
        if (e != hipSuccess && !c->pool.empty() && pool_hand_back(c, cur_in, bytes)) e = hipMalloc(&base, want);
        if (e != hipSuccess) { (void)hipGetLastError(); harc_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return HARC_AMD_ENOMEM; }
        if (getenv("HARC_AMD_POISON")) (void)hipMemset(base, 0xA5, want);   // debugging aid: make reads of uninitialised pool memory show
        if (getenv("HARC_AMD_TRACE")) fprintf(stderr, "[pool] chunk %zu: %.2f GB for a request of %.2f GB (pool %.2f GB, in use %.2f GB)\n", c->pool.size(), want / 1e9, bytes / 1e9, (c->pool_total + want) / 1e9, pool_in_use(c) / 1e9);
        c->pool.push_back({ (char *)base, want, 0 });
        c->pool_total += want;
        c->pool_cur = c->pool.size() - 1;
    }
    const size_t live = c->dev_bytes + pool_in_use(c);
    if (live > c->dev_peak) c->dev_peak = live;
    return HARC_AMD_OK;
}
void harc_dev_free(harc_amd_ctx *c, void *p) { (void)c; (void)p; }   // pool memory is released by mark, not by pointer
harc_mark_t harc_pool_mark(harc_amd_ctx *c)
{
    if (c->pool.empty()) return 0;
    return ((harc_mark_t)c->pool_cur << 48) | (harc_mark_t)c->pool[c->pool_cur].used;
}
void harc_pool_release(harc_amd_ctx *c, harc_mark_t m)
{
    if (c->pool.empty()) return;
    const size_t cur = (size_t)(m >> 48), used = (size_t)(m & (((harc_mark_t)1 << 48) - 1));
    for (size_t i = cur + 1; i < c->pool.size(); i++) c->pool[i].used = 0;
    c->pool[cur].used = used;
    c->pool_cur = cur;
}
int harc_host_alloc(harc_amd_ctx *c, void **p, size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes == 0) bytes = 64;
    for (auto &k : c->harena) if (k.used + bytes <= k.size) { *p = k.base + k.used; k.used += bytes; return HARC_AMD_OK; }
    size_t want = bytes < ((size_t)16 << 20) ? ((size_t)16 << 20) : bytes + (bytes >> 2);
    void *base = nullptr;
    hipError_t e = hipHostMalloc(&base, want, hipHostMallocDefault);
    if (e != hipSuccess) { harc_set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return HARC_AMD_ENOMEM; }
    c->harena.push_back({ (char *)base, want, bytes });
    *p = base;
    return HARC_AMD_OK;
}
void harc_host_reset(harc_amd_ctx *c) { for (auto &k : c->harena) k.used = 0; }
int harc_d2h(harc_amd_ctx *c, std::vector<uint8_t> &dst, const void *d_src, size_t bytes)
{
    dst.resize(bytes);
    if (bytes) HIP_TRY(hipMemcpyAsync(dst.data(), d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    return HARC_AMD_OK;
}

extern "C" int harc_amd_default_params(int32_t L, harc_amd_params *p)
{
    if (!p || L < 1 || L > 255) { harc_set_error("readlen %d out of range 1..255 (harc:46-49; 256 breaks the pos byte, decoder.cpp:93)", L); return HARC_AMD_EINVAL; }
    memset(p, 0, sizeof *p);
    p->readlen = L; p->num_thr = 8; p->num_chains = 0;
    p->maxmatch = L / 2; p->thresh = 4; p->thresh_s = 24; p->maxsearch = 1000;          // harc:52-56
    const int h = L / 2;
    if (L > 100) { p->dict_start[0] = h - 32; p->dict_end[0] = h - 1; p->dict_start[1] = h; p->dict_end[1] = h - 1 + 32; }   // harc:57-60
    else { p->dict_start[0] = h - L * 32 / 100; p->dict_end[0] = h - 1; p->dict_start[1] = h; p->dict_end[1] = h - 1 + L * 32 / 100; }
    return HARC_AMD_OK;
}

static int check_params(const harc_amd_params *p)
{
    if (!p) return HARC_AMD_EINVAL;
    const int L = p->readlen;
    if (L < 1 || L > 255) { harc_set_error("readlen %d out of range 1..255", L); return HARC_AMD_EINVAL; }
    if (p->num_thr < 1 || p->num_thr > 4096) { harc_set_error("num_thr %d out of range", p->num_thr); return HARC_AMD_EINVAL; }
    if (p->num_chains < 0) { harc_set_error("num_chains < 0"); return HARC_AMD_EINVAL; }
    if (p->num_steps < 0 || p->num_steps > 64) { harc_set_error("num_steps %d out of range 0..64", p->num_steps); return HARC_AMD_EINVAL; }
    if (p->maxmatch < 0 || p->maxmatch > 128 || p->maxmatch > L) { harc_set_error("maxmatch out of range"); return HARC_AMD_EINVAL; }
    if (p->maxsearch < 1 || p->thresh < 0 || p->thresh_s < 0) { harc_set_error("thresholds out of range"); return HARC_AMD_EINVAL; }
    for (int l = 0; l < 2; l++) {
        const int n = p->dict_end[l] - p->dict_start[l] + 1;
        if (p->dict_start[l] < 0 || p->dict_end[l] >= L || n < 0 || n > 32) { harc_set_error("dictionary window %d out of range", l); return HARC_AMD_EINVAL; }
    }
    return HARC_AMD_OK;
}

extern "C" int harc_amd_create(const harc_amd_params *params, harc_amd_ctx **out)
{
    if (!out) return HARC_AMD_EINVAL;
    *out = nullptr;
    RC_TRY(check_params(params));
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { harc_set_error("no HIP device (%s); libharc_amd has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e)); return HARC_AMD_ENODEVICE; }
    if (params->device < 0 || params->device >= ndev) { harc_set_error("device %d not in [0,%d)", params->device, ndev); return HARC_AMD_EINVAL; }
    HIP_TRY(hipSetDevice(params->device));
    harc_amd_ctx *c = new harc_amd_ctx();
    c->P = *params;
    c->W = (2 * params->readlen + 63) / 64;
    c->W3 = (3 * params->readlen + 63) / 64;
    memset(&c->C, 0, sizeof c->C);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; harc_set_error("hipStreamCreate failed"); return HARC_AMD_ENODEVICE; }
    // (Under rocprofv3 the device -> host copies of the copy stream show up as kernels of the runtime's, __amd_rocclr_copyBuffer, and the kernels of the main
    // stream beside them as stretched to the copy's length -- configs[3]: k_acc_flags 30 ms beside a 1.4 GB copy.  Outside the profiler the copies do not
    // take compute units: a copy stream held to 8 / 16 / 32 units by a CU mask changed nothing, profiles/r05.)
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess) {
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        (void)hipStreamDestroy(c->stream); delete c; harc_set_error("hipStreamCreate failed"); return HARC_AMD_ENODEVICE;
    }
    *out = c;
    return HARC_AMD_OK;
}

extern "C" void harc_amd_destroy(harc_amd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->P.device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); }
    delete c->comm; c->comm = nullptr;
    for (void *p : c->owned) (void)hipFree(p);
    for (auto &k : c->pool) (void)hipFree(k.base);
    for (auto &k : c->harena) (void)hipHostFree(k.base);
    if (c->feed_ring) (void)hipHostFree(c->feed_ring);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    delete c;
}

void harc_drop_results(harc_amd_ctx *c)
{
    harc_pool_release(c, 0);                                     // every per-run buffer lives in the pool
    c->d_order = nullptr; c->d_flag = c->d_pos = c->d_rc = nullptr; c->d_order_s = nullptr; c->d_oreads = nullptr; c->d_sreads = nullptr;
    c->have_s1 = c->have_s2 = c->s1_from_files = false; c->M = c->S = 0;
    c->d_s2_order = c->d_s2_orderN = nullptr; c->n_s2_order = c->n_s2_orderN = 0;
    // results go; what came with the inputs (HARC_AMD_IN_ORDER_N, owned bytes) stays until the inputs are replaced
    for (auto it = c->out.begin(); it != c->out.end();) { if (it->first.first < HARC_AMD_IN_ORDER_N) it = c->out.erase(it); else ++it; }
    harc_host_reset(c);
}

int harc_in_reserve(harc_amd_ctx *c, harc_amd_ctx::InBuf *b, size_t bytes)
{
    if (b->p && b->cap >= bytes) return HARC_AMD_OK;
    if (b->p) { harc_raw_free(c, b->p); b->p = nullptr; b->cap = 0; }
    RC_TRY(harc_raw_alloc(c, &b->p, bytes));
    b->cap = bytes;
    return HARC_AMD_OK;
}
void harc_reset_shard(harc_amd_ctx *c)
{
    c->d_reads = (uint64_t *)c->own_reads.p; c->N = c->N_own;
    c->d_nreads3 = (uint64_t *)c->own_nreads3.p; c->NN = c->NN_own;
    c->d_gid = c->d_ngid = nullptr;
    c->replicated = false;
    memset(c->shard_info, 0, sizeof c->shard_info);
    c->C.n_clean = c->N; c->C.n_N = c->NN;
}

static int upload(harc_amd_ctx *c, const void *host, size_t bytes, char **d)
{
    RC_TRY(harc_raw_alloc(c, (void **)d, bytes + 16));
    if (bytes) HIP_TRY(hipMemcpyAsync(*d, host, bytes, hipMemcpyHostToDevice, c->stream));
    return HARC_AMD_OK;
}

extern "C" int harc_amd_set_reads_ascii_device(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride)
{
    if (!c || (n && !d_ascii) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    harc_drop_results(c);
    c->out.erase(std::make_pair((int)HARC_AMD_IN_ORDER_N, 0));
    RC_TRY(harc_in_reserve(c, &c->own_reads, ((size_t)n * c->W + 1) * 8));
    c->N_own = n; c->nrec_own = (uint64_t)n + c->NN_own;
    harc_reset_shard(c);
    RC_TRY(s1_pack_ascii(c, d_ascii, n, stride, c->d_reads));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}
extern "C" int harc_amd_set_reads_ascii(harc_amd_ctx *c, const char *ascii, uint32_t n, uint32_t stride)
{
    if (!c || (n && !ascii) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    char *d = nullptr;
    const size_t bytes = n ? (size_t)(n - 1) * stride + c->P.readlen : 0;
    RC_TRY(upload(c, ascii, bytes, &d));
    int r = harc_amd_set_reads_ascii_device(c, d, n, stride);
    harc_raw_free(c, d);
    return r;
}
extern "C" int harc_amd_set_reads_packed_device(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n)
{
    if (!c || (n && !d_packed)) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    harc_drop_results(c);
    c->out.erase(std::make_pair((int)HARC_AMD_IN_ORDER_N, 0));
    RC_TRY(harc_in_reserve(c, &c->own_reads, ((size_t)n * c->W + 1) * 8));
    c->N_own = n; c->nrec_own = (uint64_t)n + c->NN_own;
    harc_reset_shard(c);
    if (n) HIP_TRY(hipMemcpyAsync(c->d_reads, d_packed, (size_t)n * c->W * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}
extern "C" int harc_amd_set_nreads_ascii_device(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride)
{
    if (!c || (n && !d_ascii) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    c->have_s2 = false;
    RC_TRY(harc_in_reserve(c, &c->own_nreads3, ((size_t)n * c->W3 + 1) * 8));
    c->NN_own = n; c->nrec_own = (uint64_t)c->N_own + n;
    harc_reset_shard(c);
    RC_TRY(s1_pack3_ascii(c, d_ascii, n, stride, c->d_nreads3));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}
extern "C" int harc_amd_set_nreads_ascii(harc_amd_ctx *c, const char *ascii, uint32_t n, uint32_t stride)
{
    if (!c || (n && !ascii) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    char *d = nullptr;
    const size_t bytes = n ? (size_t)(n - 1) * stride + c->P.readlen : 0;
    RC_TRY(upload(c, ascii, bytes, &d));
    int r = harc_amd_set_nreads_ascii_device(c, d, n, stride);
    harc_raw_free(c, d);
    return r;
}

extern "C" int harc_amd_set_stage1_streams(harc_amd_ctx *c, const char *temp_dna, const uint8_t *flag, const uint8_t *pos,
                                           const uint32_t *order, const uint8_t *rc, uint32_t M,
                                           const char *temp_dna_s, const uint32_t *order_s, uint32_t S)
{
    if (!c || (M && (!temp_dna || !flag || !pos || !order || !rc)) || (S && (!temp_dna_s || !order_s))) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    harc_drop_results(c);
    const int L = c->P.readlen;
    c->M = M; c->S = S;
    RC_TRY(dalloc(c, &c->d_order, (size_t)M + 1)); RC_TRY(dalloc(c, &c->d_flag, (size_t)M + 1)); RC_TRY(dalloc(c, &c->d_pos, (size_t)M + 1));
    RC_TRY(dalloc(c, &c->d_rc, (size_t)M + 1)); RC_TRY(dalloc(c, &c->d_order_s, (size_t)S + 1));
    RC_TRY(dalloc(c, &c->d_oreads, (size_t)M * c->W + 1)); RC_TRY(dalloc(c, &c->d_sreads, (size_t)S * c->W + 1));
    if (M) {
        HIP_TRY(hipMemcpyAsync(c->d_order, order, (size_t)M * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_flag, flag, M, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_pos, pos, M, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_rc, rc, M, hipMemcpyHostToDevice, c->stream));
        char *d = nullptr; RC_TRY(upload(c, temp_dna, (size_t)(M - 1) * (L + 1) + L, &d));
        RC_TRY(s1_pack_ascii(c, d, M, (uint32_t)L + 1, c->d_oreads));
        HIP_TRY(hipStreamSynchronize(c->stream)); harc_raw_free(c, d);
    }
    if (S) {
        HIP_TRY(hipMemcpyAsync(c->d_order_s, order_s, (size_t)S * 4, hipMemcpyHostToDevice, c->stream));
        char *d = nullptr; RC_TRY(upload(c, temp_dna_s, (size_t)(S - 1) * (L + 1) + L, &d));
        RC_TRY(s1_pack_ascii(c, d, S, (uint32_t)L + 1, c->d_sreads));
        HIP_TRY(hipStreamSynchronize(c->stream)); harc_raw_free(c, d);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_s1 = true; c->s1_from_files = true;
    c->C.n_main = M; c->C.n_singleton = S;
    return HARC_AMD_OK;
}

extern "C" int harc_amd_pack_reads_device(harc_amd_ctx *c, const char *d_ascii, uint32_t n, uint32_t stride, uint64_t *d_out)
{
    if (!c || (n && (!d_ascii || !d_out)) || stride < (uint32_t)c->P.readlen) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    RC_TRY(s1_pack_ascii(c, d_ascii, n, stride, d_out));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}
extern "C" int harc_amd_partition_reads_device(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint64_t *d_out, uint64_t *d_counts)
{
    if (!c || nb == 0 || nb > 4096 || !d_counts || (n && (!d_packed || !d_out))) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    return s1_partition_reads(c, d_packed, n, nb, d_out, reinterpret_cast<unsigned long long *>(d_counts));
}
extern "C" int harc_amd_bucket_reads_device(harc_amd_ctx *c, const uint64_t *d_packed, uint32_t n, uint32_t nb, uint32_t *d_out)
{
    if (!c || nb == 0 || (n && (!d_packed || !d_out))) return HARC_AMD_EINVAL;
    HIP_TRY(hipSetDevice(c->P.device));
    RC_TRY(s1_bucket_reads(c, d_packed, n, nb, d_out));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HARC_AMD_OK;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" int harc_amd_reorder(harc_amd_ctx *c)
{
    if (!c) return HARC_AMD_EINVAL;
    if (!c->d_reads) { harc_set_error("harc_amd_reorder: no reads set"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    harc_drop_results(c);
    const double t0 = now_ms();
    RC_TRY(stage1_run(c));
    c->C.total_ms = now_ms() - t0;
    c->C.device_bytes_peak = c->dev_peak;
    return HARC_AMD_OK;
}

extern "C" int harc_amd_encode(harc_amd_ctx *c)
{
    if (!c) return HARC_AMD_EINVAL;
    if (!c->have_s1) { harc_set_error("harc_amd_encode: no stage-I result (call harc_amd_reorder or harc_amd_set_stage1_streams)"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    const double t0 = now_ms();
    // design (R) on more than one rank: stage II is partitioned by encoder shard (stage2_run's header); HARC_AMD_S2_PART=0 keeps it replicated --
    // every rank then ends with ALL streams of the job, as a single GPU does (tests compare the two)
    c->s2_part = false; c->s2_e0 = 0; c->s2_e1 = 0;
    uint32_t oi0 = 0, oi1 = 0xFFFFFFFFu;
    int simr = 0, simw = 0;
#ifdef HARC_AMD_EXPERIMENTS
    // HARC_AMD_S2_SIM=rank/world (profiling only, `make EXPERIMENTS=1` builds only: the streams it leaves are NOT an archive): what ONE rank of a
    // partitioned stage II computes, without peers -- no all-reduce, so the claims of the other ranks' columns are missing; tools/s2_share.sh times
    // configs[3]'s share with it
    if (const char *e = getenv("HARC_AMD_S2_SIM")) { if (sscanf(e, "%d/%d", &simr, &simw) != 2 || simw < 2 || simr < 0 || simr >= simw) simw = 0; }
#endif
    // HARC_AMD_S2_PART: 1 (default) = partitioned from two ranks on; 0 = replicated; 2 = partitioned even at world 1 (tests: the all-reduce(min) of the
    // claims and the exchange of the large-bin events then run over RCCL on a one-GPU box; the rank holds every shard, the streams are the single GPU's)
    const int s2p = getenv("HARC_AMD_S2_PART") ? atoi(getenv("HARC_AMD_S2_PART")) : 1;
    const bool real_part = c->replicated && c->comm && (c->comm->world > 1 || s2p == 2) && s2p != 0;
    if ((real_part || simw) && !c->s1_from_files) {
        const uint32_t E = (uint32_t)c->P.num_thr, wd = real_part ? (uint32_t)c->comm->world : (uint32_t)simw, rk = real_part ? (uint32_t)c->comm->rank : (uint32_t)simr;
        c->s2_world = (int)wd; c->s2_rank = (int)rk;
        c->s2_part = true; c->s2_e0 = (int)((uint64_t)E * rk / wd); c->s2_e1 = (int)((uint64_t)E * (rk + 1) / wd);
        uint32_t q = 1u + (uint32_t)((c->M - 1u) / E); if (q == 0) q = 1;          // encoder.cpp:171
        const uint64_t a0 = (uint64_t)c->s2_e0 * q, a1 = (uint64_t)c->s2_e1 * q;
        oi0 = (uint32_t)(a0 > c->M ? c->M : a0); oi1 = (uint32_t)((uint32_t)c->s2_e1 >= E || a1 > c->M ? c->M : a1);
    }
    if (!c->s1_from_files) RC_TRY(stage1_make_oriented(c, oi0, oi1));
    RC_TRY(stage2_run(c));
    c->C.encode_ms = now_ms() - t0;
    c->C.total_ms += c->C.encode_ms;
    c->C.device_bytes_peak = c->dev_peak;
    c->have_s2 = true;
    return HARC_AMD_OK;
}

extern "C" int harc_amd_pack_order(harc_amd_ctx *c)
{
    if (!c) return HARC_AMD_EINVAL;
    if (!c->have_s2) { harc_set_error("harc_amd_pack_order: encode first"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    return pack_order_run(c);
}

extern "C" int harc_amd_get_stream(harc_amd_ctx *c, int32_t id, int32_t shard, const void **ptr, size_t *len)
{
    if (!c || !ptr || !len) return HARC_AMD_EINVAL;
    *ptr = nullptr; *len = 0;
    HIP_TRY(hipSetDevice(c->P.device));
    auto key = std::make_pair((int)id, (int)shard);
    auto it = c->out.find(key);
    if (it == c->out.end()) {
        if (id >= HARC_AMD_S1_ORDER && id <= HARC_AMD_S1_DNA_SINGLETON) {
            if (!c->have_s1 || shard != 0) { harc_set_error("stream %d not available", id); return HARC_AMD_ESTATE; }
            std::vector<uint8_t> &b = out_buf(c, id, shard);
            const int L = c->P.readlen;
            bool made_oriented = false;
            switch (id) {
            case HARC_AMD_S1_ORDER: RC_TRY(harc_d2h(c, b, c->d_order, (size_t)c->M * 4)); break;
            case HARC_AMD_S1_FLAG: RC_TRY(harc_d2h(c, b, c->d_flag, c->M)); break;
            case HARC_AMD_S1_POS: RC_TRY(harc_d2h(c, b, c->d_pos, c->M)); break;
            case HARC_AMD_S1_RC: RC_TRY(harc_d2h(c, b, c->d_rc, c->M)); break;
            case HARC_AMD_S1_ORDER_SINGLETON: RC_TRY(harc_d2h(c, b, c->d_order_s, (size_t)c->S * 4)); break;
            case HARC_AMD_S1_DNA: {
                if (!c->d_oreads) { RC_TRY(stage1_make_oriented(c)); made_oriented = true; }
                const harc_mark_t mk = harc_pool_mark(c);
                char *d = nullptr; RC_TRY(dalloc(c, &d, (size_t)c->M * (L + 1) + 1));
                RC_TRY(s1_unpack_to_ascii(c, c->d_oreads, c->M, d));
                RC_TRY(harc_d2h(c, b, d, (size_t)c->M * (L + 1)));
                HIP_TRY(hipStreamSynchronize(c->stream)); harc_pool_release(c, mk); (void)made_oriented;
                break; }
            case HARC_AMD_S1_DNA_SINGLETON: {
                const harc_mark_t mk = harc_pool_mark(c);
                uint64_t *sr = c->d_sreads; bool tmp = false;
                if (!sr) { RC_TRY(dalloc(c, &sr, (size_t)c->S * c->W + 1)); tmp = true; RC_TRY(s1_orient(c, c->d_reads, c->d_order_s, nullptr, c->S, sr)); }
                char *d = nullptr; RC_TRY(dalloc(c, &d, (size_t)c->S * (L + 1) + 1));
                RC_TRY(s1_unpack_to_ascii(c, sr, c->S, d));
                RC_TRY(harc_d2h(c, b, d, (size_t)c->S * (L + 1)));
                HIP_TRY(hipStreamSynchronize(c->stream)); harc_pool_release(c, mk); (void)tmp;
                break; }
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            it = c->out.find(key);
        } else if ((id == HARC_AMD_S2_ORDER || id == HARC_AMD_S2_ORDER_N_PE) && shard == 0 && c->have_s2 && c->d_s2_order) {
            // the order streams of stage II wait in HBM until they are wanted
            const uint32_t *d = id == HARC_AMD_S2_ORDER ? c->d_s2_order : c->d_s2_orderN;
            const size_t n = id == HARC_AMD_S2_ORDER ? c->n_s2_order : c->n_s2_orderN;
            uint8_t *hp = nullptr;
            RC_TRY(harc_host_alloc(c, (void **)&hp, n));
            if (n) HIP_TRY(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            out_slice(c, id, shard, hp, n);
            it = c->out.find(key);
        } else { harc_set_error("stream %d shard %d not available", id, shard); return HARC_AMD_ESTATE; }
    }
    if (it->second.ptr) { *ptr = it->second.ptr; *len = it->second.len; }
    else { *ptr = it->second.own.data(); *len = it->second.own.size(); }
    return HARC_AMD_OK;
}

extern "C" int harc_amd_stream_digest(harc_amd_ctx *c, uint64_t out[4])
{
    if (!c || !out) return HARC_AMD_EINVAL;
    if (!c->have_s2 || !c->have_digest) { harc_set_error("harc_amd_stream_digest: no digest (params.stream_digest = 1, then harc_amd_encode)"); return HARC_AMD_ESTATE; }
    for (int k = 0; k < 4; k++) out[k] = c->digest[k];
    return HARC_AMD_OK;
}
#include "build_id.h"
extern "C" const char *harc_amd_build_id(void) { return HARC_AMD_BUILD_ID; }

extern "C" int harc_amd_build_has(const char *feature)
{
    if (!feature) return 0;
#ifdef HARC_AMD_WITH_GRP
    if (!strcmp(feature, "grp")) return 1;
#endif
#ifdef HARC_AMD_TEST_TRANSPORT
    if (!strcmp(feature, "test_transport")) return 1;
#endif
#ifdef HARC_AMD_EXPERIMENTS
    if (!strcmp(feature, "experiments")) return 1;
#endif
    return 0;
}
extern "C" int harc_amd_get_counters(harc_amd_ctx *c, harc_amd_counters *out)
{
    if (!c || !out) return HARC_AMD_EINVAL;
    *out = c->C;
    return HARC_AMD_OK;
}
