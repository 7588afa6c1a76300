// comm.cpp -- transports of the multi-GPU exchange (one process per GPU; BASELINE.json north_star: "reads shard by k-mer hash bucket
// across the 8 GPUs of one node with a single RCCL all-to-all over xGMI").
//
//   RcclComm     ncclCommInitRank on the context's device; the all-to-all(v) is ONE ncclGroupStart .. ncclGroupEnd of
//                ncclSend / ncclRecv pairs, one pair per (array, peer): point-to-point chunks, each over its own xGMI link, no ring.
//                librccl.so.1 is opened on first use (a 570 MB library that single-GPU runs never need; inside a torch process the
//                loader hands back the copy torch already mapped, same SONAME).
//   MailboxComm  every chunk goes through a file of a shared directory.  Test transport: lets two ranks that share ONE GPU (which
//                RCCL refuses: "duplicate GPU") run the whole sharded path on a one-GPU box.  Never chosen implicitly.
#include "internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <string>

// ------------------------------------------------------------------------------------------------ RCCL
namespace {
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;               // optional: used when a collective times out
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.h) return HARC_AMD_OK;
    void *h = nullptr;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { harc_set_error("cannot load librccl.so.1: %s", dlerror()); return HARC_AMD_ENODEVICE; }
#define SYM(field, name) do { g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
        if (!g_rccl.field) { harc_set_error("librccl: symbol %s missing", name); dlclose(h); return HARC_AMD_ENODEVICE; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl.h = h;
    return HARC_AMD_OK;
}
#define NCCL_TRY(expr)                                                                                             \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess) {                                                                                   \
            harc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));               \
            return HARC_AMD_ENODEVICE;                                                                             \
        }                                                                                                          \
    } while (0)

// seconds a collective may stay on the stream before the rank gives up on its peers (HARC_AMD_COMM_TIMEOUT; 0 = wait for ever)
static double comm_timeout_s()
{
    double t = 600.0;
    if (const char *e = getenv("HARC_AMD_COMM_TIMEOUT")) { char *end = nullptr; const double v = strtod(e, &end); if (end != e && v >= 0) t = v; }
    return t;
}
static double mono_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

struct RcclComm : HarcComm {
    ncclComm_t comm = nullptr;
    uint64_t *d_ag = nullptr; size_t ag_cap = 0;               // small device buffer for the count all-gather
    uint64_t *h_ag = nullptr;                                   // its pinned host twin: the results leave the device only AFTER wait() has seen the collective finish
    ~RcclComm() override { if (comm) g_rccl.CommDestroy(comm); if (d_ag) (void)hipFree(d_ag); if (h_ag) (void)hipHostFree(h_ag); }
    const char *name() const override { return "rccl"; }
    // A peer that died (or never came) leaves this rank's stream waiting inside the collective for ever: the stream is polled, and
    // when the collective has not finished after the timeout the communicator is aborted and the call fails with HARC_AMD_ETIMEOUT.
    // The context cannot be used for another exchange after that: the caller is expected to end the process.
    // Nothing that blocks the HOST may be enqueued behind a collective before wait() has returned: a device-to-host copy into pageable
    // memory is staged by the runtime and holds the calling thread until all earlier work of the stream has finished -- with a dead peer
    // the thread would sit in hipMemcpyAsync and never reach the poll below.  Results go to pinned memory, or are copied after wait().
    int wait(harc_amd_ctx *c, const char *what) override
    {
        const double lim = comm_timeout_s(), t0 = mono_now();
        for (;;) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) return HARC_AMD_OK;
            if (q != hipErrorNotReady) { harc_set_error("%s: %s", what, hipGetErrorString(q)); return HARC_AMD_ENODEVICE; }
            if (lim > 0 && mono_now() - t0 > lim) {
                harc_set_error("%s: no answer from the peers after %.0f s (rank %d of %d); communicator aborted", what, lim, rank, world);
                if (g_rccl.CommAbort && comm) { (void)g_rccl.CommAbort(comm); comm = nullptr; }
                return HARC_AMD_ETIMEOUT;
            }
            usleep(mono_now() - t0 < 0.05 ? 50 : 1000);
        }
    }
    int allgather_u64(harc_amd_ctx *c, const uint64_t *in, int n, uint64_t *out) override
    {
        const size_t need = (size_t)(world + 1) * n * 8;
        if (need > ag_cap) {
            if (d_ag) (void)hipFree(d_ag);
            if (h_ag) (void)hipHostFree(h_ag);
            d_ag = nullptr; h_ag = nullptr; ag_cap = 0;
            HIP_TRY(hipMalloc((void **)&d_ag, need)); HIP_TRY(hipHostMalloc((void **)&h_ag, need)); ag_cap = need;
        }
        uint64_t *d_in = d_ag, *d_out = d_ag + n;
        memcpy(h_ag, in, (size_t)n * 8);                          // pinned on both sides: neither copy blocks the host
        HIP_TRY(hipMemcpyAsync(d_in, h_ag, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(g_rccl.AllGather(d_in, d_out, (size_t)n, ncclUint64, comm, c->stream));
        HIP_TRY(hipMemcpyAsync(h_ag + n, d_out, (size_t)world * n * 8, hipMemcpyDeviceToHost, c->stream));
        RC_TRY(wait(c, "all-gather of the counts"));
        memcpy(out, h_ag + n, (size_t)world * n * 8);
        return HARC_AMD_OK;
    }
    int allreduce_min_u64(harc_amd_ctx *c, unsigned long long *d_buf, size_t n) override
    {
        // pieces of at most 2^28 elements (2 GB): one ring all-reduce each, in place
        for (size_t o = 0; o < n; o += (size_t)1 << 28) {
            const size_t m = n - o < ((size_t)1 << 28) ? n - o : ((size_t)1 << 28);
            NCCL_TRY(g_rccl.AllReduce(d_buf + o, d_buf + o, m, ncclUint64, ncclMin, comm, c->stream));
        }
        return HARC_AMD_OK;
    }
    int alltoallv(harc_amd_ctx *c, int narr, const void *const *send, const size_t *const *soff, const size_t *const *sbytes,
                  void *const *recv, const size_t *const *roff, const size_t *const *rbytes) override
    {
        // One group; every (array, peer) chunk goes as pieces of at most PIECE bytes -- sender and receiver cut the same byte count the same
        // way, and the pieces of a pair are matched in the order they are posted.  (With two ranks a chunk of configs[2] is 5.6 GB; a
        // single send of 10 GB to the rank itself, configs[4] at 1/16 on one GPU, never came back.)
        size_t PIECE = (size_t)1 << 30;
        if (const char *e = getenv("HARC_AMD_XCHG_PIECE")) {                          // every rank must use the same value; nonsense falls back to the default
            char *end = nullptr; const unsigned long long v = strtoull(e, &end, 10);
            if (end != e && v >= ((unsigned long long)1 << 20)) PIECE = (size_t)v;
        }
        NCCL_TRY(g_rccl.GroupStart());
        // an error between GroupStart and GroupEnd must not leave the group open: the first one is remembered, posting stops, the group is closed
        ncclResult_t first = ncclSuccess; const char *where = "";
        for (int a = 0; a < narr && first == ncclSuccess; a++)
            for (int p = 0; p < world && first == ncclSuccess; p++) {
                // zero-byte chunks are skipped on both sides: sender and receiver see the same count matrix
                for (size_t o = 0; o < sbytes[a][p] && first == ncclSuccess; o += PIECE) {
                    const size_t n = sbytes[a][p] - o < PIECE ? sbytes[a][p] - o : PIECE;
                    first = g_rccl.Send((const char *)send[a] + soff[a][p] + o, n, ncclUint8, p, comm, c->stream); where = "ncclSend";
                }
                for (size_t o = 0; o < rbytes[a][p] && first == ncclSuccess; o += PIECE) {
                    const size_t n = rbytes[a][p] - o < PIECE ? rbytes[a][p] - o : PIECE;
                    first = g_rccl.Recv((char *)recv[a] + roff[a][p] + o, n, ncclUint8, p, comm, c->stream); where = "ncclRecv";
                }
            }
        const ncclResult_t ge = g_rccl.GroupEnd();
        if (first != ncclSuccess) { harc_set_error("all-to-all: %s -> %s", where, g_rccl.GetErrorString(first)); return HARC_AMD_ENODEVICE; }
        if (ge != ncclSuccess) { harc_set_error("all-to-all: ncclGroupEnd -> %s", g_rccl.GetErrorString(ge)); return HARC_AMD_ENODEVICE; }
        return HARC_AMD_OK;                                       // enqueued on c->stream; wait() tells when (and whether) it has finished
    }
};

// ------------------------------------------------------------------------------------------------ mailbox (tests)
// Compiled only with -DHARC_AMD_TEST_TRANSPORT (the Makefile's default, TEST_TRANSPORT=1: the one-GPU tests of the sharded path need it);
// a production build (make TEST_TRANSPORT=0) has RCCL only and harc_amd_comm_init_mailbox answers HARC_AMD_ESTATE.
#ifdef HARC_AMD_TEST_TRANSPORT
struct MailboxComm : HarcComm {
    std::string dir;
    uint64_t seq = 0;
    double timeout_s = 300.0;
    const char *name() const override { return "mailbox"; }
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    std::string path(const char *kind, uint64_t s, int src, int dst, int arr) const
    {
        char b[128]; snprintf(b, sizeof b, "/%s.%llu.%d.%d.%d", kind, (unsigned long long)s, src, dst, arr);
        return dir + b;
    }
    int put(const std::string &p, const void *data, size_t n) const
    {
        const std::string tmp = p + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) { harc_set_error("mailbox: cannot create %s", tmp.c_str()); return HARC_AMD_EIO; }
        const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
        fclose(f);
        if (!ok || rename(tmp.c_str(), p.c_str()) != 0) { harc_set_error("mailbox: cannot write %s", p.c_str()); return HARC_AMD_EIO; }
        return HARC_AMD_OK;
    }
    int get(const std::string &p, void *data, size_t n) const
    {
        const double t0 = now();
        FILE *f = nullptr;
        while (!(f = fopen(p.c_str(), "rb"))) {
            if (now() - t0 > timeout_s) { harc_set_error("mailbox: timed out waiting for %s", p.c_str()); return HARC_AMD_EIO; }
            usleep(2000);
        }
        const bool ok = n == 0 || fread(data, 1, n, f) == n;
        fclose(f);
        if (!ok) { harc_set_error("mailbox: short read on %s", p.c_str()); return HARC_AMD_EIO; }
        return HARC_AMD_OK;
    }
    int allgather_u64(harc_amd_ctx *, const uint64_t *in, int n, uint64_t *out) override
    {
        const uint64_t s = seq++;
        RC_TRY(put(path("ag", s, rank, 0, 0), in, (size_t)n * 8));
        for (int p = 0; p < world; p++) RC_TRY(get(path("ag", s, p, 0, 0), out + (size_t)p * n, (size_t)n * 8));
        return HARC_AMD_OK;
    }
    int alltoallv(harc_amd_ctx *c, int narr, const void *const *send, const size_t *const *soff, const size_t *const *sbytes,
                  void *const *recv, const size_t *const *roff, const size_t *const *rbytes) override
    {
        const uint64_t s = seq++;
        std::vector<char> h;
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int a = 0; a < narr; a++)
            for (int p = 0; p < world; p++) {
                h.resize(sbytes[a][p]);
                if (sbytes[a][p]) HIP_TRY(hipMemcpy(h.data(), (const char *)send[a] + soff[a][p], sbytes[a][p], hipMemcpyDeviceToHost));
                RC_TRY(put(path("xx", s, rank, p, a), h.data(), h.size()));
            }
        for (int a = 0; a < narr; a++)
            for (int p = 0; p < world; p++) {
                h.resize(rbytes[a][p]);
                RC_TRY(get(path("xx", s, p, rank, a), h.data(), h.size()));
                if (rbytes[a][p]) HIP_TRY(hipMemcpy((char *)recv[a] + roff[a][p], h.data(), rbytes[a][p], hipMemcpyHostToDevice));
            }
        return HARC_AMD_OK;
    }
    int allreduce_min_u64(harc_amd_ctx *c, unsigned long long *d_buf, size_t n) override
    {
        const uint64_t s = seq++;
        std::vector<unsigned long long> mine(n), other(n);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (n) HIP_TRY(hipMemcpy(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost));
        RC_TRY(put(path("ar", s, rank, 0, 0), mine.data(), n * 8));
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            RC_TRY(get(path("ar", s, p, 0, 0), other.data(), n * 8));
            for (size_t i = 0; i < n; i++) if (other[i] < mine[i]) mine[i] = other[i];
        }
        if (n) HIP_TRY(hipMemcpy(d_buf, mine.data(), n * 8, hipMemcpyHostToDevice));
        return HARC_AMD_OK;
    }
    int wait(harc_amd_ctx *c, const char *) override { HIP_TRY(hipStreamSynchronize(c->stream)); return HARC_AMD_OK; }
};
#endif
} // namespace

// ------------------------------------------------------------------------------------------------ C-ABI
extern "C" int harc_amd_comm_get_id(uint8_t *id, size_t id_bytes)
{
    if (!id || id_bytes < sizeof(ncclUniqueId)) { harc_set_error("harc_amd_comm_get_id: buffer of at least %zu bytes needed", sizeof(ncclUniqueId)); return HARC_AMD_EINVAL; }
    RC_TRY(rccl_load());
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memset(id, 0, id_bytes);
    memcpy(id, &u, sizeof u);
    return HARC_AMD_OK;
}

extern "C" int harc_amd_comm_init(harc_amd_ctx *c, const uint8_t *id, size_t id_bytes, int32_t world, int32_t rank)
{
    if (!c || !id || id_bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) { harc_set_error("harc_amd_comm_init: bad arguments"); return HARC_AMD_EINVAL; }
    HIP_TRY(hipSetDevice(c->P.device));
    RC_TRY(rccl_load());
    delete c->comm; c->comm = nullptr;
    RcclComm *r = new RcclComm();
    r->world = world; r->rank = rank;
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    ncclResult_t e = g_rccl.CommInitRank(&r->comm, world, u, rank);
    if (e != ncclSuccess) { r->comm = nullptr; delete r; harc_set_error("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, g_rccl.GetErrorString(e)); return HARC_AMD_ENODEVICE; }
    c->comm = r;
    return HARC_AMD_OK;
}

extern "C" int harc_amd_comm_init_mailbox(harc_amd_ctx *c, const char *dir, int32_t world, int32_t rank)
{
    if (!c || !dir || world < 1 || rank < 0 || rank >= world) { harc_set_error("harc_amd_comm_init_mailbox: bad arguments"); return HARC_AMD_EINVAL; }
#ifndef HARC_AMD_TEST_TRANSPORT
    harc_set_error("harc_amd_comm_init_mailbox: this library was built without the test transport (make TEST_TRANSPORT=1)");
    return HARC_AMD_ESTATE;
#else
    delete c->comm; c->comm = nullptr;
    MailboxComm *m = new MailboxComm();
    m->world = world; m->rank = rank; m->dir = dir;
    if (const char *e = getenv("HARC_AMD_MAILBOX_TIMEOUT")) m->timeout_s = atof(e);
    c->comm = m;
    return HARC_AMD_OK;
#endif
}

extern "C" int harc_amd_comm_destroy(harc_amd_ctx *c)
{
    if (!c) return HARC_AMD_EINVAL;
    (void)hipSetDevice(c->P.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->comm; c->comm = nullptr;
    return HARC_AMD_OK;
}

extern "C" int harc_amd_comm_barrier(harc_amd_ctx *c)
{
    if (!c || !c->comm) { harc_set_error("harc_amd_comm_barrier: no communicator"); return HARC_AMD_ESTATE; }
    HIP_TRY(hipSetDevice(c->P.device));
    std::vector<uint64_t> all((size_t)c->comm->world);
    const uint64_t one = 1;
    return c->comm->allgather_u64(c, &one, 1, all.data());
}
