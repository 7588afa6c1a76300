// How fast can a file in /dev/shm reach HBM?  (round 6: the end-to-end leg of bench.py is bound by the file reads of ingest.hip's FileFeeder)
//   (a) T threads pread() 16-MB slices into malloc'd / hipHostMalloc'd buffers (no upload): the CPU copy rate out of the page cache
//   (b) mmap the file, hipHostRegister the mapping (piecewise, P threads), hipMemcpyAsync straight from it: no CPU copy at all
//   hipcc -O2 -o feed_rate feed_rate.cpp -lpthread ; ./feed_rate [GB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t GB = argc > 1 ? (size_t)atoi(argv[1]) : 8, n = GB << 30, SL = (size_t)16 << 20;
    const char *path = "/dev/shm/harc_feed_rate.bin";
    {   // the file: 64 writer threads
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600); if (fd < 0 || ftruncate(fd, (off_t)n)) { perror("file"); return 1; }
        std::atomic<size_t> nx{0}; std::vector<std::thread> th; const double t0 = now();
        for (int t = 0; t < 16; t++) th.emplace_back([&] { std::vector<char> b(SL, 'A'); for (;;) { size_t k = nx.fetch_add(1); if (k * SL >= n) break; b[0] = (char)k; if (pwrite(fd, b.data(), SL, (off_t)(k * SL)) != (ssize_t)SL) perror("pwrite"); } });
        for (auto &x : th) x.join();
        printf("wrote %zu GB with 16 threads: %.2f GB/s\n", GB, (double)n / (now() - t0) / 1e9); close(fd);
    }
    int fd = open(path, O_RDONLY);
    if (argc > 2) {     // ./feed_rate GB ring: what ingest.hip's FileFeeder does -- file written by ONE thread in 868-MB writes, 16 readers, a rotating ring of 24 pinned slices, uploads
        close(fd); unlink(path);
        { int wf = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600); std::vector<char> b((size_t)868 << 20, 'C'); const double t0 = now(); for (size_t a = 0; a < n; a += b.size()) { const size_t len = n - a < b.size() ? n - a : b.size(); if (write(wf, b.data(), len) != (ssize_t)len) perror("write"); }
          printf("wrote %zu GB with ONE thread in 868-MB writes: %.2f GB/s\n", GB, (double)n / (now() - t0) / 1e9); close(wf); }
        fd = open(path, O_RDONLY);
        char *dd = nullptr; if (hipMalloc((void **)&dd, n) != hipSuccess) return 3;
        hipStream_t s2; (void)hipStreamCreate(&s2);
        for (int pass = 0; pass < 4; pass++) {
            const int T = 16, NS = 24; const bool upload = pass >= 2; const size_t sl = pass == 3 ? (size_t)64 << 20 : SL;
            char *ring; if (hipHostMalloc((void **)&ring, sl * NS) != hipSuccess) return 2;
            std::atomic<size_t> nx{0}; std::vector<std::thread> th; std::atomic<long long> ns_pread{0};
            // slice k uses ring slot k % NS; an uploader thread (this one) copies slice k when it is read; readers wait until slot's previous copy is done
            const size_t nchunk = (n + sl - 1) / sl; std::vector<std::atomic<int>> state(nchunk); for (auto &x : state) x = 0;   // 1 = read, 2 = uploaded
            const double t0 = now();
            for (int t = 0; t < T; t++) th.emplace_back([&] { for (;;) { size_t k = nx.fetch_add(1); if (k >= nchunk) break; if (k >= (size_t)NS) while (state[k - NS].load() < 2) std::this_thread::yield();
                const size_t len = (k + 1) * sl <= n ? sl : n - k * sl; const double a = now(); size_t got = 0; while (got < len) { ssize_t r = pread(fd, ring + (k % NS) * sl + got, len - got, (off_t)(k * sl + got)); if (r <= 0) break; got += (size_t)r; }
                ns_pread += (long long)((now() - a) * 1e9); state[k] = 1; } });
            std::vector<hipEvent_t> ev(NS); for (auto &e : ev) (void)hipEventCreate(&e);
            for (size_t k = 0; k < nchunk; k++) { while (state[k].load() < 1) std::this_thread::yield(); const size_t len = (k + 1) * sl <= n ? sl : n - k * sl;
                if (upload) { (void)hipMemcpyAsync(dd + k * sl, ring + (k % NS) * sl, len, hipMemcpyHostToDevice, s2); (void)hipEventRecord(ev[k % NS], s2); }
                if (k >= (size_t)NS / 2) { const size_t j = k - NS / 2; if (upload) (void)hipEventSynchronize(ev[j % NS]); state[j] = 2; } }
            (void)hipStreamSynchronize(s2); for (size_t k = 0; k < nchunk; k++) state[k] = 2;
            for (auto &x : th) x.join();
            printf("pass %d (%s, %zu-MB slices, ring of %d, %d readers%s): %.2f GB/s; readers inside pread %.2f s summed\n", pass, pass == 0 ? "first read of the file" : "file read before", sl >> 20, NS, T, upload ? ", uploads" : ", no uploads", (double)n / (now() - t0) / 1e9, ns_pread.load() / 1e9);
            (void)hipHostFree(ring);
        }
        close(fd); unlink(path); return 0;
    }
    for (int pinned = 0; pinned < 2; pinned++)
        for (int T : { 4, 8, 16, 32, 64 }) {
            std::vector<char *> buf(T);
            for (int t = 0; t < T; t++) { if (pinned) { if (hipHostMalloc((void **)&buf[t], SL) != hipSuccess) return 2; } else buf[t] = (char *)malloc(SL); memset(buf[t], 1, SL); }
            std::atomic<size_t> nx{0}; std::vector<std::thread> th; const double t0 = now();
            for (int t = 0; t < T; t++) th.emplace_back([&, t] { for (;;) { size_t k = nx.fetch_add(1); if (k * SL >= n) break; size_t got = 0; while (got < SL) { ssize_t r = pread(fd, buf[t] + got, SL - got, (off_t)(k * SL + got)); if (r <= 0) break; got += (size_t)r; } } });
            for (auto &x : th) x.join();
            printf("pread into %s buffers, %2d threads: %.2f GB/s\n", pinned ? "pinned  " : "malloc'd", T, (double)n / (now() - t0) / 1e9);
            for (int t = 0; t < T; t++) { if (pinned) (void)hipHostFree(buf[t]); else free(buf[t]); }
        }
    // (b) the mapping registered piecewise
    char *d = nullptr; if (hipMalloc((void **)&d, n) != hipSuccess) return 3;
    hipStream_t st; (void)hipStreamCreate(&st);
    for (int P : { 1, 4, 16 }) for (size_t piece : { (size_t)256 << 20, (size_t)1 << 30 }) {
        double t0 = now();
        char *m = (char *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0); if (m == MAP_FAILED) { perror("mmap"); return 4; }
        const size_t np = (n + piece - 1) / piece; std::vector<int> ok(np, 0); std::atomic<size_t> nx{0}; std::vector<std::thread> th;
        std::atomic<int> bad{0};
        for (int t = 0; t < P; t++) th.emplace_back([&] { for (;;) { size_t k = nx.fetch_add(1); if (k >= np) break; const size_t len = (k + 1) * piece <= n ? piece : n - k * piece;
            if (hipHostRegister(m + k * piece, len, hipHostRegisterDefault) != hipSuccess) { bad++; continue; } ok[k] = 1; } });
        for (auto &x : th) x.join();
        const double t1 = now();
        if (bad) printf("hipHostRegister failed on %d pieces\n", bad.load());
        for (size_t k = 0; k < np; k++) if (ok[k]) { const size_t len = (k + 1) * piece <= n ? piece : n - k * piece; (void)hipMemcpyAsync(d + k * piece, m + k * piece, len, hipMemcpyHostToDevice, st); }
        (void)hipStreamSynchronize(st);
        const double t2 = now();
        for (size_t k = 0; k < np; k++) if (ok[k]) (void)hipHostUnregister(m + k * piece);
        munmap(m, n);
        const double t3 = now();
        printf("mmap + hipHostRegister in %4zu-MB pieces by %2d threads: register %.2f GB/s, upload from the mapping %.2f GB/s, unregister + unmap %.2f s; all three %.2f GB/s\n", piece >> 20, P, (double)n / (t1 - t0) / 1e9, (double)n / (t2 - t1) / 1e9, t3 - t2, (double)n / (t3 - t0) / 1e9);
    }
    // plain pinned upload for comparison
    { char *h; (void)hipHostMalloc((void **)&h, (size_t)1 << 30); memset(h, 1, (size_t)1 << 30); const double t0 = now(); for (size_t k = 0; k < GB; k++) (void)hipMemcpyAsync(d + (k << 30), h, (size_t)1 << 30, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st);
      printf("upload from a pinned buffer: %.2f GB/s\n", (double)n / (now() - t0) / 1e9); (void)hipHostFree(h); }
    close(fd); unlink(path);
    return 0;
}
