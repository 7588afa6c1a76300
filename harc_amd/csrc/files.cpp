// files.cpp -- the reference's stage programs as library calls with the same file contract:
//   reorder.out <basedir>    src/reorder.cpp:100-131   reads  output/input_clean.dna, output/numreads.bin
//                                                       writes temp.dna, temp.dna.singleton, read_rev.txt, tempflag.txt,
//                                                              temppos.txt, read_order.bin, read_order.bin.singleton
//   encoder.out <basedir>    src/encoder.cpp:108-152   reads  the files above + input_N.dna
//                                                       writes read_{seq,pos,noise,noisepos,rev}.txt.<e>(+.tail), read_singleton.txt(+.tail),
//                                                              read_order.bin, read_order_N_pe.bin, input_N.dna (rewritten), read_meta.txt
//   pack_order.out <basedir> src/pack_order.cpp:11-77  rewrites read_order.bin, writes read_order.bin.tail
#include "internal.h"
#include <string>

static bool slurp(const std::string &path, std::vector<char> &out, bool must_exist)
{
    out.clear();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { if (must_exist) harc_set_error("cannot open %s", path.c_str()); return !must_exist; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    if (!ok) harc_set_error("short read on %s", path.c_str());
    return ok;
}
static int spit(const std::string &path, const void *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { harc_set_error("cannot create %s", path.c_str()); return HARC_AMD_EIO; }
    if (n && fwrite(p, 1, n, f) != n) { fclose(f); harc_set_error("short write on %s", path.c_str()); return HARC_AMD_EIO; }
    fclose(f);
    return HARC_AMD_OK;
}
static int spit_stream(harc_amd_ctx *c, int id, int shard, const std::string &path)
{
    const void *p = nullptr; size_t n = 0;
    RC_TRY(harc_amd_get_stream(c, id, shard, &p, &n));
    return spit(path, p, n);
}

struct CtxGuard { harc_amd_ctx *c = nullptr; ~CtxGuard() { harc_amd_destroy(c); } };

static int load_clean(harc_amd_ctx *c, const std::string &od)
{
    const int L = c->P.readlen;
    std::vector<char> nb, dna;
    if (!slurp(od + "numreads.bin", nb, true) || nb.size() < 4) { harc_set_error("numreads.bin missing or short"); return HARC_AMD_EIO; }
    uint32_t N; memcpy(&N, nb.data(), 4);                                          // reorder.cpp:112-113
    if (!slurp(od + "input_clean.dna", dna, true)) return HARC_AMD_EIO;
    if (dna.size() < (size_t)N * (L + 1)) { harc_set_error("input_clean.dna holds fewer than %u lines of %d bases", N, L); return HARC_AMD_EIO; }
    return harc_amd_set_reads_ascii(c, dna.data(), N, (uint32_t)L + 1);             // (readlen+1) stride, reorder.cpp:252
}
static int load_N(harc_amd_ctx *c, const std::string &od)
{
    const int L = c->P.readlen;
    std::vector<char> n;
    if (!slurp(od + "input_N.dna", n, false)) return HARC_AMD_EIO;
    return harc_amd_set_nreads_ascii(c, n.data(), (uint32_t)(n.size() / (L + 1)), (uint32_t)L + 1);   // encoder.cpp:804-808
}
static int write_stage1(harc_amd_ctx *c, const std::string &od)
{
    RC_TRY(spit_stream(c, HARC_AMD_S1_DNA, 0, od + "temp.dna"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_DNA_SINGLETON, 0, od + "temp.dna.singleton"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_RC, 0, od + "read_rev.txt"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_FLAG, 0, od + "tempflag.txt"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_POS, 0, od + "temppos.txt"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_ORDER, 0, od + "read_order.bin"));
    RC_TRY(spit_stream(c, HARC_AMD_S1_ORDER_SINGLETON, 0, od + "read_order.bin.singleton"));
    return HARC_AMD_OK;
}
static int write_stage2(harc_amd_ctx *c, const std::string &od)
{
    for (int e = 0; e < c->P.num_thr; e++) {
        const std::string s = "." + std::to_string(e);
        RC_TRY(spit_stream(c, HARC_AMD_S2_SEQ, e, od + "read_seq.txt" + s));
        RC_TRY(spit_stream(c, HARC_AMD_S2_SEQ_TAIL, e, od + "read_seq.txt" + s + ".tail"));
        RC_TRY(spit_stream(c, HARC_AMD_S2_POS, e, od + "read_pos.txt" + s));
        RC_TRY(spit_stream(c, HARC_AMD_S2_NOISE, e, od + "read_noise.txt" + s));
        RC_TRY(spit_stream(c, HARC_AMD_S2_NOISEPOS, e, od + "read_noisepos.txt" + s));
        RC_TRY(spit_stream(c, HARC_AMD_S2_REV, e, od + "read_rev.txt" + s));
        RC_TRY(spit_stream(c, HARC_AMD_S2_REV_TAIL, e, od + "read_rev.txt" + s + ".tail"));
    }
    RC_TRY(spit_stream(c, HARC_AMD_S2_ORDER, 0, od + "read_order.bin"));
    RC_TRY(spit_stream(c, HARC_AMD_S2_ORDER_N_PE, 0, od + "read_order_N_pe.bin"));
    RC_TRY(spit_stream(c, HARC_AMD_S2_INPUT_N, 0, od + "input_N.dna"));
    RC_TRY(spit_stream(c, HARC_AMD_S2_META, 0, od + "read_meta.txt"));
    RC_TRY(spit_stream(c, HARC_AMD_S2_SINGLETON, 0, od + "read_singleton.txt"));
    RC_TRY(spit_stream(c, HARC_AMD_S2_SINGLETON_TAIL, 0, od + "read_singleton.txt.tail"));
    return HARC_AMD_OK;
}

extern "C" int harc_amd_reorder_files(const harc_amd_params *params, const char *basedir)
{
    if (!params || !basedir) return HARC_AMD_EINVAL;
    CtxGuard g; RC_TRY(harc_amd_create(params, &g.c));
    const std::string od = std::string(basedir) + "/output/";
    RC_TRY(load_clean(g.c, od));
    RC_TRY(harc_amd_reorder(g.c));
    RC_TRY(write_stage1(g.c, od));
    harc_amd_counters C; harc_amd_get_counters(g.c, &C);
    printf("Reordering done, %llu were unmatched\n", (unsigned long long)C.unmatched);                 // reorder.cpp:701
    return HARC_AMD_OK;
}

extern "C" int harc_amd_encoder_files(const harc_amd_params *params, const char *basedir)
{
    if (!params || !basedir) return HARC_AMD_EINVAL;
    CtxGuard g; RC_TRY(harc_amd_create(params, &g.c));
    const int L = params->readlen;
    const std::string od = std::string(basedir) + "/output/";
    std::vector<char> dna, flag, pos, order, rc, dna_s, order_s;
    if (!slurp(od + "temp.dna", dna, true) || !slurp(od + "tempflag.txt", flag, true) || !slurp(od + "temppos.txt", pos, true) ||
        !slurp(od + "read_order.bin", order, true) || !slurp(od + "read_rev.txt", rc, true) ||
        !slurp(od + "temp.dna.singleton", dna_s, true) || !slurp(od + "read_order.bin.singleton", order_s, true)) return HARC_AMD_EIO;
    const uint32_t M = (uint32_t)(order.size() / 4), S = (uint32_t)(dna_s.size() / (L + 1));           // encoder.cpp:781-803
    if (dna.size() < (size_t)M * (L + 1) || flag.size() < M || pos.size() < M || rc.size() < M || order_s.size() < (size_t)S * 4) {
        harc_set_error("stage-I files are inconsistent"); return HARC_AMD_EIO;
    }
    RC_TRY(harc_amd_set_stage1_streams(g.c, dna.data(), (const uint8_t *)flag.data(), (const uint8_t *)pos.data(), (const uint32_t *)order.data(),
                                       (const uint8_t *)rc.data(), M, dna_s.data(), (const uint32_t *)order_s.data(), S));
    RC_TRY(load_N(g.c, od));
    RC_TRY(harc_amd_encode(g.c));
    RC_TRY(write_stage2(g.c, od));
    harc_amd_counters C; harc_amd_get_counters(g.c, &C);
    printf("Encoding done:\n%llu singleton reads were aligned\n%llu reads with N were aligned\n",        // encoder.cpp:506-508
           (unsigned long long)C.aligned_singletons, (unsigned long long)C.aligned_N);
    return HARC_AMD_OK;
}

extern "C" int harc_amd_compress_files(const harc_amd_params *params, const char *basedir)
{
    if (!params || !basedir) return HARC_AMD_EINVAL;
    CtxGuard g; RC_TRY(harc_amd_create(params, &g.c));
    const std::string od = std::string(basedir) + "/output/";
    RC_TRY(load_clean(g.c, od));
    RC_TRY(load_N(g.c, od));
    RC_TRY(harc_amd_reorder(g.c));
    RC_TRY(harc_amd_encode(g.c));
    RC_TRY(spit_stream(g.c, HARC_AMD_S1_ORDER_SINGLETON, 0, od + "read_order.bin.singleton"));          // harc:133 removes *.singleton
    RC_TRY(write_stage2(g.c, od));
    harc_amd_counters C; harc_amd_get_counters(g.c, &C);
    printf("Reordering done, %llu were unmatched\n", (unsigned long long)C.unmatched);
    printf("Encoding done:\n%llu singleton reads were aligned\n%llu reads with N were aligned\n",
           (unsigned long long)C.aligned_singletons, (unsigned long long)C.aligned_N);
    return HARC_AMD_OK;
}

extern "C" int harc_amd_pack_order_files(const harc_amd_params *params, const char *basedir)
{
    if (!params || !basedir) return HARC_AMD_EINVAL;
    CtxGuard g; RC_TRY(harc_amd_create(params, &g.c));
    const std::string od = std::string(basedir) + "/output/";
    std::vector<char> in;
    if (!slurp(od + "read_order.bin", in, true)) return HARC_AMD_EIO;
    std::vector<uint8_t> &b = out_buf(g.c, HARC_AMD_S2_ORDER, 0);
    b.assign(in.begin(), in.end());
    g.c->have_s2 = true;
    RC_TRY(harc_amd_pack_order(g.c));
    RC_TRY(spit_stream(g.c, HARC_AMD_P_ORDER, 0, od + "read_order.bin"));
    RC_TRY(spit_stream(g.c, HARC_AMD_P_ORDER_TAIL, 0, od + "read_order.bin.tail"));
    return HARC_AMD_OK;
}

// preprocess.out <fastq> <basedir> <preserve_order> <preserve_quality> <readlen>   (src/preprocess.cpp:50-137)
// Splits the FASTQ into output/input_clean.dna (reads without N), output/input_N.dna, output/read_order_N.bin and
// output/numreads.bin.  Quality / id side files (-q) are outside the hot path and not produced here.
extern "C" int harc_amd_preprocess_files(const char *fastq, const char *basedir, int32_t readlen)
{
    if (!fastq || !basedir || readlen < 1 || readlen > 255) { harc_set_error("preprocess: bad arguments"); return HARC_AMD_EINVAL; }
    FILE *in = fopen(fastq, "rb");
    if (!in) { harc_set_error("cannot open %s", fastq); return HARC_AMD_EIO; }
    const std::string od = std::string(basedir) + "/output/";
    FILE *fc = fopen((od + "input_clean.dna").c_str(), "wb"), *fn = fopen((od + "input_N.dna").c_str(), "wb"),
         *fo = fopen((od + "read_order_N.bin").c_str(), "wb");
    if (!fc || !fn || !fo) { if (fc) fclose(fc); if (fn) fclose(fn); if (fo) fclose(fo); fclose(in); harc_set_error("cannot create files under %s", od.c_str()); return HARC_AMD_EIO; }
    std::vector<char> buf(1 << 16);
    std::string line;
    uint64_t readnum = 0, nclean = 0; int li = 0, rc = HARC_AMD_OK;
    auto flush_line = [&]() {
        if (li == 1) {
            if ((int)line.size() != readlen) {                    // preprocess.cpp:92-97
                printf("Read length not fixed. Found two different read lengths: %d and %zu\n", readlen, line.size());
                harc_set_error("read length not fixed"); rc = HARC_AMD_EINVAL; return;
            }
            if (line.find('N') != std::string::npos) {
                fwrite(line.data(), 1, line.size(), fn); fputc('\n', fn);
                const uint32_t rn = (uint32_t)readnum; fwrite(&rn, 4, 1, fo);          // low 4 bytes of the counter, preprocess.cpp:102
            } else { fwrite(line.data(), 1, line.size(), fc); fputc('\n', fc); nclean++; }
        }
        if (li == 3) readnum++;
        li = (li + 1) & 3;
        line.clear();
    };
    size_t got;
    while (rc == HARC_AMD_OK && (got = fread(buf.data(), 1, buf.size(), in)) > 0) {
        size_t s = 0;
        for (size_t i = 0; i < got && rc == HARC_AMD_OK; i++)
            if (buf[i] == '\n') { line.append(buf.data() + s, i - s); s = i + 1; flush_line(); }
        if (s < got) line.append(buf.data() + s, got - s);
    }
    if (rc == HARC_AMD_OK && !line.empty()) flush_line();
    fclose(in); fclose(fc); fclose(fn); fclose(fo);
    if (rc != HARC_AMD_OK) return rc;
    if (readnum > 4294967290ull) { printf("Too many reads. HARC supports at most 4294967290 reads\n"); harc_set_error("too many reads"); return HARC_AMD_EINVAL; }   // :122-126
    const uint32_t n32 = (uint32_t)nclean;
    RC_TRY(spit(od + "numreads.bin", &n32, 4));
    printf("Read length: %d\nTotal number of reads: %llu\nTotal number of reads without N: %llu\nPreprocessing Done!\n", readlen,
           (unsigned long long)readnum, (unsigned long long)nclean);
    return HARC_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU: merge of the rank parts
// Every rank of harc_amd_compress_fastq_shard_files leaves, under <basedir>/output/.shard/, its part of the files that exist once per
// archive.  The layout the decoders expect (encoder.cpp:457-503, decoder.cpp:141-169, decoder_preserve.cpp:212-290):
//   read_order.bin       = [aligned clean reads, shard by shard][unaligned singletons in read_singleton.txt order]
//   read_order_N_pe.bin  = [aligned N reads, shard by shard][unaligned N reads in input_N.dna order]
// so the aligned halves of all ranks come first, in rank order (= shard order rank*E + e), then the unaligned halves in rank order,
// and read_singleton.txt / input_N.dna are the rank parts in the same rank order.  read_singleton.txt packs 4 bases per byte
// (encoder.cpp:527-548) with up to 3 bases of ASCII tail: the parts are re-packed across the joints.
namespace {
struct Appender {
    FILE *f = nullptr; std::string path;
    int open(const std::string &p) { path = p; f = fopen(p.c_str(), "wb"); if (!f) { harc_set_error("cannot create %s", p.c_str()); return HARC_AMD_EIO; } return HARC_AMD_OK; }
    int add(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { harc_set_error("short write on %s", path.c_str()); return HARC_AMD_EIO; } return HARC_AMD_OK; }
    int add_file(const std::string &src, bool must_exist)
    {
        FILE *in = fopen(src.c_str(), "rb");
        if (!in) { if (must_exist) { harc_set_error("cannot open %s", src.c_str()); return HARC_AMD_EIO; } return HARC_AMD_OK; }
        std::vector<char> buf((size_t)4 << 20); size_t got; int rc = HARC_AMD_OK;
        while (rc == HARC_AMD_OK && (got = fread(buf.data(), 1, buf.size(), in)) > 0) rc = add(buf.data(), got);
        fclose(in);
        return rc;
    }
    ~Appender() { if (f) fclose(f); }
};
}

extern "C" int harc_amd_merge_shard_files(const char *basedir, int32_t world)
{
    if (!basedir || world < 1) { harc_set_error("merge_shard_files: bad arguments"); return HARC_AMD_EINVAL; }
    const std::string od = std::string(basedir) + "/output/", sd = od + ".shard/";
    auto part = [&](const char *stem, int r) { return sd + stem + "." + std::to_string(r); };
    // totals (and the proof that every rank finished)
    int L = 0; unsigned long long nrec = 0, nclean = 0, unmatched = 0, al_s = 0, al_N = 0;
    for (int r = 0; r < world; r++) {
        std::vector<char> st;
        if (!slurp(part("stats", r), st, true)) { harc_set_error("rank %d left no result under %s", r, sd.c_str()); return HARC_AMD_EIO; }
        st.push_back(0);
        int l = 0; unsigned long long v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (sscanf(st.data(), "%d %llu %llu %llu %llu %llu %llu %llu %llu", &l, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) != 9) { harc_set_error("stats of rank %d unreadable", r); return HARC_AMD_EIO; }
        if (r && l != L) { harc_set_error("ranks disagree on the read length (%d vs %d)", L, l); return HARC_AMD_EINVAL; }
        L = l; nrec += v[0]; nclean += v[1]; unmatched += v[3]; al_s += v[4]; al_N += v[5];
    }
    if (nrec > 4294967290ull) { printf("Too many reads. HARC supports at most 4294967290 reads\n"); harc_set_error("too many reads"); return HARC_AMD_EINVAL; }
    printf("Read length: %d\nTotal number of reads: %llu\nTotal number of reads without N: %llu\nPreprocessing Done!\n", L, nrec, nclean);   // preprocess.cpp:133-136
    printf("Reordering done, %llu were unmatched\n", unmatched);                                                    // reorder.cpp:701
    printf("Encoding done:\n%llu singleton reads were aligned\n%llu reads with N were aligned\n", al_s, al_N);      // encoder.cpp:506-508
    {
        Appender o, on, in_n, oN;
        RC_TRY(o.open(od + "read_order.bin")); RC_TRY(on.open(od + "read_order_N_pe.bin")); RC_TRY(in_n.open(od + "input_N.dna")); RC_TRY(oN.open(od + "read_order_N.bin"));
        for (int r = 0; r < world; r++) { RC_TRY(o.add_file(part("order_a", r), true)); RC_TRY(on.add_file(part("orderN_a", r), true)); }
        for (int r = 0; r < world; r++) {
            RC_TRY(o.add_file(part("order_u", r), true)); RC_TRY(on.add_file(part("orderN_u", r), true));
            RC_TRY(in_n.add_file(part("input_N", r), true)); RC_TRY(oN.add_file(part("order_N", r), true));
        }
    }
    {   // read_singleton.txt: 2 bits per base across the joints (A0 C1 G2 T3, first base in the low bits: encoder.cpp:540-541)
        Appender sg; RC_TRY(sg.open(od + "read_singleton.txt"));
        unsigned pend = 0; int npend = 0;                          // bases waiting for a full byte
        std::vector<char> in, tail; std::vector<uint8_t> outb;
        auto code = [](char ch) -> unsigned { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u; };
        for (int r = 0; r < world; r++) {
            if (!slurp(part("singleton", r), in, true) || !slurp(part("singleton_tail", r), tail, true)) return HARC_AMD_EIO;
            if (npend == 0) RC_TRY(sg.add(in.data(), in.size()));
            else {
                outb.resize(in.size());
                const int sh = 2 * npend;
                for (size_t i = 0; i < in.size(); i++) { const unsigned b = (uint8_t)in[i]; outb[i] = (uint8_t)(pend | (b << sh)); pend = b >> (8 - sh); }
                RC_TRY(sg.add(outb.data(), outb.size()));
            }
            for (char ch : tail) {
                pend |= code(ch) << (2 * npend);
                if (++npend == 4) { const uint8_t b = (uint8_t)pend; RC_TRY(sg.add(&b, 1)); pend = 0; npend = 0; }
            }
        }
        char t[4]; for (int k = 0; k < npend; k++) t[k] = "ACGT"[(pend >> (2 * k)) & 3];
        RC_TRY(spit(od + "read_singleton.txt.tail", t, (size_t)npend));
    }
    { const uint32_t n32 = (uint32_t)nclean; RC_TRY(spit(od + "numreads.bin", &n32, 4)); }
    { char m[32]; const int ml = snprintf(m, sizeof m, "%d\n", L); RC_TRY(spit(od + "read_meta.txt", m, (size_t)ml)); }
    {   // -q -p: quality values and ids in file order = the slices in rank order
        FILE *probe = fopen(part("quality", 0).c_str(), "rb");
        if (probe) {
            fclose(probe);
            Appender q, i; RC_TRY(q.open(od + "output.quality")); RC_TRY(i.open(od + "output.id"));
            for (int r = 0; r < world; r++) { RC_TRY(q.add_file(part("quality", r), true)); RC_TRY(i.add_file(part("id", r), true)); }
        }
    }
    // the parts are gone from the archive
    static const char *stems[] = { "stats", "order_a", "order_u", "orderN_a", "orderN_u", "input_N", "order_N", "singleton", "singleton_tail", "quality", "id" };
    for (int r = 0; r < world; r++) for (const char *s : stems) (void)remove(part(s, r).c_str());
    (void)remove((sd + "comm_id").c_str());
    (void)remove(sd.substr(0, sd.size() - 1).c_str());            // rmdir when empty (a mailbox directory, if any, is the caller's)
    return HARC_AMD_OK;
}
