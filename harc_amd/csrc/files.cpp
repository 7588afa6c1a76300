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
