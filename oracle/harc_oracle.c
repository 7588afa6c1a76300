/*
 * harc_oracle.c -- CPU restatement of the HARC reorder + encode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the MI355X implementation in harc_amd/: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libharc_amd.so) never links or calls it.
 *
 * PARITY PIN: harc_oracle_reorder / harc_oracle_encoder / harc_oracle_pack_order with K=1, E=1 are checked
 * byte-for-byte against tests/golden/*.tar.xz, which were produced by the real reference compiled from
 * /root/reference by oracle/build_ref.sh and run at num_thr=1 (oracle/make_goldens.py).
 *
 * What is restated (file:line are into /root/reference):
 *   stage I   src/reorder.cpp: 2-bit read store :184-209, dictionary :277-432 (BBHash src/BooPHF.h replaced by an
 *             exact key->bin map: the MPHF value never reaches an output byte, it only addresses startpos and picks
 *             a lock stripe, reorder.cpp:508-513), greedy chaining :455-689, consensus :863-915, output files :722-830
 *   stage II  src/encoder.cpp: contigs :219-441, buildcontig :619-652, singleton/N realignment :231-418,
 *             writecontig :654-717, enc_noise :751-771, tail section :457-503, packbits :512-616
 *   -p        src/pack_order.cpp:20-77
 *   decoder   src/decoder.cpp:65-280 (used as the round-trip checker on the GPU box, where the reference is absent)
 *
 * Parallel semantics.  The reference is a race for num_thr>1 (reorder.cpp:545-552, encoder.cpp:282-283).  K chains /
 * E shards are therefore DEFINED here, deterministically, so that K=1,E=1 is the reference at num_thr=1:
 *   stage I, K chains x S steps, super-round-synchronous (see stage1_run): every chain walks up to S steps against the frozen
 *   claim state, candidate priority exactly reorder.cpp:517-649; a read goes to the smallest (step, chain) bid, a chain keeps
 *   the steps before its first lost bid; chains out of candidates reseed in ascending chain id from ONE global descending
 *   cursor (reorder.cpp:652-668).  Seeds c*floor(N/K) (reorder.cpp:490).  Output = per-chain streams concatenated in chain
 *   order (reorder.cpp:778-821).
 *   stage II, E shards: shard ranges encoder.cpp:171-180; singleton claims resolve to the minimum
 *   (shard, contig, j, direction, dict) tuple, which is what the sequential loop below produces.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MAXW 8          /* ceil(2*255/64) */
#define MAXL 255
#define NONE 0xFFFFFFFFu
#define MAXSTEPS 64
#define NSUGG 8           /* look-ahead seeds handed to a chain at every reseed */
#define LOOK_CHUNKS 16     /* the look-ahead inspects at most this many chunks of 1024 64-bit bitmap words below the cursor (the GPU's k_reseed) */
#define LARGEBIN 16u      /* HARC_LARGEBIN of harc_amd/csrc/stage1.hip */
#define STEP_CAP 12       /* HARC_STEP_CAP: a step that has made this many probes into such bins without a hit is put off: the walk ends in front of it and the
                              next super-round takes the step up again BEHIND the probes already made (they found nothing against fewer claims) */
#define BO_FREE 1          /* HARC_BO_FREE / HARC_BO_CAP of stage1.hip: on repeat-rich input with more than 16 384 chains a chain whose walk was cut at a lost bid */
#define BO_CAP 3           /* sits out 2^(k - BO_FREE) - 1 super-rounds, k = its cuts in a row, at most BO_FREE + BO_CAP (0, 1, 3, 7, 7 ... rounds) */
#define SCAN_BUDGET 12     /* HARC_SCAN_BUDGET: a walk ends after the step in which its probes into bins of more than LARGEBIN reads (not yet exhausted) reach this number */

/* ------------------------------------------------------------------ parameters (harc:52-60) */
typedef struct {
    int L, W, maxmatch, thresh, thresh_s, maxsearch;
    int ds[2], de[2], kbits[2];
} params_t;

static void params_init(params_t *p, int L)
{
    p->L = L; p->W = (2 * L + 63) / 64;
    p->maxmatch = L / 2; p->thresh = 4; p->thresh_s = 24; p->maxsearch = 1000;
    int h = L / 2;
    if (L > 100) { p->ds[0] = h - 32; p->de[0] = h - 1; p->ds[1] = h; p->de[1] = h - 1 + 32; }
    else { p->ds[0] = h - L * 32 / 100; p->de[0] = h - 1; p->ds[1] = h; p->de[1] = h - 1 + L * 32 / 100; }
    for (int l = 0; l < 2; l++) p->kbits[l] = 2 * (p->de[l] - p->ds[l] + 1);
}

/* ------------------------------------------------------------------ 2-bit read store (reorder.cpp:184-209) */
/* base i at bits 2i,2i+1: A=(0,0) C=(0,1) G=(1,0) T=(1,1)  => as integer A=0 G=1 C=2 T=3 */
static int pc_of(char c) { switch (c) { case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; } return -1; }
static const char pc_char[4] = { 'A', 'G', 'C', 'T' };               /* reorder.cpp:59 */
static const int pc_to_idx[4] = { 0, 2, 1, 3 };                      /* packed code -> chartoint (A0 C1 G2 T3, :139-142) */
static const int idx_to_pc[4] = { 0, 2, 1, 3 };

static void pack_read(const char *s, int L, int W, uint64_t *w)
{
    for (int i = 0; i < W; i++) w[i] = 0;
    for (int i = 0; i < L; i++) w[(2 * i) >> 6] |= (uint64_t)pc_of(s[i]) << ((2 * i) & 63);
}
static void unpack_read(const uint64_t *w, int L, char *s, int rc)
{
    for (int i = 0; i < L; i++) {
        int v = (int)((w[(2 * i) >> 6] >> ((2 * i) & 63)) & 3);
        if (!rc) s[i] = pc_char[v]; else s[L - 1 - i] = pc_char[3 - v];
    }
}
static uint64_t extract_bits(const uint64_t *w, int W, int off, int nbits)
{
    int wi = off >> 6, sh = off & 63;
    uint64_t v = w[wi] >> sh;
    if (sh && wi + 1 < W) v |= w[wi + 1] << (64 - sh);
    if (nbits < 64) v &= (((uint64_t)1) << nbits) - 1;
    return v;
}

/* ------------------------------------------------------------------ exact dictionary (reorder.cpp:277-432 semantics) */
typedef struct { uint64_t key; uint32_t id; } kv_t;
typedef struct {
    uint32_t nkeys;
    uint64_t *keys;      /* sorted unique */
    uint32_t *start;     /* nkeys+1 */
    uint32_t *ids;       /* ascending inside a bin (reorder.cpp:372-384) */
    uint32_t *live_end;  /* hint: entries at/above live_end[bin] are all claimed */
    uint32_t *binof;     /* per read id: its bin */
    uint32_t *nlive;     /* per bin: reads not yet claimed */
    uint32_t hmask; uint32_t *hslot;   /* open addressing key -> bin+1 */
} dict_t;

static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

static void dict_build(dict_t *d, const uint64_t *keys, uint32_t n)
{
    memset(d, 0, sizeof *d);
    if (n == 0) return;
    kv_t *kv = malloc(sizeof(kv_t) * n);
    for (uint32_t i = 0; i < n; i++) { kv[i].key = keys[i]; kv[i].id = i; }
    qsort(kv, n, sizeof(kv_t), kv_cmp);
    uint32_t nk = 0;
    for (uint32_t i = 0; i < n; i++) if (i == 0 || kv[i].key != kv[i - 1].key) nk++;
    d->nkeys = nk;
    d->keys = malloc(8 * (size_t)nk); d->start = malloc(4 * ((size_t)nk + 1)); d->ids = malloc(4 * (size_t)n);
    d->live_end = malloc(4 * (size_t)nk); d->binof = malloc(4 * (size_t)n); d->nlive = malloc(4 * (size_t)nk);
    uint32_t b = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (i == 0 || kv[i].key != kv[i - 1].key) { d->keys[b] = kv[i].key; d->start[b] = i; b++; }
        d->ids[i] = kv[i].id; d->binof[kv[i].id] = b - 1;
    }
    d->start[nk] = n;
    for (uint32_t i = 0; i < nk; i++) { d->live_end[i] = d->start[i + 1]; d->nlive[i] = d->start[i + 1] - d->start[i]; }
    uint32_t cap = 16; while (cap < 2 * nk) cap <<= 1;
    d->hmask = cap - 1; d->hslot = calloc(cap, 4);
    for (uint32_t i = 0; i < nk; i++) {
        uint32_t h = (uint32_t)mix64(d->keys[i]) & d->hmask;
        while (d->hslot[h]) h = (h + 1) & d->hmask;
        d->hslot[h] = i + 1;
    }
    free(kv);
}
static void dict_free(dict_t *d) { free(d->keys); free(d->start); free(d->ids); free(d->live_end); free(d->hslot); free(d->binof); free(d->nlive); memset(d, 0, sizeof *d); }
static uint32_t dict_lookup(const dict_t *d, uint64_t key)
{
    if (!d->nkeys) return NONE;
    uint32_t h = (uint32_t)mix64(key) & d->hmask;
    while (d->hslot[h]) { uint32_t b = d->hslot[h] - 1; if (d->keys[b] == key) return b; h = (h + 1) & d->hmask; }
    return NONE;
}

/* ------------------------------------------------------------------ stage I */
typedef struct { uint32_t *v; size_t n, cap; } vec32;
static void vpush(vec32 *a, uint32_t x) { if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 16; a->v = realloc(a->v, 4 * a->cap); } a->v[a->n++] = x; }

typedef struct {
    int active; uint32_t id, cur, prev; int prev_unmatched;   /* id: the chain's number (its seed, its place in the output, its scan start in large bins) */
    int32_t *count;        /* [4][L], rows A C G T (reorder.cpp:467) */
    uint8_t *cons;         /* consensus, idx codes A0 C1 G2 T3 */
    vec32 m_order, m_meta; /* main stream: order; meta = pos | flag<<8 | rc<<9 */
    vec32 s_order;         /* singleton stream */
    uint32_t p_rid; int p_j, p_dir;   /* proposal of the current step */
    int p_stalled, p_next, resume_p;  /* the step was put off (STEP_CAP) before probe number p_next of its priority order; where the chain's NEXT step takes up again (0: at the start) */
    int p_big;                        /* probes of the step, up to and including the winning one, into bins of more than LARGEBIN reads that still had unclaimed ones */
    /* super-round: up to MAXSTEPS speculative steps against the frozen claim state */
    uint32_t s_rid[MAXSTEPS]; uint8_t s_j[MAXSTEPS], s_dir[MAXSTEPS], s_kind[MAXSTEPS], s_sidx[MAXSTEPS]; int nsteps, need_reseed;
    uint32_t sugg[NSUGG]; int nsugg, sugg_pos;   /* unclaimed ids right below the cursor at the chain's last reseed, highest first */
    int32_t *count0; uint8_t *cons0;  /* state at the start of the super-round (rollback point) */
    int sleep, ncut;                  /* back-off (BO_*): super-rounds the chain still sits out; its walks cut in a row */
} chain_t;

typedef struct {
    uint32_t M, S, unmatched;
    uint32_t *order; uint8_t *flag, *pos, *rc;     /* main stream, M entries; flag/rc as ASCII chars */
    uint32_t *order_s;                             /* singleton stream, S entries */
    uint64_t rounds, probes, cands, conflicts;
} stage1_out_t;

static void cons_reset(chain_t *c, const uint64_t *r, int L)     /* reorder.cpp:875-883 */
{
    memset(c->count, 0, sizeof(int32_t) * 4 * L);
    for (int i = 0; i < L; i++) {
        int v = pc_to_idx[(r[(2 * i) >> 6] >> ((2 * i) & 63)) & 3];
        c->count[v * L + i] = 1; c->cons[i] = (uint8_t)v;
    }
}
static void cons_update(chain_t *c, const uint64_t *r, int L, int rev, int shift)    /* reorder.cpp:884-909 */
{
    uint8_t cur[MAXL];
    for (int i = 0; i < L; i++) {
        int v = pc_to_idx[(r[(2 * i) >> 6] >> ((2 * i) & 63)) & 3];
        if (!rev) cur[i] = (uint8_t)v; else cur[L - 1 - i] = (uint8_t)(3 - v);
    }
    for (int i = 0; i < L - shift; i++) {
        int max = 0, ind = 0;
        for (int j = 0; j < 4; j++) c->count[j * L + i] = c->count[j * L + i + shift];
        c->count[cur[i] * L + i] += 1;
        for (int j = 0; j < 4; j++) if (c->count[j * L + i] > max) { max = c->count[j * L + i]; ind = j; }
        c->cons[i] = (uint8_t)ind;
    }
    for (int i = L - shift; i < L; i++) {
        for (int j = 0; j < 4; j++) c->count[j * L + i] = 0;
        c->count[cur[i] * L + i] = 1; c->cons[i] = cur[i];
    }
}

/* scan one bin: ids from highest to lowest, only unclaimed ones, at most maxsearch of them (reorder.cpp:540).
 * rot (schedule rule of round 3, bins of more than LARGEBIN reads only; 0 for every other bin and for chain 0, so that one chain is the
 * reference at -t 1): the scan starts at the rot-th unclaimed entry from the top, goes down, and takes the rot entries above it last.
 * Chains that sit in the same repeat all wanted the SAME read -- the highest unclaimed id of the bin -- and all but one lost their bid and
 * the rest of their walk (a third of all walked steps on an exact-repeat family); a start that depends on the chain spreads them over
 * the bin.  rot = (chain * 0x9E3779B1 >> 8) mod min(unclaimed entries of the bin, maxsearch), in the frozen state of the super-round. */
static uint32_t scan_bin(dict_t *d, uint32_t bin, const uint64_t *reads, int W, const uint8_t *claimed,
                         const uint64_t *refsh, const uint64_t *mask, int thresh, int maxsearch, uint64_t *cands,
                         const uint32_t *own, int nown, uint32_t rot)
{
    uint32_t s = d->start[bin], e = d->live_end[bin];
    while (e > s && claimed[d->ids[e - 1]]) e--;
    d->live_end[bin] = e;
    int seen = 0;
    for (int phase = 0; phase < 2; phase++) {                     /* 0: from the rot-th unclaimed entry down; 1: the rot entries above it */
        if (phase == 1 && rot == 0) break;
        uint32_t u = 0;                                           /* unclaimed entries passed so far, from the top */
        for (uint32_t i = e; i > s && seen < maxsearch; i--) {
            uint32_t rid = d->ids[i - 1];
            if (claimed[rid]) continue;
            const uint32_t ui = u++;
            if (phase == 0 ? ui < rot : ui >= rot) { if (phase == 1) break; continue; }
            int mine = 0;                                         /* reads this chain already took earlier in the same super-round */
            for (int k = 0; k < nown; k++) if (own[k] == rid) mine = 1;
            if (mine) continue;
            seen++; (*cands)++;
            const uint64_t *r = reads + (size_t)rid * W;
            int hd = 0;
            for (int w = 0; w < W; w++) hd += __builtin_popcountll(refsh[w] ^ (r[w] & mask[w]));
            if (hd <= thresh) return rid;
        }
    }
    return NONE;
}

static uint32_t scan_rot(uint32_t chain, uint32_t nlive, int maxsearch)
{
    const uint32_t m = nlive < (uint32_t)maxsearch ? nlive : (uint32_t)maxsearch;
    return m ? (uint32_t)((chain * 0x9E3779B1u) >> 8) % m : 0;
}

static void propose(chain_t *c, dict_t *dict, const uint64_t *reads, const uint8_t *claimed, const params_t *p,
                    const uint64_t *mask, const uint64_t *revmask, stage1_out_t *st, const uint32_t *own, int nown)
{
    int L = p->L, W = p->W;
    uint64_t ref[MAXW] = { 0 }, rev[MAXW] = { 0 }, topmask;
    for (int i = 0; i < L; i++) {
        ref[(2 * i) >> 6] |= (uint64_t)idx_to_pc[c->cons[i]] << ((2 * i) & 63);
        int k = L - 1 - i;                                        /* revref[k] = comp(cons[i]) */
        rev[(2 * k) >> 6] |= (uint64_t)idx_to_pc[3 - c->cons[i]] << ((2 * k) & 63);
    }
    topmask = ((2 * L) & 63) ? ((((uint64_t)1) << ((2 * L) & 63)) - 1) : ~(uint64_t)0;
    c->p_rid = NONE; c->p_big = 0; c->p_stalled = 0;
    int pidx = 0;                                                 /* number of the probe in the priority order of the step */
    for (int j = 0; j < p->maxmatch; j++) {
        for (int l = 0; l < 2; l++) {                             /* forward, reorder.cpp:520-580 */
            if (p->de[l] + j >= L) continue;
            if (pidx++ < c->resume_p) continue;                   /* made in an earlier super-round, before the step was put off */
            uint64_t key = extract_bits(ref, W, 2 * p->ds[l], p->kbits[l]);
            st->probes++;
            uint32_t bin = dict_lookup(&dict[l], key);
            if (bin == NONE) continue;
            int big = dict[l].start[bin + 1] - dict[l].start[bin] > LARGEBIN && dict[l].nlive[bin] > 0;
            if (big) c->p_big++;
            uint32_t rid = scan_bin(&dict[l], bin, reads, W, claimed, ref, mask + (size_t)j * W, p->thresh, p->maxsearch, &st->cands, own, nown, big ? scan_rot(c->id, dict[l].nlive[bin], p->maxsearch) : 0);
            if (rid != NONE) { c->p_rid = rid; c->p_j = j; c->p_dir = 0; return; }
            if (big && c->p_big >= STEP_CAP) { c->p_stalled = 1; c->p_next = pidx; return; }
        }
        for (int l = 0; l < 2; l++) {                             /* reverse, reorder.cpp:585-643 */
            if (p->ds[l] <= j) continue;
            if (pidx++ < c->resume_p) continue;
            uint64_t key = extract_bits(rev, W, 2 * p->ds[l], p->kbits[l]);
            st->probes++;
            uint32_t bin = dict_lookup(&dict[l], key);
            if (bin == NONE) continue;
            int big = dict[l].start[bin + 1] - dict[l].start[bin] > LARGEBIN && dict[l].nlive[bin] > 0;
            if (big) c->p_big++;
            uint32_t rid = scan_bin(&dict[l], bin, reads, W, claimed, rev, revmask + (size_t)j * W, p->thresh, p->maxsearch, &st->cands, own, nown, big ? scan_rot(c->id, dict[l].nlive[bin], p->maxsearch) : 0);
            if (rid != NONE) { c->p_rid = rid; c->p_j = j; c->p_dir = 1; return; }
            if (big && c->p_big >= STEP_CAP) { c->p_stalled = 1; c->p_next = pidx; return; }
        }
        /* revref <<= 2; ref >>= 2  (reorder.cpp:647-648), bitset<2L> semantics */
        for (int w = W - 1; w > 0; w--) rev[w] = (rev[w] << 2) | (rev[w - 1] >> 62);
        rev[0] <<= 2; rev[W - 1] &= topmask;
        for (int w = 0; w < W - 1; w++) ref[w] = (ref[w] >> 2) | (ref[w + 1] << 62);
        ref[W - 1] >>= 2;
    }
}

static void emit_main(chain_t *c, uint32_t order, int flag, int pos, int rc)
{
    vpush(&c->m_order, order);
    vpush(&c->m_meta, (uint32_t)pos | ((uint32_t)flag << 8) | ((uint32_t)rc << 9));
}

/* reads: N x W packed words.  K chains, S speculative steps per super-round.
 *
 * One super-round: (A) every live chain walks up to S steps on its own against the FROZEN claim state (its own picks of this
 * super-round excluded), bidding (step, chain) for every read it takes; it stops early when a step finds no candidate.
 * (B) a read goes to the smallest (step, chain) bid; a chain keeps its steps up to (not including) the first one it did not
 * win, rolls its consensus back to that point and retries from there in the next super-round.  (C) chains that ran out of
 * candidates without losing a bid reseed, in chain order, from the one global descending cursor.
 * K=1 is the reference at num_thr=1 for every S (nothing to lose, the chain just runs until it needs a new seed); S=1 is the
 * plain round-synchronous schedule. */
static int stage1_run(const uint64_t *reads, uint32_t N, const params_t *p, uint32_t K, uint32_t nsteps, stage1_out_t *out)
{
    int L = p->L, W = p->W;
    memset(out, 0, sizeof *out);
    if (K == 0) K = 1;
    if (nsteps == 0) nsteps = 1;
    if (nsteps > MAXSTEPS) nsteps = MAXSTEPS;
    dict_t dict[2];
    uint64_t *keys = malloc(8 * ((size_t)N + 1));
    for (int l = 0; l < 2; l++) {
        for (uint32_t i = 0; i < N; i++) keys[i] = extract_bits(reads + (size_t)i * W, W, 2 * p->ds[l], p->kbits[l]);
        dict_build(&dict[l], keys, N);
    }
    free(keys);
    /* the back-off applies where the library applies it: more than 16 384 chains and the bins of more than LARGEBIN reads hold more than 2 % of N entries */
    int backoff = 0;
    {
        uint64_t large_entries = 0;
        for (int l = 0; l < 2; l++) for (uint32_t b = 0; b < dict[l].nkeys; b++) { uint32_t nb = dict[l].start[b + 1] - dict[l].start[b]; if (nb > LARGEBIN) large_entries += nb; }
        backoff = K > 16384 && large_entries * 50 > (uint64_t)N;
    }
    /* generatemasks (reorder.cpp:706-718) */
    uint64_t *mask = calloc((size_t)p->maxmatch * W + 1, 8), *revmask = calloc((size_t)p->maxmatch * W + 1, 8);
    for (int j = 0; j < p->maxmatch; j++) {
        for (int b = 0; b < 2 * L - 2 * j; b++) mask[(size_t)j * W + (b >> 6)] |= (uint64_t)1 << (b & 63);
        for (int b = 2 * j; b < 2 * L; b++) revmask[(size_t)j * W + (b >> 6)] |= (uint64_t)1 << (b & 63);
    }
    uint8_t *claimed = calloc((size_t)N + 1, 1);
#define CLAIM(r) do { uint32_t r_ = (r); claimed[r_] = 1; dict[0].nlive[dict[0].binof[r_]]--; dict[1].nlive[dict[1].binof[r_]]--; } while (0)
    uint32_t *bid = malloc(4 * ((size_t)N + 1));
    for (uint32_t i = 0; i < N; i++) bid[i] = NONE;
    chain_t *ch = calloc(K, sizeof(chain_t));
    for (uint32_t c = 0; c < (uint32_t)K; c++) ch[c].id = c;
    uint32_t firstread = 0, nactive = 0;
    for (uint32_t c = 0; c < K; c++) {                            /* reorder.cpp:476-497 */
        chain_t *x = &ch[c];
        x->count = malloc(sizeof(int32_t) * 4 * L); x->cons = malloc(L);
        x->count0 = malloc(sizeof(int32_t) * 4 * L); x->cons0 = malloc(L);
        uint32_t cur = firstread;
        if (N == 0 || claimed[cur]) x->active = 0;
        else {
            CLAIM(cur); out->unmatched++; x->active = 1; x->cur = cur; x->prev = cur; x->prev_unmatched = 1;
            cons_reset(x, reads + (size_t)cur * W, L); nactive++;
        }
        firstread += N / K;
    }
    int64_t remainingpos = (int64_t)N - 1;
    while (nactive) {
        out->rounds++;
        /* (A) speculative steps against the frozen state */
        for (uint32_t c = 0; c < K; c++) {
            chain_t *x = &ch[c];
            if (!x->active || x->sleep) continue;                  /* a chain that sits the round out is not touched by it */
            memcpy(x->count0, x->count, sizeof(int32_t) * 4 * L); memcpy(x->cons0, x->cons, L);
            x->nsteps = 0; x->need_reseed = 0;
            int spos = x->sugg_pos, bigprobes = 0;
            for (uint32_t t = 0; t < nsteps; t++) {
                propose(x, dict, reads, claimed, p, mask, revmask, out, x->s_rid, x->nsteps);
                if (x->p_stalled) { x->resume_p = x->p_next; break; }             /* the step is put off: the walk ends in front of it */
                x->resume_p = 0;                                   /* the step is made (a read or a new seed): the next one starts at its first probe */
                uint32_t key = (t << 20) | c;
                if (x->p_rid != NONE) {
                    x->s_rid[t] = x->p_rid; x->s_j[t] = (uint8_t)x->p_j; x->s_dir[t] = (uint8_t)x->p_dir; x->s_kind[t] = 0; x->s_sidx[t] = 0;
                    x->nsteps = (int)t + 1;
                    if (key < bid[x->p_rid]) bid[x->p_rid] = key;
                    cons_update(x, reads + (size_t)x->p_rid * W, L, x->p_dir, x->p_j);
                    bigprobes += x->p_big;
                    if (bigprobes >= SCAN_BUDGET) break;              /* schedule rule: the budget of probes into large bins */
                    continue;
                }
                /* no candidate: continue from the chain's look-ahead seeds, highest id first, skipping what got claimed meanwhile
                   (for K=1 this is exactly "the highest unclaimed id below the previous seed", reorder.cpp:652-668) */
                uint32_t sid = NONE;
                while (spos < x->nsugg) {
                    uint32_t id = x->sugg[spos++];
                    int mine = 0;
                    for (int k = 0; k < x->nsteps; k++) if (x->s_rid[k] == id) mine = 1;
                    if (!claimed[id] && !mine) { sid = id; break; }
                }
                if (sid == NONE) { x->need_reseed = 1; break; }
                x->s_rid[t] = sid; x->s_j[t] = 0; x->s_dir[t] = 0; x->s_kind[t] = 1; x->s_sidx[t] = (uint8_t)spos;
                x->nsteps = (int)t + 1;
                if (key < bid[sid]) bid[sid] = key;
                cons_reset(x, reads + (size_t)sid * W, L);
                bigprobes += x->p_big;
                if (bigprobes >= SCAN_BUDGET) break;
            }
            x->p_j = spos;                                        /* scratch: look-ahead position reached by this walk */
        }
        /* (B) keep the steps before the first lost bid (reorder.cpp:560-578 / :624-642 for each kept step) */
        for (uint32_t c = 0; c < K; c++) {
            chain_t *x = &ch[c];
            if (!x->active) continue;
            if (x->sleep) { x->sleep--; continue; }
            int v = 0;
            while (v < x->nsteps && bid[x->s_rid[v]] == (((uint32_t)v << 20) | c)) v++;
            if (v < x->nsteps) {                                  /* lost a bid: roll back and replay the kept steps */
                out->conflicts++; x->need_reseed = 0; x->resume_p = 0;     /* rolled back: a step put off belonged to a state that is gone */
                memcpy(x->count, x->count0, sizeof(int32_t) * 4 * L); memcpy(x->cons, x->cons0, L);
                for (int t = 0; t < v; t++) {
                    if (x->s_kind[t]) cons_reset(x, reads + (size_t)x->s_rid[t] * W, L);
                    else cons_update(x, reads + (size_t)x->s_rid[t] * W, L, x->s_dir[t], x->s_j[t]);
                }
                for (int t = v - 1; t >= 0; t--) if (x->s_kind[t]) { x->sugg_pos = x->s_sidx[t]; break; }   /* look-ahead seeds of dropped steps stay available */
            } else x->sugg_pos = x->p_j;
            if (backoff) {
                if (v < x->nsteps) { if (x->ncut < BO_FREE + BO_CAP) x->ncut++; x->sleep = x->ncut > BO_FREE ? (1 << (x->ncut - BO_FREE)) - 1 : 0; }
                else if (x->nsteps > 0) x->ncut = 0;
            }
            for (int t = 0; t < v; t++) {
                uint32_t k = x->s_rid[t];
                CLAIM(k); x->cur = k;
                if (x->s_kind[t]) {                               /* a new seed (reorder.cpp:678-687) */
                    if (x->prev_unmatched) vpush(&x->s_order, x->prev);
                    out->unmatched++; x->prev_unmatched = 1; x->prev = k;
                } else {
                    if (x->prev_unmatched) emit_main(x, x->prev, 0, L & 0xFF, 0);
                    emit_main(x, k, 1, x->s_j[t], x->s_dir[t]);
                    x->prev_unmatched = 0;
                }
            }
        }
        for (uint32_t c = 0; c < K; c++) if (ch[c].active) for (int t = 0; t < ch[c].nsteps; t++) bid[ch[c].s_rid[t]] = NONE;
        /* (C) reseed from the global descending cursor (reorder.cpp:650-688) */
        for (uint32_t c = 0; c < K; c++) {
            chain_t *x = &ch[c];
            if (!x->active || !x->need_reseed) continue;
            int found = 0;
            while (remainingpos >= 0) {
                if (!claimed[remainingpos]) { found = 1; break; }
                remainingpos--;
            }
            if (x->prev_unmatched) vpush(&x->s_order, x->prev);
            x->nsugg = 0; x->sugg_pos = 0;
            if (!found) { x->active = 0; nactive--; x->need_reseed = 0; continue; }
            uint32_t cur = (uint32_t)remainingpos; remainingpos--;
            CLAIM(cur); out->unmatched++;
            x->cur = cur; cons_reset(x, reads + (size_t)cur * W, L);
            x->prev_unmatched = 1; x->prev = cur;
            x->need_reseed = 2;                                   /* got a seed: entitled to look-ahead seeds below */
        }
        /* look-ahead: the next NSUGG unclaimed ids below the cursor for the first reseeded chain, the NSUGG after those for the
           second, ...; nothing is claimed, and the cursor moves below the last one handed out: a look-ahead read stays with its
           chain until the chain takes it or finds it claimed by a walk (later reseeds do not hand it out again) */
        {
            int64_t look = remainingpos, last_taken = -1;
            /* the look-ahead only inspects LOOK_CHUNKS bitmap chunks of the GPU's k_reseed: 1024 64-bit words each, ending at the cursor's word */
            int64_t lim = remainingpos >= 0 ? ((remainingpos >> 6) - (1024 * (int64_t)LOOK_CHUNKS - 1)) * 64 : 0;
            if (lim < 0) lim = 0;
            for (uint32_t c = 0; c < K; c++) {
                chain_t *x = &ch[c];
                if (!x->active || x->need_reseed != 2) continue;
                x->need_reseed = 0;
                while (x->nsugg < NSUGG && look >= lim) { if (!claimed[look]) { x->sugg[x->nsugg++] = (uint32_t)look; last_taken = look; } look--; }
            }
            if (last_taken >= 0) remainingpos = last_taken - 1;
        }
    }
    /* concatenate per-chain streams in chain order (reorder.cpp:778-821) */
    size_t M = 0, S = 0;
    for (uint32_t c = 0; c < K; c++) { M += ch[c].m_order.n; S += ch[c].s_order.n; }
    out->M = (uint32_t)M; out->S = (uint32_t)S;
    out->order = malloc(4 * (M + 1)); out->flag = malloc(M + 1); out->pos = malloc(M + 1); out->rc = malloc(M + 1);
    out->order_s = malloc(4 * (S + 1));
    M = S = 0;
    for (uint32_t c = 0; c < K; c++) {
        for (size_t i = 0; i < ch[c].m_order.n; i++, M++) {
            uint32_t m = ch[c].m_meta.v[i];
            out->order[M] = ch[c].m_order.v[i]; out->pos[M] = (uint8_t)(m & 0xFF);
            out->flag[M] = (m >> 8) & 1 ? '1' : '0'; out->rc[M] = (m >> 9) & 1 ? 'r' : 'd';
        }
        for (size_t i = 0; i < ch[c].s_order.n; i++) out->order_s[S++] = ch[c].s_order.v[i];
        free(ch[c].count); free(ch[c].cons); free(ch[c].count0); free(ch[c].cons0); free(ch[c].m_order.v); free(ch[c].m_meta.v); free(ch[c].s_order.v);
    }
    free(ch); free(claimed); free(bid); free(mask); free(revmask);
    dict_free(&dict[0]); dict_free(&dict[1]);
    return 0;
}
static void stage1_free(stage1_out_t *o) { free(o->order); free(o->flag); free(o->pos); free(o->rc); free(o->order_s); memset(o, 0, sizeof *o); }

/* ------------------------------------------------------------------ small file helpers */
static char *path_join(const char *dir, const char *name)
{
    size_t n = strlen(dir) + strlen(name) + 16;
    char *s = malloc(n); snprintf(s, n, "%s/output/%s", dir, name); return s;
}
static uint8_t *slurp(const char *dir, const char *name, size_t *len)
{
    char *p = path_join(dir, name); FILE *f = fopen(p, "rb"); free(p);
    *len = 0; if (!f) return calloc(1, 1);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc((size_t)n + 1); if (n && fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return calloc(1, 1); }
    b[n] = 0; fclose(f); *len = (size_t)n; return b;
}
static int spit(const char *dir, const char *name, const void *buf, size_t len, const char *mode)
{
    char *p = path_join(dir, name); FILE *f = fopen(p, mode); free(p);
    if (!f) return -1;
    if (len) fwrite(buf, 1, len, f);
    fclose(f); return 0;
}
typedef struct { uint8_t *v; size_t n, cap; } bytes;
static void bput(bytes *b, const void *src, size_t n)
{
    if (b->n + n > b->cap) { while (b->n + n > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->v = realloc(b->v, b->cap); }
    memcpy(b->v + b->n, src, n); b->n += n;
}
static void bputc(bytes *b, char c) { bput(b, &c, 1); }

/* ------------------------------------------------------------------ stage I, file contract of reorder.out <basedir> */
int harc_oracle_reorder(const char *basedir, int L, uint32_t K, uint32_t S, uint32_t *unmatched_out, uint64_t *stats4)
{
    if (L < 1 || L > MAXL) return -2;
    params_t p; params_init(&p, L);
    size_t n; uint8_t *nb = slurp(basedir, "numreads.bin", &n);
    if (n < 4) { free(nb); return -1; }
    uint32_t N; memcpy(&N, nb, 4); free(nb);
    size_t flen; uint8_t *txt = slurp(basedir, "input_clean.dna", &flen);
    if (flen < (size_t)N * (L + 1)) { free(txt); return -1; }
    uint64_t *reads = calloc((size_t)N * p.W + 1, 8);
    for (uint32_t i = 0; i < N; i++) pack_read((const char *)txt + (size_t)i * (L + 1), L, p.W, reads + (size_t)i * p.W);   /* reorder.cpp:252 stride */
    free(txt);
    stage1_out_t o; stage1_run(reads, N, &p, K, S, &o);
    /* writetofile (reorder.cpp:722-830) */
    bytes dna = { 0 }, dna_s = { 0 };
    char s[MAXL + 2];
    for (uint32_t i = 0; i < o.M; i++) { unpack_read(reads + (size_t)o.order[i] * p.W, L, s, o.rc[i] == 'r'); s[L] = '\n'; bput(&dna, s, L + 1); }
    for (uint32_t i = 0; i < o.S; i++) { unpack_read(reads + (size_t)o.order_s[i] * p.W, L, s, 0); s[L] = '\n'; bput(&dna_s, s, L + 1); }
    spit(basedir, "temp.dna", dna.v, dna.n, "wb"); spit(basedir, "temp.dna.singleton", dna_s.v, dna_s.n, "wb");
    spit(basedir, "read_rev.txt", o.rc, o.M, "wb"); spit(basedir, "tempflag.txt", o.flag, o.M, "wb");
    spit(basedir, "temppos.txt", o.pos, o.M, "wb"); spit(basedir, "read_order.bin", o.order, 4 * (size_t)o.M, "wb");
    spit(basedir, "read_order.bin.singleton", o.order_s, 4 * (size_t)o.S, "wb");
    if (unmatched_out) *unmatched_out = o.unmatched;
    if (stats4) { stats4[0] = o.rounds; stats4[1] = o.probes; stats4[2] = o.cands; stats4[3] = o.conflicts; }
    free(dna.v); free(dna_s.v); free(reads); stage1_free(&o);
    return 0;
}

/* in-memory stage I for timing (cpu_baseline "port"): reads = N*L ASCII, no separators */
int harc_oracle_stage1_mem(const char *ascii, uint32_t N, int L, uint32_t K, uint32_t nsteps, uint32_t *order, uint8_t *flag, uint8_t *pos,
                           uint8_t *rc, uint32_t *M, uint32_t *order_s, uint32_t *S, uint32_t *unmatched, uint64_t *stats4)
{
    params_t p; params_init(&p, L);
    uint64_t *reads = calloc((size_t)N * p.W + 1, 8);
    for (uint32_t i = 0; i < N; i++) pack_read(ascii + (size_t)i * L, L, p.W, reads + (size_t)i * p.W);
    stage1_out_t o; stage1_run(reads, N, &p, K, nsteps, &o);
    if (order) memcpy(order, o.order, 4 * (size_t)o.M);
    if (flag) memcpy(flag, o.flag, o.M);
    if (pos) memcpy(pos, o.pos, o.M);
    if (rc) memcpy(rc, o.rc, o.M);
    if (order_s) memcpy(order_s, o.order_s, 4 * (size_t)o.S);
    if (M) *M = o.M; if (S) *S = o.S; if (unmatched) *unmatched = o.unmatched;
    if (stats4) { stats4[0] = o.rounds; stats4[1] = o.probes; stats4[2] = o.cands; stats4[3] = o.conflicts; }
    free(reads); stage1_free(&o);
    return 0;
}

/* ------------------------------------------------------------------ stage II (encoder.cpp) */
/* 3-bit code, bit 3i = LSB: A=000 C=100 G=010 T=110 N=001  => integers A0 N1 G2 C4 T6 (encoder.cpp:729-749) */
static int c3_of(char c) { switch (c) { case 'A': return 0; case 'N': return 1; case 'G': return 2; case 'C': return 4; case 'T': return 6; } return 0; }
static char comp_of(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; } return 'N'; }
static char enc_noise_of(char ref, char rd)                          /* encoder.cpp:751-771 */
{
    static const char *row_A = "CGTN", *row_C = "AGTN", *row_G = "TACN", *row_T = "GCAN", *row_N = "AGCT";
    const char *row = ref == 'A' ? row_A : ref == 'C' ? row_C : ref == 'G' ? row_G : ref == 'T' ? row_T : row_N;
    for (int i = 0; i < 4; i++) if (row[i] == rd) return (char)('0' + i);
    return 0;
}
static int ham3(const char *a, const char *b, int L)
{
    int h = 0;
    for (int i = 0; i < L; i++) h += __builtin_popcount((unsigned)(c3_of(a[i]) ^ c3_of(b[i])));
    return h;
}

typedef struct { int64_t pos; const char *read; int owned; uint32_t order; char rc; } centry;   /* one read of a contig */
typedef struct { centry *v; size_t n, cap; } clist;
static void cl_push(clist *c, centry e) { if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->v = realloc(c->v, sizeof(centry) * c->cap); } c->v[c->n++] = e; }

typedef struct { bytes seq, pos, noise, noisepos, order, order_N, rc; } shard_out;

static char *buildcontig(const clist *c, int L, size_t *len_out)     /* encoder.cpp:619-652; pos = deltas, first ignored */
{
    size_t len = L;
    for (size_t i = 1; i < c->n; i++) len += (size_t)c->v[i].pos;
    char *ref = malloc(len + 1);
    if (c->n == 1) { memcpy(ref, c->v[0].read, L); ref[L] = 0; *len_out = L; return ref; }
    int32_t *cnt = calloc(len * 4, sizeof(int32_t));
    size_t cur = 0;
    for (size_t i = 0; i < c->n; i++) {
        if (i) cur += (size_t)c->v[i].pos;
        for (int k = 0; k < L; k++) {
            char ch = c->v[i].read[k];
            int v = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
            cnt[(cur + k) * 4 + v]++;
        }
    }
    for (size_t i = 0; i < len; i++) {
        int max = 0, ind = 0;
        for (int j = 0; j < 4; j++) if (cnt[i * 4 + j] > max) { max = cnt[i * 4 + j]; ind = j; }
        ref[i] = "ACGT"[ind];
    }
    ref[len] = 0; free(cnt); *len_out = len; return ref;
}

static void writecontig(const char *ref, const clist *c, int L, shard_out *o)      /* encoder.cpp:654-717 */
{
    bput(&o->seq, ref, strlen(ref));
    if (c->n == 1) {
        bputc(&o->noise, '\n'); bputc(&o->pos, (char)L);
        bput(&o->order, &c->v[0].order, 4); bputc(&o->rc, c->v[0].rc);
        return;
    }
    int64_t cur = 0;
    for (size_t i = 0; i < c->n; i++) {
        const centry *e = &c->v[i];
        if (i) cur += e->pos;
        int prevj = 0;
        for (int j = 0; j < L; j++) if (e->read[j] != ref[cur + j]) {
            bputc(&o->noise, enc_noise_of(ref[cur + j], e->read[j]));
            bputc(&o->noisepos, (char)(j - prevj)); prevj = j;
        }
        bputc(&o->noise, '\n');
        bputc(&o->pos, i == 0 ? (char)L : (char)e->pos);
        if (memchr(e->read, 'N', L)) bput(&o->order_N, &e->order, 4); else bput(&o->order, &e->order, 4);
        bputc(&o->rc, e->rc);
    }
}

static uint64_t key3(const char *s, int ds, int de)
{
    uint64_t k = 0;
    for (int i = ds; i <= de; i++) k |= (uint64_t)c3_of(s[i]) << (3 * (i - ds));
    return k;
}

/* realign singletons / N reads to one contig (encoder.cpp:231-418). rd = strings of the S+NN candidates. */
static void realign(clist *c, const char *ref, size_t reflen, int L, dict_t *dict, const int *ds, const int *de,
                    const char *rd, const uint32_t *order_s, uint8_t *claimed, int thresh_s, int maxsearch)
{
    /* pos -> cumulative with first = 0 (encoder.cpp:243-251) */
    c->v[0].pos = 0;
    for (size_t i = 1; i < c->n; i++) c->v[i].pos += c->v[i - 1].pos;
    clist out = { 0 };
    size_t it = 0, norig = c->n;
    char win[MAXL + 1], rwin[MAXL + 1];
    for (size_t j = 0; j + L <= reflen; j++) {
        while (it < norig && c->v[it].pos <= (int64_t)j) cl_push(&out, c->v[it++]);   /* originals with cum pos <= j first (:254-268) */
        memcpy(win, ref + j, L);
        for (int i = 0; i < L; i++) rwin[i] = comp_of(ref[j + L - 1 - i]);
        for (int dir = 0; dir < 2; dir++) {
            const char *w = dir ? rwin : win;
            for (int l = 0; l < 2; l++) {
                uint32_t bin = dict_lookup(&dict[l], key3(w, ds[l], de[l]));
                if (bin == NONE) continue;
                dict_t *d = &dict[l];
                uint32_t s = d->start[bin], e = d->live_end[bin];
                while (e > s && claimed[d->ids[e - 1]]) e--;
                d->live_end[bin] = e;
                int seen = 0;
                for (uint32_t i = e; i > s && seen < maxsearch; i--) {     /* window = top maxsearch LIVE ids at scan start (:293) */
                    uint32_t rid = d->ids[i - 1];
                    if (claimed[rid]) continue;
                    seen++;
                    const char *r = rd + (size_t)rid * L;
                    if (ham3(w, r, L) <= thresh_s) {                       /* every passing candidate is taken (:296-317) */
                        claimed[rid] = 1;                                  /* stays counted in this scan's window (seen++ above) */
                        centry en; en.pos = (int64_t)j; en.order = order_s[rid]; en.rc = dir ? 'r' : 'd'; en.owned = 1;
                        char *cp = malloc(L);
                        if (!dir) memcpy(cp, r, L); else for (int k = 0; k < L; k++) cp[k] = comp_of(r[L - 1 - k]);
                        en.read = cp; cl_push(&out, en);
                    }
                }
            }
        }
    }
    while (it < norig) cl_push(&out, c->v[it++]);
    /* back to deltas (:411-417) */
    int64_t prev = 0;
    for (size_t i = 0; i < out.n; i++) { int64_t cum = out.v[i].pos; out.v[i].pos = cum - prev; prev = cum; }
    free(c->v); *c = out;
}

static void packbits_seq(const bytes *txt, bytes *bin, bytes *tail)       /* encoder.cpp:527-548: A0 C1 G2 T3, base0 low bits */
{
    size_t n4 = txt->n / 4;
    for (size_t i = 0; i < n4; i++) {
        uint8_t b = 0;
        for (int k = 0; k < 4; k++) { char ch = (char)txt->v[4 * i + k]; int v = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; b |= (uint8_t)(v << (2 * k)); }
        bputc(bin, (char)b);
    }
    bput(tail, txt->v + 4 * n4, txt->n % 4);
}
static void packbits_rev(const bytes *txt, bytes *bin, bytes *tail)       /* encoder.cpp:560-578 */
{
    size_t n8 = txt->n / 8;
    for (size_t i = 0; i < n8; i++) {
        uint8_t b = 0;
        for (int k = 0; k < 8; k++) if (txt->v[8 * i + k] == 'r') b |= (uint8_t)(1u << k);
        bputc(bin, (char)b);
    }
    bput(tail, txt->v + 8 * n8, txt->n % 8);
}

/* file contract of encoder.out <basedir>; E = num_thr */
int harc_oracle_encoder(const char *basedir, int L, uint32_t E, uint32_t *matched_s_out, uint32_t *matched_N_out)
{
    if (L < 1 || L > MAXL || E == 0) return -2;
    size_t n_dna, n_flag, n_pos, n_order, n_rc, n_sing, n_os, n_N;
    uint8_t *dna = slurp(basedir, "temp.dna", &n_dna), *flag = slurp(basedir, "tempflag.txt", &n_flag),
            *pos = slurp(basedir, "temppos.txt", &n_pos), *ord = slurp(basedir, "read_order.bin", &n_order),
            *rc = slurp(basedir, "read_rev.txt", &n_rc), *sing = slurp(basedir, "temp.dna.singleton", &n_sing),
            *os = slurp(basedir, "read_order.bin.singleton", &n_os), *Ntxt = slurp(basedir, "input_N.dna", &n_N);
    uint32_t M = (uint32_t)(n_order / 4), S = (uint32_t)(n_sing / (L + 1)), NN = (uint32_t)(n_N / (L + 1));   /* getDataParams :781-813 */
    uint32_t T = S + NN;
    /* candidate strings + order_s (readsingletons :823-872) */
    char *rd = malloc((size_t)T * L + 1); uint32_t *order_s = malloc(4 * ((size_t)T + 1));
    for (uint32_t i = 0; i < S; i++) { memcpy(rd + (size_t)i * L, sing + (size_t)i * (L + 1), L); memcpy(&order_s[i], os + 4 * (size_t)i, 4); }
    for (uint32_t i = 0; i < NN; i++) { memcpy(rd + (size_t)(S + i) * L, Ntxt + (size_t)i * (L + 1), L); order_s[S + i] = i; }
    int ds[2], de[2];                                                  /* encoder.cpp:132-145 */
    if (L > 50) { ds[0] = 0; de[0] = 20; ds[1] = 21; de[1] = 41; }
    else { ds[0] = 0; de[0] = 20 * L / 50; ds[1] = 20 * L / 50 + 1; de[1] = 41 * L / 50; }
    dict_t dict[2]; memset(dict, 0, sizeof dict);
    if (T) {
        uint64_t *keys = malloc(8 * (size_t)T);
        for (int l = 0; l < 2; l++) { for (uint32_t i = 0; i < T; i++) keys[i] = key3(rd + (size_t)i * L, ds[l], de[l]); dict_build(&dict[l], keys, T); }
        free(keys);
    }
    uint8_t *claimed = calloc((size_t)T + 1, 1);
    uint32_t per = 1 + ((M - 1) / E);                                  /* uint32 arithmetic as encoder.cpp:171 */
    uint32_t prev_end = 0;
    bytes order_all = { 0 }, orderN_all = { 0 };
    char name[64];
    for (uint32_t e = 0; e < E; e++) {
        uint32_t start = e == 0 ? 0 : prev_end; if (start > M) start = M;
        uint32_t end = start + per; if (end > M || end < start) end = M;
        prev_end = end;
        shard_out so; memset(&so, 0, sizeof so);
        clist c = { 0 };
        for (uint32_t i = start; i < end; i++) {
            if (flag[i] == '0' || c.n > 10000000) {                   /* encoder.cpp:226 */
                if (c.n) {
                    size_t reflen; char *ref = buildcontig(&c, L, &reflen);
                    if (T) realign(&c, ref, reflen, L, dict, ds, de, rd, order_s, claimed, 24, 1000);
                    writecontig(ref, &c, L, &so);
                    for (size_t k = 0; k < c.n; k++) if (c.v[k].owned) free((void *)c.v[k].read);
                    free(ref); c.n = 0;
                }
            }
            centry en; en.pos = pos[i]; en.read = (const char *)dna + (size_t)i * (L + 1); en.owned = 0; en.rc = (char)rc[i];
            memcpy(&en.order, ord + 4 * (size_t)i, 4);
            cl_push(&c, en);
        }
        if (start != end) {                                            /* last contig: no realignment (encoder.cpp:438-441) */
            size_t reflen; char *ref = buildcontig(&c, L, &reflen);
            writecontig(ref, &c, L, &so); free(ref);
        }
        free(c.v);
        bytes seqb = { 0 }, seqt = { 0 }, revb = { 0 }, revt = { 0 };
        packbits_seq(&so.seq, &seqb, &seqt); packbits_rev(&so.rc, &revb, &revt);
        snprintf(name, sizeof name, "read_seq.txt.%u", e); spit(basedir, name, seqb.v, seqb.n, "wb");
        snprintf(name, sizeof name, "read_seq.txt.%u.tail", e); spit(basedir, name, seqt.v, seqt.n, "wb");
        snprintf(name, sizeof name, "read_rev.txt.%u", e); spit(basedir, name, revb.v, revb.n, "wb");
        snprintf(name, sizeof name, "read_rev.txt.%u.tail", e); spit(basedir, name, revt.v, revt.n, "wb");
        snprintf(name, sizeof name, "read_pos.txt.%u", e); spit(basedir, name, so.pos.v, so.pos.n, "wb");
        snprintf(name, sizeof name, "read_noise.txt.%u", e); spit(basedir, name, so.noise.v, so.noise.n, "wb");
        snprintf(name, sizeof name, "read_noisepos.txt.%u", e); spit(basedir, name, so.noisepos.v, so.noisepos.n, "wb");
        bput(&order_all, so.order.v, so.order.n); bput(&orderN_all, so.order_N.v, so.order_N.n);
        free(so.seq.v); free(so.pos.v); free(so.noise.v); free(so.noisepos.v); free(so.order.v); free(so.order_N.v); free(so.rc.v);
        free(seqb.v); free(seqt.v); free(revb.v); free(revt.v);
    }
    /* tail section (encoder.cpp:457-503) */
    bytes singtxt = { 0 }, Nout = { 0 };
    uint32_t matched_s = S, matched_N = NN;
    for (uint32_t i = 0; i < S; i++) if (!claimed[i]) { matched_s--; bput(&order_all, &order_s[i], 4); bput(&singtxt, rd + (size_t)i * L, L); }
    for (uint32_t i = S; i < T; i++) if (!claimed[i]) { matched_N--; bput(&Nout, rd + (size_t)i * L, L); bputc(&Nout, '\n'); bput(&orderN_all, &order_s[i], 4); }
    spit(basedir, "read_order.bin", order_all.v, order_all.n, "wb");
    spit(basedir, "read_order_N_pe.bin", orderN_all.v, orderN_all.n, "wb");
    spit(basedir, "input_N.dna", Nout.v, Nout.n, "wb");
    snprintf(name, sizeof name, "%d\n", L); spit(basedir, "read_meta.txt", name, strlen(name), "wb");
    bytes sb = { 0 }, st = { 0 }; packbits_seq(&singtxt, &sb, &st);
    spit(basedir, "read_singleton.txt", sb.v, sb.n, "wb"); spit(basedir, "read_singleton.txt.tail", st.v, st.n, "wb");
    if (matched_s_out) *matched_s_out = matched_s;
    if (matched_N_out) *matched_N_out = matched_N;
    free(sb.v); free(st.v); free(singtxt.v); free(Nout.v); free(order_all.v); free(orderN_all.v);
    free(claimed); dict_free(&dict[0]); dict_free(&dict[1]); free(rd); free(order_s);
    free(dna); free(flag); free(pos); free(ord); free(rc); free(sing); free(os); free(Ntxt);
    return 0;
}

/* ------------------------------------------------------------------ pack_order.cpp:20-77 */
int harc_oracle_pack_order(const char *basedir)
{
    size_t n; uint8_t *in = slurp(basedir, "read_order.bin", &n);
    uint32_t numreads = (uint32_t)(n / 4);
    if (numreads == 0) { free(in); return -3; }                       /* reference: log2(0) UB, pack_order.cpp:36 */
    int numbits = (int)(log2((double)numreads) + 1);
    bytes out = { 0 }, tail = { 0 };
    bput(&out, &numbits, sizeof(int)); bput(&out, &numreads, 4);
    uint32_t *arr = malloc(4 * ((size_t)numbits + 2));
    for (uint32_t i = 0; i < numreads / 32; i++) {
        memset(arr, 0, 4 * ((size_t)numbits + 2));
        int pa = 0, pi = 0;
        for (int k = 0; k < 32; k++) {
            uint32_t order; memcpy(&order, in + 4 * ((size_t)i * 32 + k), 4);
            arr[pa] |= order << pi;
            if (pi + numbits > 32) { pa++; arr[pa] = order >> (32 - pi); pi = numbits - (32 - pi); }
            else if (pi + numbits == 32) { pa++; pi = 0; }
            else pi += numbits;
        }
        bput(&out, arr, 4 * (size_t)numbits);
    }
    bput(&tail, in + 4 * (size_t)(numreads / 32) * 32, 4 * (size_t)(numreads % 32));
    spit(basedir, "read_order.bin", out.v, out.n, "wb"); spit(basedir, "read_order.bin.tail", tail.v, tail.n, "wb");
    free(arr); free(out.v); free(tail.v); free(in);
    return 0;
}

/* ------------------------------------------------------------------ decoder.cpp:65-280 (non -p) -> output/output.dna */
static char dec_noise_of(char ref, char code)                        /* inverse of enc_noise (decoder.cpp setglobalarrays) */
{
    static const char *row_A = "CGTN", *row_C = "AGTN", *row_G = "TACN", *row_T = "GCAN", *row_N = "AGCT";
    const char *row = ref == 'A' ? row_A : ref == 'C' ? row_C : ref == 'G' ? row_G : ref == 'T' ? row_T : row_N;
    return row[code - '0'];
}
int harc_oracle_decoder(const char *basedir, uint32_t E)
{
    size_t n; uint8_t *meta = slurp(basedir, "read_meta.txt", &n);
    int L = atoi((const char *)meta); free(meta);
    if (L < 1 || L > MAXL) return -2;
    bytes out = { 0 }, outN = { 0 };
    char name[64];
    for (uint32_t e = 0; e < E; e++) {
        size_t nseq, nseqt, npos, nnoise, nnp, nrev, nrevt;
        snprintf(name, sizeof name, "read_seq.txt.%u", e); uint8_t *seqb = slurp(basedir, name, &nseq);
        snprintf(name, sizeof name, "read_seq.txt.%u.tail", e); uint8_t *seqt = slurp(basedir, name, &nseqt);
        snprintf(name, sizeof name, "read_pos.txt.%u", e); uint8_t *pos = slurp(basedir, name, &npos);
        snprintf(name, sizeof name, "read_noise.txt.%u", e); uint8_t *noise = slurp(basedir, name, &nnoise);
        snprintf(name, sizeof name, "read_noisepos.txt.%u", e); uint8_t *np = slurp(basedir, name, &nnp);
        snprintf(name, sizeof name, "read_rev.txt.%u", e); uint8_t *revb = slurp(basedir, name, &nrev);
        snprintf(name, sizeof name, "read_rev.txt.%u.tail", e); uint8_t *revt = slurp(basedir, name, &nrevt);
        size_t seqlen = 4 * nseq + nseqt; char *seq = malloc(seqlen + 1);
        for (size_t i = 0; i < nseq; i++) for (int k = 0; k < 4; k++) seq[4 * i + k] = "ACGT"[(seqb[i] >> (2 * k)) & 3];
        memcpy(seq + 4 * nseq, seqt, nseqt);
        size_t revlen = 8 * nrev + nrevt; char *rev = malloc(revlen + 1);
        for (size_t i = 0; i < nrev; i++) for (int k = 0; k < 8; k++) rev[8 * i + k] = (revb[i] >> k) & 1 ? 'r' : 'd';
        memcpy(rev + 8 * nrev, revt, nrevt);
        char ref[MAXL + 1], cur[MAXL + 2];
        memset(ref, 'A', sizeof ref);
        size_t sp = 0, nzp = 0, npp = 0;
        for (size_t i = 0; i < npos; i++) {
            int p = pos[i];
            if (p != 0) {
                for (int k = 0; k <= L - 1 - p; k++) ref[k] = ref[k + p];
                if (sp + (size_t)p > seqlen) { free(seq); free(rev); return -4; }
                memcpy(ref + L - p, seq + sp, (size_t)p); sp += (size_t)p;
            }
            memcpy(cur, ref, L);
            int prevnp = 0;
            while (nzp < nnoise && noise[nzp] != '\n') {
                int q = np[npp++] + prevnp;
                cur[q] = dec_noise_of(ref[q], (char)noise[nzp]); prevnp = q; nzp++;
            }
            nzp++;
            char rcbuf[MAXL + 2];
            const char *w = cur;
            if (rev[i] == 'r') { for (int k = 0; k < L; k++) rcbuf[k] = comp_of(cur[L - 1 - k]); w = rcbuf; }
            bytes *dst = memchr(cur, 'N', L) ? &outN : &out;
            bput(dst, w, L); bputc(dst, '\n');
        }
        free(seq); free(rev); free(seqb); free(seqt); free(pos); free(noise); free(np); free(revb); free(revt);
    }
    size_t ns, nst; uint8_t *sb = slurp(basedir, "read_singleton.txt", &ns), *st = slurp(basedir, "read_singleton.txt.tail", &nst);
    size_t slen = 4 * ns + nst; char *s = malloc(slen + 1);
    for (size_t i = 0; i < ns; i++) for (int k = 0; k < 4; k++) s[4 * i + k] = "ACGT"[(sb[i] >> (2 * k)) & 3];
    memcpy(s + 4 * ns, st, nst);
    for (size_t i = 0; i + L <= slen; i += L) { bput(&out, s + i, L); bputc(&out, '\n'); }
    bput(&out, outN.v, outN.n);
    size_t nN; uint8_t *Ntxt = slurp(basedir, "input_N.dna", &nN); bput(&out, Ntxt, nN);
    spit(basedir, "output.dna", out.v, out.n, "wb");
    free(s); free(sb); free(st); free(Ntxt); free(out.v); free(outN.v);
    return 0;
}

/* ------------------------------------------------------------------ preprocess.cpp:50-137 (N split) -- used by tests to stage inputs */
int harc_oracle_preprocess(const char *reads_txt, size_t len, int L, const char *basedir)
{
    bytes clean = { 0 }, withN = { 0 }, orderN = { 0 };
    uint32_t readnum = 0, nclean = 0;
    for (size_t off = 0; off + L <= len; off += L + 1, readnum++) {
        const char *r = reads_txt + off;
        if (memchr(r, 'N', L)) { bput(&withN, r, L); bputc(&withN, '\n'); bput(&orderN, &readnum, 4); }
        else { bput(&clean, r, L); bputc(&clean, '\n'); nclean++; }
    }
    spit(basedir, "input_clean.dna", clean.v, clean.n, "wb"); spit(basedir, "input_N.dna", withN.v, withN.n, "wb");
    spit(basedir, "read_order_N.bin", orderN.v, orderN.n, "wb"); spit(basedir, "numreads.bin", &nclean, 4, "wb");
    free(clean.v); free(withN.v); free(orderN.v);
    return 0;
}

/* ------------------------------------------------------------------ -q: ids and quality values (preprocess.cpp:61-118, reorder_quality.cpp)
 * fastq: the whole FASTQ text.  preserve_order != 0: output.id / output.quality are lines 1 and 4 of every record in file order
 * (preprocess.cpp:64-69: both "final" files are written directly).  preserve_order == 0: preprocess splits the lines into clean / N
 * files, then reorder_quality.out gathers them by the post-encoding orders:
 *     output.quality = quality_clean[read_order.bin[p]] for every p, then quality_N[read_order_N_pe.bin[i]] for every i
 *     output.id      = id_clean[read_order.bin[p]], then the N-id list AS IT IS (file order)
 * (reorder_quality.cpp:47-98 builds reverse_index and walks it bin by bin, which is this gather; :100-133 likewise for the N part).
 * The id side follows the reference to the letter, quirks included (pinned by tests/golden/q_*):
 *   - the id line is routed by flag_N BEFORE the record's own sequence line has updated it (preprocess.cpp:83-88 vs :98-110), i.e.
 *     by whether the PREVIOUS record had an N;
 *   - reorder_id_N writes its permuted ids to `infile_N` = input_N.quality, not to input_N.id (reorder_quality.cpp:208), so
 *     reorder_id appends input_N.id unpermuted (:181).
 * Returns -2 where the reference itself divides by zero (numreads/4 or numreads/8 == 0, reorder_quality.cpp:61,151). */
typedef struct { const char *p; size_t n; } span;
int harc_oracle_quality(const char *fastq, size_t len, int L, int preserve_order, const char *basedir)
{
    size_t nlines = 0, cap = 1024; span *ln = malloc(cap * sizeof(span));
    for (size_t s = 0; s < len;) {
        const char *e = memchr(fastq + s, '\n', len - s);
        size_t n = e ? (size_t)(e - (fastq + s)) : len - s;
        if (nlines == cap) { cap *= 2; ln = realloc(ln, cap * sizeof(span)); }
        ln[nlines].p = fastq + s; ln[nlines].n = n; nlines++;
        s += n + 1;
    }
    const size_t nrec = nlines / 4;
    bytes oq = { 0 }, oi = { 0 };
    int rc = 0;
    if (preserve_order) {
        for (size_t r = 0; r < nrec; r++) {
            bput(&oi, ln[4 * r].p, ln[4 * r].n); bputc(&oi, '\n');
            bput(&oq, ln[4 * r + 3].p, ln[4 * r + 3].n); bputc(&oq, '\n');
        }
        /* a dangling id line of a truncated last record is still written by the getline loop (case 0 runs before EOF is met) */
        if (nlines % 4 >= 1) { bput(&oi, ln[4 * nrec].p, ln[4 * nrec].n); bputc(&oi, '\n'); }
    } else {
        span *qc = malloc((nrec + 1) * sizeof(span)), *qn = malloc((nrec + 1) * sizeof(span)), *ic = malloc((nrec + 2) * sizeof(span)), *in = malloc((nrec + 2) * sizeof(span));
        size_t nqc = 0, nqn = 0, nic = 0, nin = 0; int flag_N = 0;
        for (size_t r = 0; r < nrec; r++) {
            if (!flag_N) ic[nic++] = ln[4 * r]; else in[nin++] = ln[4 * r];          /* preprocess.cpp:83-88: flag_N of the previous record */
            flag_N = memchr(ln[4 * r + 1].p, 'N', ln[4 * r + 1].n) != NULL;             /* :98-110 */
            if (ln[4 * r + 3].n != (size_t)L) rc = -3;                                 /* reorder_quality.cpp:78-79 assumes the (readlen+1) stride */
            if (!flag_N) qc[nqc++] = ln[4 * r + 3]; else qn[nqn++] = ln[4 * r + 3];     /* :112-117 */
        }
        if (nlines % 4 >= 1) { if (!flag_N) ic[nic++] = ln[4 * nrec]; else in[nin++] = ln[4 * nrec]; }   /* id line of a truncated last record */
        size_t no, nno; uint8_t *ob = slurp(basedir, "read_order.bin", &no), *nb = slurp(basedir, "read_order_N_pe.bin", &nno);
        const uint32_t *ord = (const uint32_t *)ob, *ordn = (const uint32_t *)nb;
        const size_t numreads = nqc, numreads_N = nqn;                               /* getDataParams, reorder_quality.cpp:221-237 */
        if (numreads / 8 == 0) rc = -2;
        if (no / 4 != numreads || nno / 4 != numreads_N) rc = rc ? rc : -4;
        if (rc == 0) {
            for (size_t p = 0; p < numreads; p++) { bput(&oq, qc[ord[p]].p, qc[ord[p]].n); bputc(&oq, '\n'); }
            for (size_t i = 0; i < numreads_N; i++) { bput(&oq, qn[ordn[i]].p, qn[ordn[i]].n); bputc(&oq, '\n'); }
            for (size_t p = 0; p < numreads; p++) { bput(&oi, ic[ord[p]].p, ic[ord[p]].n); bputc(&oi, '\n'); }      /* nic >= numreads always */
            for (size_t i = 0; i < nin; i++) { bput(&oi, in[i].p, in[i].n); bputc(&oi, '\n'); }                          /* f << f_N.rdbuf(), :181 */
        }
        free(ob); free(nb); free(qc); free(qn); free(ic); free(in);
    }
    if (rc == 0) { spit(basedir, "output.quality", oq.v, oq.n, "wb"); spit(basedir, "output.id", oi.v, oi.n, "wb"); }
    free(oq.v); free(oi.v); free(ln);
    return rc;
}
