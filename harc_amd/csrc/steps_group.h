// steps_group.h -- k_steps_grp: the dense chain kernel with SEVERAL CHAINS PER WAVE (round 5).  Included by stage1.hip behind k_steps.
//
// k_steps gives a 64-lane wave to one chain.  At eight waves per SIMD that is eight chains in flight per SIMD -- all the hardware has -- and
// most of a wave's vector instructions run with most lanes idle: a batch of 64 probes finds its read at probe ~37 on average (configs[2]),
// the Hamming test of a candidate uses NW = 8 lanes, the window rows 16.  PMC (profiles/r04/pmc_c3_activity.txt): the vector pipe busy 73 % of
// the launch while every wave waits 65-68 % of its cycles -- issue and latency in balance, neither to be had by trimming.
//
// Here a wave holds 64 / G chains, G = 16 or 32 lanes each (a GROUP).  Each group is its own small state machine -- which step of its walk,
// which probe batch of the step -- and the wave's loop body is ONE SLOT: every live group makes one batch of G probes (reorder.cpp:517-649 in
// priority order, as k_steps' SPEC form: key window from the group's rows in LDS, scrambled key, minimizer-lined bitmap, table slot), tests its
// small-bin candidates one after the other with NW of its lanes (claim word + read in one trip, XOR + popcount against the mask row of
// (direction, shift), row_shr adds), and whichever group found its read finishes its step (steps[], the chain's own-reads table, rows from the
// read or updaterefcount of reorder.cpp:863-915) while the others simply go on to their next batch.  Nothing of a chain is wave-uniform any more:
// what k_steps keeps in scalar registers lives in vector registers (the same value in the G lanes of a group) or in LDS, ballots are cut to the
// group's lanes, readlane becomes ds_bpermute.  Same steps, same bids, same hints, same counts as k_steps for every chain: which kernel walks a
// super-round is not visible in the bytes (tests/test_gpu_parity.py::test_kernel_variants_same_bytes, HARC_AMD_GRP).
//
// What it buys: 2-4 times the chains in flight per wave, a batch that wastes G - x lanes behind the hit instead of 64 - x, and the per-step work
// (rows, counts, bookkeeping) done for 2-4 chains by one instruction stream.  What it leaves to k_steps: walks that reach a bin of more than
// HARC_LARGEBIN reads stop in front of it (CH_COOP, as before), runs whose parameters are not the SPEC ones (short reads, hashed bitmap lines).
//
// Column counts: 4 x u16 per ring slot in LDS (C16: 1 KB per chain of 100-150 bases instead of 2-3 KB -- LDS is what limits the chains per CU).  A
// chain whose counts came from HBM with a maximum above GRP_WIDE_LIMIT is not walked by the C16 kernel: it marks the chain CH_WIDE before it has
// changed anything of it, and the same kernel with u32 counts (k_steps_grp<W, 32, false>, launched behind it) walks exactly the marked chains and
// takes the mark off again -- every chain is walked once per super-round whatever its counts.  The host starts launching the second kernel when the
// first reports a chain above GRP_WIDE_WARN (a sticky word of the statistics, seen at the batch boundary -- a thousand super-rounds before any
// count can get from WARN to LIMIT: 16 per super-round at most), and launches it anyway during the first batch of a run of this kernel.
//
// WHERE IT STANDS (round 5, configs[2], one MI355X; profiles/r05/grp_*.txt): byte-identical to k_steps and to the oracle on every test, and NOT faster --
// 1001 us per launch against k_steps' 992 (G = 16, two probes per lane, 4 waves per SIMD of 126 registers), 18 % slower at 1 % errors (configs[3]:
// the column counts of a chain are 8 slots per lane here, 2 there).  It is therefore opt-in (HARC_AMD_GRP=1) and kept as a second, independent
// implementation of the walk that the parity tests hold to the same bytes.  What the counters say about why (tools: the HARC_GRP_STATS build):
//  * it does what it was built for: 295 M vector instructions per launch instead of 394 M, 188 M scalar instead of 325 M, 20 chains per SIMD instead of 8;
//  * but ONE slot of a wave is a chain of dependent waits that no lane of the wave escapes: keys (two LDS round trips, ~110 instructions) -> bitmap
//    word -> table slot -> claim word + read -> test -> step end (own-reads table, rows): 4.7 us per slot with ONE wave per SIMD (nothing to contend
//    with), 5.2 / 5.7 / 7.0 us at 2 / 3 / 5 waves, and in 72 % of the slots SOME group of the wave is in the expensive part, so every group pays for
//    it in its cheap slots too (a batch of 16 probes finds its read in one slot out of 3.4);
//  * five waves of 96 registers (LDS: 31 KB per workgroup of 16 chains) hide far less of that than eight waves of 64 do for k_steps: per SIMD
//    0.71 slots / us x 1.17 chain steps per slot = 0.84 steps / us against k_steps' 0.98; the register file is what a chain per 16 lanes costs (what k_steps
//    holds in 80 scalar registers is a vector register each here: the compiler wants 116-126);
//  * set-up and finish of a chain (counts in and out, flush of the lazy steps: 8 ring slots per lane) run at a quarter of the lanes: with every group for
//    itself (tickets per group) they were 19 % of a wave's life, with the groups of a wave in step (HARC_GRP_WSYNC) 13 %, at the price of 3.3 of 4 groups
//    walking on average;
//  * tried: G = 32 (1069-1178 us), 1 / 2 / 4 probes per lane (1 and 4 lose to 2: 1125 / 1034 / 1186 us), 5 waves with spills (+3 %), no look at the claim
//    bitmap ahead of the tests (-2 %: kept), a persistent grid with tickets (kept; the static grid had 84 % of its waves resident on average).
// What would make it win is a slot half as long, not more chains: the next thing to try here is a table whose slots are reached without the bitmap
// trip for the first probes of a step, or the set-up / finish moved to a kernel of their own so that a walk is nothing but slots.
#pragma once

#ifndef HARC_GRP_WAVES
#define HARC_GRP_WAVES 4            // waves per SIMD the register budget is cut for (launch bound): 128 registers and no spills; at 5 (96 registers, 9-17 of them
                                    // spilled inside the walk) the kernel is 3-7 % slower, see the measurements at the end of this header's comment
#endif
#ifndef HARC_GRP_WSYNC
#define HARC_GRP_WSYNC 1           // 1: the groups of a wave take their chains together and finish them together (set-up and finish run for all of them at once);
                                    // 0: every group for itself (set-up and finish of ONE chain then hold the other groups of the wave)
#endif
#define HARC_GRP_OWN 32             // slots of a chain's own-reads table in LDS (at most 16 reads of a super-round are in it)
#define GRP_WIDE_WARN 0x4000u
#define GRP_WIDE_LIMIT 0x8000u

template <int W, int G, bool C16> struct GrpGeom {
    static constexpr int NW = 2 * W, ROW = 3 * NW + 1, MROW = (NW + 3) & ~3;
    static constexpr int GPW = 64 / G, CPB = 4 * GPW;             // chains per wave / per workgroup of four waves
    static constexpr int LP = 64 * ((W + 1) / 2), CT = LP / G;    // ring slots of a chain's counts (the HBM layout of k_steps) / per lane
    static constexpr int QD = C16 ? 2 : 4;                        // dwords per ring slot in LDS
    static_assert(G == 32 || G == 16, "groups of 16 or 32 lanes");
    static_assert(2 * NW <= G, "the rows of a chain are squeezed by 2 NW lanes of its group");
    static_assert(NW <= 16, "the Hamming distance of a candidate is summed over one row of 16 lanes");
    static_assert(CT * 2 <= 32, "consensus bases of a lane's slots packed into one register");
};
static inline size_t steps_grp_lds_bytes(int W, int G, bool c16, int maxmatch, int nprobe)
{
    const int NW = 2 * W, ROW = 3 * NW + 1, MROW = (NW + 3) & ~3, CPB = 4 * (64 / G), LP = 64 * ((W + 1) / 2);
    return ((size_t)CPB * LP * (c16 ? 2 : 4) + (size_t)2 * maxmatch * MROW + (size_t)CPB * 2 * ROW + (size_t)CPB * MROW + (size_t)CPB * 8 * NW + (size_t)CPB * 8 + (size_t)CPB * HARC_GRP_OWN + (size_t)CPB * 8 + (size_t)2 * nprobe + 8) * 4 + 16;
}
// the lanes of the own group for which p holds, bit k = lane k of the group
template <int G> __device__ __forceinline__ uint32_t gballot(bool p, int g)
{
    const unsigned long long b = __ballot(p);
    if constexpr (G == 32) return g ? (uint32_t)(b >> 32) : (uint32_t)b;
    else return (uint32_t)(b >> (g * G)) & ((1u << G) - 1u);
}
__device__ __forceinline__ void grp_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void grp_own_insert(uint32_t *tab, uint32_t id)
{
    uint32_t h = (id * 0x9E3779B1u) >> 27;
    while (tab[h] != HARC_NONE) h = (h + 1u) & (HARC_GRP_OWN - 1u);
    tab[h] = id;
}
__device__ __forceinline__ bool grp_own_has(const uint32_t *tab, uint32_t id)
{
    uint32_t h = (id * 0x9E3779B1u) >> 27;
    for (;;) {
        const uint32_t v = tab[h];
        if (v == id) return true;
        if (v == HARC_NONE) return false;
        h = (h + 1u) & (HARC_GRP_OWN - 1u);
    }
}

// Consensus state of one chain, spread over the G lanes of its group: lane gl owns ring slots p = gl + G t (t < CT); the slot of consensus column i
// is (i + base) mod LP (ConsState of k_steps with a stride of G).  Counts in LDS (q: the lane's first slot), consensus bases two bits per slot in vpk.
template <int W, int G, bool C16> struct GCons {
    uint32_t *q;       // the chain's ring in LDS, at the lane's first slot (QD dwords per slot)
    uint32_t vpk;      // consensus base (count row A0 C1 G2 T3) of slot t at bits 2t
    int base;
    static constexpr int QD = C16 ? 2 : 4;
    __device__ __forceinline__ int v(int t) const { return (int)((vpk >> (2 * t)) & 3u); }
    __device__ __forceinline__ uint4 get(int t) const
    {
        if constexpr (C16) { const uint2 p = *reinterpret_cast<const uint2 *>(q + (size_t)G * t * QD); return make_uint4(p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16); }
        else return *reinterpret_cast<const uint4 *>(q + (size_t)G * t * QD);
    }
    __device__ __forceinline__ void set(int t, const uint4 &x)
    {
        if constexpr (C16) *reinterpret_cast<uint2 *>(q + (size_t)G * t * QD) = make_uint2(x.x | (x.y << 16), x.z | (x.w << 16));
        else *reinterpret_cast<uint4 *>(q + (size_t)G * t * QD) = x;
    }
};
#define GC_T template <int W, int G, bool C16>
#define GC_CT GrpGeom<W, G, C16>::CT
#define GC_LP GrpGeom<W, G, C16>::LP
// counts and consensus = the read whose dwords lie in rdl (reorder.cpp:875-883); `on`: this group takes part
GC_T __device__ __forceinline__ void gcons_reset(GCons<W, G, C16> &st, bool on, const uint32_t *rdl, int L, int gl)
{
    if (!on) return;
    st.base = 0; uint32_t vp = 0;
#pragma unroll
    for (int t = 0; t < GC_CT; t++) {
        const int i = gl + G * t;
        int b = 0; uint4 q = make_uint4(0, 0, 0, 0);
        if (i < L) {
            const int pc = (int)((rdl[i >> 4] >> (2 * (i & 15))) & 3u);
            b = ((pc & 1) << 1) | (pc >> 1);
            q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3);
        }
        st.set(t, q); vp |= (uint32_t)b << (2 * t);
    }
    st.vpk = vp;
}
// updaterefcount (reorder.cpp:884-909) for a match at `shift` with the read in rdl, oriented by `rev`
GC_T __device__ __forceinline__ void gcons_update(GCons<W, G, C16> &st, bool on, const uint32_t *rdl, int L, int rev, int shift, int gl)
{
    if (!on) return;
    constexpr int LP = GC_LP;
    int nb = st.base + shift; if (nb >= LP) nb -= LP;
    uint32_t vp = 0;
#pragma unroll
    for (int t = 0; t < GC_CT; t++) {
        const int p = gl + G * t;
        int oldl = p - st.base; if (oldl < 0) oldl += LP;
        int newl = p - nb; if (newl < 0) newl += LP;
        uint4 q = st.get(t); int v = 0;
        if (newl < L) {
            const int sc = rev ? (L - 1 - newl) : newl;
            const int pc = (int)((rdl[sc >> 4] >> (2 * (sc & 15))) & 3u);
            int b = ((pc & 1) << 1) | (pc >> 1);
            if (rev) b = 3 - b;
            if (oldl < L && oldl >= shift) { q.x += (b == 0); q.y += (b == 1); q.z += (b == 2); q.w += (b == 3); v = argmax4(q); }
            else { q.x = (b == 0); q.y = (b == 1); q.z = (b == 2); q.w = (b == 3); v = b; }
        } else q = make_uint4(0, 0, 0, 0);
        st.set(t, q); vp |= (uint32_t)v << (2 * t);
    }
    st.vpk = vp; st.base = nb;
}
// returns the largest count loaded (the lane's own columns)
GC_T __device__ __forceinline__ uint32_t gcons_load(GCons<W, G, C16> &st, bool on, const uint4 *src, int L, int gl, uint32_t fmt)
{
    if (!on) return 0u;
    st.base = 0; uint32_t vp = 0, mx = 0;
#pragma unroll
    for (int t = 0; t < GC_CT; t++) {
        const int i = gl + G * t;
        uint4 q = make_uint4(0, 0, 0, 0); int v = 0;
        if (i < L) {
            if (fmt == 2u) q = src[i];
            else if (fmt == 1u) { const uint2 p = reinterpret_cast<const uint2 *>(src)[i]; q = make_uint4(p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16); }
            else { const uint32_t p = reinterpret_cast<const uint32_t *>(src)[i]; q = make_uint4(p & 0xFFu, (p >> 8) & 0xFFu, (p >> 16) & 0xFFu, p >> 24); }
            v = argmax4(q);
            const uint32_t a = q.x > q.y ? q.x : q.y, b = q.z > q.w ? q.z : q.w, m = a > b ? a : b; mx = m > mx ? m : mx;
        }
        if (!C16 || fmt != 2u) st.set(t, q);                      // (a u32 form never goes into u16 slots: such a chain is not walked by this kernel)
        vp |= (uint32_t)v << (2 * t);
    }
    st.vpk = vp;
    return mx;
}
// the narrowest of the three forms of cons_store that holds every count of the chain; returns it (the same in all lanes of the group), *mxout = the
// lane's largest count.  Every lane of the wave must come here (the form is decided by a ballot); groups with !on write nothing
GC_T __device__ __forceinline__ uint32_t gcons_store(const GCons<W, G, C16> &st, bool on, uint4 *dst, int L, int gl, int g, uint32_t *mxout)
{
    constexpr int LP = GC_LP;
    uint32_t mx = 0;
    if (on) {
#pragma unroll
        for (int t = 0; t < GC_CT; t++) {
            int l = gl + G * t - st.base; if (l < 0) l += LP;
            if (l < L) { const uint4 q = st.get(t); const uint32_t a = q.x > q.y ? q.x : q.y, b = q.z > q.w ? q.z : q.w; const uint32_t m = a > b ? a : b; mx = m > mx ? m : mx; }
        }
    }
    *mxout = mx;
    const uint32_t fmt = gballot<G>(mx > 0xFFFFu, g) != 0 ? 2u : (gballot<G>(mx > 0xFFu, g) != 0 ? 1u : 0u);
    if (on) {
#pragma unroll
        for (int t = 0; t < GC_CT; t++) {
            int l = gl + G * t - st.base; if (l < 0) l += LP;
            if (l < L) {
                const uint4 q = st.get(t);
                if (fmt == 2u) dst[l] = q;
                else if (fmt == 1u) reinterpret_cast<uint2 *>(dst)[l] = make_uint2(q.x | (q.y << 16), q.z | (q.w << 16));
                else reinterpret_cast<uint32_t *>(dst)[l] = q.x | (q.y << 8) | (q.z << 16) | (q.w << 24);
            }
        }
    }
    return fmt;
}
// consensus -> the group's window rows (cons_rows of k_steps): column bytes into tmp, then 2 NW lanes squeeze 16 bytes into a dword each.
// Every lane of the wave comes here (wave barriers inside)
GC_T __device__ __forceinline__ void gcons_rows(const GCons<W, G, C16> &st, bool on, int L, int gl, uint8_t *tmp, uint32_t *rowF, uint32_t *rowR)
{
    constexpr int LP = GC_LP, NW = 2 * W;
    if (on) {
#pragma unroll
        for (int t = 0; t < GC_CT; t++) {
            int l = gl + G * t - st.base; if (l < 0) l += LP;
            if (l < L) {
                const int v = st.v(t);
                const int pc = ((v & 1) << 1) | (v >> 1);
                tmp[l] = (uint8_t)pc;
                tmp[16 * NW + (L - 1 - l)] = (uint8_t)(3 - pc);
            }
        }
    }
    grp_sync();
    if (on && gl < 2 * NW) {
        const uint4 x = *reinterpret_cast<const uint4 *>(tmp + 16 * gl);
        auto pk = [](uint32_t b) -> uint32_t { b = (b | (b >> 6)) & 0x000F000Fu; return (b | (b >> 12)) & 0xFFu; };
        const uint32_t d = pk(x.x) | (pk(x.y) << 8) | (pk(x.z) << 16) | (pk(x.w) << 24);
        (gl < NW ? rowF + NW + gl : rowR + gl)[0] = d;
    }
    grp_sync();
}
// rows_from_read of k_steps for a group: forward match -> rowF = the read, rowR = its reverse complement; reverse match the other way round
template <int W> __device__ __forceinline__ void grows_from_read(bool on, const uint32_t *rdl, int L, int rev, int gl, uint32_t *rowF, uint32_t *rowR)
{
    constexpr int NW = 2 * W;
    if (on && gl < 2 * NW) {
        const bool copy = gl < NW;
        const int k = copy ? gl : gl - NW;
        const int P = 2 * (L - 16 * k - 16), i0 = P >> 5;
        const int ia = i0 < 0 ? 0 : (i0 > NW - 1 ? NW - 1 : i0), ib = i0 + 1 < 0 ? 0 : (i0 + 1 > NW - 1 ? NW - 1 : i0 + 1);
        const uint32_t ra = rdl[copy ? k : ia], rb = rdl[ib];
        const uint32_t lo = (i0 >= 0 && i0 < NW) ? ra : 0u, hi = (i0 + 1 >= 0 && i0 + 1 < NW) ? rb : 0u;
        uint32_t w = __brev(__builtin_amdgcn_alignbit(hi, lo, P & 31));
        w = ((w & 0xAAAAAAAAu) >> 1) | ((w & 0x55555555u) << 1);
        const int vb = 2 * L - 32 * k;
        const uint32_t d = copy ? ra : (~w & (vb >= 32 ? 0xFFFFFFFFu : (vb <= 0 ? 0u : ((1u << vb) - 1u))));
        (copy != (rev != 0) ? rowF : rowR)[NW + k] = d;
    }
    grp_sync();
}
// cons_flush of k_steps for a group: `pend` steps that agreed with the consensus everywhere, cumulative shifts ps[0 .. pend), the last = ptot.
// Every lane of the wave comes here (the loop over the steps is the wave's)
GC_T __device__ __forceinline__ void gcons_flush(GCons<W, G, C16> &st, bool on, const uint16_t *ps, int pend, int ptot, const uint32_t *rowF, int L, int gl)
{
    constexpr int LP = GC_LP, CT = GC_CT, NW = 2 * W;
    int nb = 0;
    if (on && ptot < L) { nb = st.base + ptot; if (nb >= LP) nb -= LP; }
    int newl0 = gl - nb; if (newl0 < 0) newl0 += LP;              // slot t of the lane holds column (newl0 + G t) mod LP
    uint32_t cntp[(CT + 3) / 4];                                   // steps that cover slot t, one byte each (at most 64 steps)
#pragma unroll
    for (int k = 0; k < (CT + 3) / 4; k++) cntp[k] = 0;
    int k = on ? pend - 1 : -1;
    while (__ballot(k >= 0)) {
        if (k >= 0) {
            const int d = ptot - (int)ps[k];
            if (d >= L) k = -1;
            else {
#pragma unroll
                for (int t = 0; t < CT; t++) { int nl = newl0 + G * t; if (nl >= LP) nl -= LP; cntp[t >> 2] += (d < L - nl) ? (1u << (8 * (t & 3))) : 0u; }
                k--;
            }
        }
    }
    if (!on) return;
    uint32_t vp = 0;
#pragma unroll
    for (int t = 0; t < CT; t++) {
        int nl = newl0 + G * t; if (nl >= LP) nl -= LP;
        uint4 q = make_uint4(0, 0, 0, 0); int v = 0;
        if (nl < L) {
            const int pc = (int)((rowF[NW + (nl >> 4)] >> (2 * (nl & 15))) & 3u);
            v = ((pc & 1) << 1) | (pc >> 1);
            if (nl + ptot < L) q = st.get(t);
            const uint32_t c = (cntp[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            q.x += v == 0 ? c : 0u; q.y += v == 1 ? c : 0u; q.z += v == 2 ? c : 0u; q.w += v == 3 ? c : 0u;
        }
        st.set(t, q); vp |= (uint32_t)v << (2 * t);
    }
    st.vpk = vp; st.base = nb;
}

// C16 = true: the kernel that walks (nearly) everything; C16 = false: only the chains the other form leaves (counts above GRP_WIDE_LIMIT).
// U: probes per lane and batch (a batch is U G probes wide: G = 16, U = 2 looks at 32 probes per trip to the bitmap with four chains per wave)
template <int W, int G, bool C16, int U> __global__ __launch_bounds__(256, HARC_GRP_WAVES) void k_steps_grp(S1Args s)
{
    typedef GrpGeom<W, G, C16> GG;
    constexpr int NW = GG::NW, ROW = GG::ROW, MROW = GG::MROW, GPW = GG::GPW, CPB = GG::CPB, LP = GG::LP, QD = GG::QD;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *const s_mask = lds + (size_t)CPB * LP * QD;
    uint32_t *const s_rows = s_mask + (size_t)2 * s.maxmatch * MROW;
    uint32_t *const s_rdl = s_rows + CPB * 2 * ROW;
    uint32_t *const s_tmp = s_rdl + CPB * MROW;
    uint32_t *const s_pend = s_tmp + CPB * 8 * NW;
    uint32_t *const s_own = s_pend + CPB * 8;
    uint32_t *const s_hdr = s_own + CPB * HARC_GRP_OWN;            // the chain's header as the walk leaves it (8 dwords): the finish reads flags / nsteps / pad0 from here, not from memory
    uint2 *const s_pinfo = reinterpret_cast<uint2 *>(s_hdr + CPB * 8);
    const int lane = threadIdx.x & 63;
    const int g = lane / G, gl = lane % G, g0 = g * G;
    const int slot = (int)(threadIdx.x >> 6) * GPW + g;           // the chain's place in the workgroup
    const int L = s.L;
    if (!C16) {                                                   // this form walks the chains the C16 launch in front of it marked CH_WIDE: nearly every workgroup leaves here
        const uint32_t c0 = blockIdx.x * CPB + (uint32_t)slot;
        const bool mine = c0 < s.K && (s.hdr[c0].flags & (CH_ACTIVE | CH_WIDE)) == (CH_ACTIVE | CH_WIDE);
        if (!__syncthreads_or(mine ? 1 : 0)) return;
    }
    {   // mask rows and probe descriptors (k_steps_tables) -> LDS; rows and column bytes cleared
        const int nm = 2 * s.maxmatch * MROW;
        for (int i = threadIdx.x; i < nm; i += 256) s_mask[i] = s.lds_tab[i];
        const uint2 *const pt = reinterpret_cast<const uint2 *>(s.lds_tab + nm);
        for (int i = threadIdx.x; i < s.nprobe; i += 256) s_pinfo[i] = pt[i];
        for (int i = threadIdx.x; i < CPB * 2 * ROW; i += 256) s_rows[i] = 0u;
        for (int i = threadIdx.x; i < CPB * 8 * NW; i += 256) s_tmp[i] = 0u;
        __syncthreads();
    }
    uint32_t *const rowF = s_rows + (size_t)slot * 2 * ROW, *const rowR = rowF + ROW, *const rdl = s_rdl + (size_t)slot * MROW;
    uint32_t *const ownt = s_own + (size_t)slot * HARC_GRP_OWN;
    uint16_t *const pshift = reinterpret_cast<uint16_t *>(s_pend + (size_t)slot * 8);
    uint32_t *const hdrl = s_hdr + (size_t)slot * 8;
    const uint32_t *const reads32 = reinterpret_cast<const uint32_t *>(s.reads);
    const uint32_t *const claimed32 = reinterpret_cast<const uint32_t *>(s.claimed);
    GCons<W, G, C16> st;
    st.q = lds + ((size_t)slot * LP + gl) * QD; st.vpk = 0; st.base = 0;
    const uint32_t cap4 = (uint32_t)(s.cap[0] >> 2);              // buckets of four slots; fewer than 2^32 slots (the host checked)
    const bool lazy = s.lazy != 0;
    // A group walks one chain after the other: its first is the one of its place in the grid, the next ones come from a ticket counter (k_resolve
    // sets it back to zero for the next launch), so that the lanes of a group whose walk was short -- or whose wave-mates' walks are long -- do not
    // idle to the end of the launch: a wave lives as long as there are chains, not as long as its slowest chain.  Which group walks a chain is not
    // visible in anything a walk writes.  (The u32 form walks its place in the grid only.)
    // (design (R): ticket k is the k-th chain of THIS rank -- chains are dealt to the ranks four at a time)
    auto chain_of = [&](uint32_t k) -> uint32_t { return (!C16 || s.own_mod <= 1) ? k : 4u * (s.own_rem + s.own_mod * (k >> 2)) + (k & 3u); };
    uint32_t c = chain_of(blockIdx.x * CPB + (uint32_t)slot);
    uint32_t tkraw = 0;                                           // (lane 0 of the group) the ticket of the chain after this one: asked for one chain ahead, so that nobody waits for the counter
    if (C16 && gl == 0) tkraw = atomicAdd(s.grp_ticket, 1u);
    uint32_t ownmeta = 0;                                         // lane t of the group: shift | direction << 8 (or the look-ahead seed's marks) of step t
    bool have = false, live = false, more = true, first = true;   // have: a chain is set up in this group; live: its walk goes on; more: there may be chains left to take
    uint32_t ownreg = HARC_NONE;                                  // lane t of the group: the read this chain took at step t of this super-round
    uint32_t np = 0, ncs = 0, nuse = 0;                           // slots inspected (per lane); candidates tested, sequential-equivalent lookups (per group)
    int t = 0, base = 0;                                          // steps walked so far; probes [0, base) of the running step are done
    int spos = 0, nsugg = 0;
    bool rows_ok = false, needseed = false, defer = false;
    int pend = 0, ptot = 0;
#ifdef HARC_GRP_STATS
    unsigned long long gs_slots = 0, gs_live = 0, gs_take = 0, gs_fin = 0, gs_cand = 0, gs_hit = 0; const long long gs_t0 = wall_clock64();
    unsigned long long gs_ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; long long gs_tl = gs_t0;
#define GPH(k) do { const long long tn_ = wall_clock64(); gs_ph[k] += (unsigned long long)(tn_ - gs_tl); gs_tl = tn_; } while (0)
#else
#define GPH(k) do { } while (0)
#endif
    for (;;) {
        // ---- groups without a chain take the next one and set it up: the chain's header, its new seed if it asked for one, and the consensus at the
        //      start of the super-round -- a fresh seed (mode 2), the counts of the last super-round (0), or those + the kept steps (1).  The header
        //      goes back to memory here where the seed or the counts' form changed it and is read again at the end: eight registers less across the walk
        const bool anyhave = HARC_GRP_WSYNC && __ballot(have) != 0ULL;      // (asked outside the condition: a ballot behind && only sees the lanes that got that far)
        const bool take = !have && more && !anyhave;
        if (__ballot(take)) {
            if (take) {
                if (!first) {
                    if (C16) { const uint32_t tk = (uint32_t)__shfl((int)tkraw, g0, 64); c = chain_of(gridDim.x * CPB + tk); if (gl == 0 && c < s.K) tkraw = atomicAdd(s.grp_ticket, 1u); }
                    else c = s.K;
                }
                first = false;
                if (c >= s.K) more = false;
            }
            // design (R): chains are dealt to the ranks four at a time
            bool on = take && more && (s.own_mod <= 1 || ((c >> 2) % s.own_mod) == s.own_rem);
            ChainHdr h; h.cur = 0; h.prev = 0; h.flags = 0; h.mode = 0; h.n_main = 0; h.n_sing = 0; h.nsteps = 0; h.pad0 = 0;
            if (on) {
                const uint4 *hp = reinterpret_cast<const uint4 *>(&s.hdr[c]);
                const uint4 h0 = hp[0], h1 = hp[1];
                h.cur = h0.x; h.prev = h0.y; h.flags = h0.z; h.mode = h0.w; h.n_main = h1.x; h.n_sing = h1.y; h.nsteps = h1.z; h.pad0 = h1.w;
                if (!(h.flags & CH_ACTIVE)) on = false;
                if (!C16 && !(h.flags & CH_WIDE)) on = false;
            }
            const uint32_t par = (h.flags & CH_PARITY) ? 1u : 0u;
            uint4 *const B0 = s.cnt + ((size_t)par * s.K + (on ? c : 0u)) * LP;
            // whose chain is it?  The C16 form takes every chain that gets a fresh seed and every chain whose counts, as they come from HBM, stay below
            // GRP_WIDE_LIMIT; it marks the others CH_WIDE -- before it has changed anything of theirs -- and the u32 form behind it walks exactly those.
            const bool seednow = on && s.need[c] != 0;
            const bool fromhbm = on && !seednow && h.mode != 2;
            const uint32_t mxl = gcons_load(st, fromhbm, B0, L, gl, (h.pad0 >> (24 + 2 * par)) & 3u);
            bool hdirty = false;                                  // the header in memory is not the header any more
            if (C16) {
                const bool wide = gballot<G>(mxl > s.grp_wide_limit, g) != 0u;
                if (gballot<G>(mxl > s.grp_wide_warn, g) != 0u && gl == 0) s.stats[ST_WIDE] = 1ULL;      // sticky: the host starts launching the u32 form
                if (on && wide) { if (gl == 0) { atomicOr(&s.hdr[c].flags, CH_WIDE); atomicAdd(&s.stats[ST_WIDE + 1], 1ULL); } on = false; }      // (stats[ST_WIDE + 1]: chains marked and not yet walked; the host checks that it is back at 0)
            } else if (on) { h.flags &= ~CH_WIDE; hdirty = true; if (gl == 0) atomicAdd(&s.stats[ST_WIDE + 1], ~0ULL); }
            if (on && seednow) {
                // the chain asked for a seed last super-round and k_reseed ranked it: take seed number `rank` (reorder.cpp:650-688), or finish
                const uint32_t r = s.needrank[c], R = s.rmeta[0], assigned = s.rmeta[1], got = s.rmeta[2];
                if (h.flags & CH_PREVUNM) {                       // previous seed found nothing: singleton (reorder.cpp:672-684)
                    if (gl == 0) s.slog[h.prev] = make_uint2(c, h.n_sing);
                    h.n_sing++;
                }
                hdirty = true;
                if (r < assigned) {
                    const uint32_t id = s.seedbuf[r];
                    h.cur = id; h.prev = id; h.flags = ((h.flags | CH_PREVUNM) & ~CH_NEEDSEED) & 0xFFFFu; h.mode = 2;
                    const uint32_t first1 = r * (uint32_t)s.nsugg_per_seed;
                    const uint32_t ng = first1 >= got ? 0u : (got - first1 < (uint32_t)s.nsugg_per_seed ? got - first1 : (uint32_t)s.nsugg_per_seed);
                    if ((uint32_t)gl < ng) s.sugg[(size_t)c * s.nsugg_stride + gl] = s.seedbuf[R + first1 + gl];
                    h.nsteps = ng << 24;
                    if (gl == 0) { uint2 q = s.cst2[c]; q.x++; s.cst2[c] = q; s.need[c] = 0; }
                } else {                                          // no reads left (reorder.cpp:670-677)
                    h.flags &= ~(CH_ACTIVE | CH_PREVUNM | CH_NEEDSEED);
                    if (gl == 0) { atomicAdd(&s.stats[ST_ACTIVE], ~0ULL); s.need[c] = 0; s.hdr[c] = h; }
                    on = false; hdirty = false;
                }
            }
            {
                const bool fresh = on && h.mode == 2;
                if (__ballot(fresh)) {
                    if (fresh && gl < NW) rdl[gl] = reads32[(size_t)h.cur * NW + gl];
                    grp_sync();
                    gcons_reset(st, fresh, rdl, L, gl);
                    grp_sync();
                }
            }
            {
                const int nrep = (on && h.mode == 1) ? (int)((h.nsteps >> 8) & 0xFF) : 0;      // rolled back last time: replay the steps that were kept
                for (int r = 0; __ballot(r < nrep); r++) {
                    const bool onr = r < nrep;
                    uint2 sp = make_uint2(0, 0);
                    if (onr) sp = s.steps[(size_t)c * 64 + r];
                    grp_sync();
                    if (onr && gl < NW) rdl[gl] = reads32[(size_t)sp.x * NW + gl];
                    grp_sync();
                    const bool seedstep = ((sp.y >> 16) & 1u) != 0;
                    gcons_reset(st, onr && seedstep, rdl, L, gl);
                    gcons_update(st, onr && !seedstep, rdl, L, (int)((sp.y >> 8) & 1u), (int)(sp.y & 0xFFu), gl);
                }
                grp_sync();
            }
            {
                const bool onb = on && h.mode != 0;               // B0 now holds the rollback point of this super-round
                if (__ballot(onb)) {
                    uint32_t mx; const uint32_t w0 = gcons_store(st, onb, B0, L, gl, g, &mx);
                    if (onb && ((h.pad0 >> (24 + 2 * par)) & 3u) != w0) { h.pad0 = (h.pad0 & ~(3u << (24 + 2 * par))) | (w0 << (24 + 2 * par)); hdirty = true; }
                }
            }
            if (on && hdirty && gl == 0) s.hdr[c] = h;
            if (on && gl == 0) { hdrl[2] = h.flags; hdrl[6] = h.nsteps; hdrl[7] = h.pad0; }
            if (on) {
                if (gl < HARC_GRP_OWN) ownt[gl] = HARC_NONE;
                if (G < HARC_GRP_OWN && gl + G < HARC_GRP_OWN) ownt[gl + G] = HARC_NONE;
                have = true; live = true; ownreg = HARC_NONE; np = 0; ncs = 0; nuse = 0; t = 0; rows_ok = false; needseed = false; defer = false; pend = 0; ptot = 0;
                base = (int)(h.flags >> 16);                      // the first step was put off by the cooperative kernel: where it takes up again
                spos = (int)((h.nsteps >> 16) & 0xFF); nsugg = (int)(h.nsteps >> 24);
                if (t >= s.S) live = false;
            }
            grp_sync();
        }
        GPH(0);
        if (!__ballot(have || more)) break;
#ifdef HARC_GRP_STATS
        gs_slots++; gs_live += (unsigned long long)__popcll(__ballot(live && gl == 0)); if (__ballot(take)) gs_take++;
#endif
        // ---- consensus and its reverse complement -> the group's window rows, where they are not current
        {
            const bool on = live && !rows_ok;
            if (__ballot(on)) { gcons_rows(st, on, L, gl, reinterpret_cast<uint8_t *>(s_tmp + (size_t)slot * 8 * NW), rowF, rowR); if (on) rows_ok = true; }
        }
        GPH(1);
        // ---- one batch of U G probes in priority order, U per lane (probe base + u G + gl): the bitmap words of all of them in ONE trip to memory,
        //      then the table slots of those that passed, pair by pair, again all of a lane's probes at once
        bool cand[U], big[U];
        uint32_t c_sst[U], c_cw[U], c_sl[U];                      // the small bin a probe found: start / id, count word, slot
        {
            uint32_t hkx[U], hky[U], sl[U]; int st8[U];           // scrambled key, bucket, 0 = searching (pair (st8 >> 4) & 1 next), 1 = not in the table, 2 = found
            uint32_t bwv[U], bmv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                cand[u] = false; big[u] = false; c_sst[u] = 0; c_cw[u] = 0; c_sl[u] = 0; st8[u] = 1; hkx[u] = 0; hky[u] = 0; sl[u] = 0; bwv[u] = 0; bmv[u] = 0;
                if (live && base + u * G + gl < s.nprobe) {
                    const uint2 pi = s_pinfo[base + u * G + gl];
                    const int i0 = (int)((pi.x & 0x1FFF) >> 5), shb = (int)(pi.x & 31);
                    const uint32_t d0 = rowF[i0], d1 = rowF[i0 + 1], d2 = rowF[i0 + 2];
                    const uint64_t key = (uint64_t)__builtin_amdgcn_alignbit(d1, d0, shb) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, shb) << 32);
                    const uint64_t hk = key_scramble(key);
                    hkx[u] = (uint32_t)hk; hky[u] = (uint32_t)(hk >> 32);
                    sl[u] = __umulhi(hky[u], cap4) << 2;
                    uint32_t bw, bm;
                    bloom_pos(key, hk, s.bloom_lines, 17, 0xFFFFFFFFu, &bw, &bm);
                    bwv[u] = ((pi.x >> 14) & 1u ? s.bloom[1] : s.bloom[0])[bw]; bmv[u] = bm;
                    st8[u] = ((pi.x >> 14) & 1u) ? 8 : 0;         // bit 3: the dictionary
                }
            }
            GPH(2);
#pragma unroll
            for (int u = 0; u < U; u++) if (!(st8[u] & 3) && (bwv[u] & bmv[u]) != bmv[u]) st8[u] |= 1;      // most keys of a step are in neither table: they stop at the bitmap
            for (;;) {
                bool any = false;
                uint4 rawq[U][2];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!(st8[u] & 3)) {
                        any = true;
                        const HashSlot *const tab = (st8[u] & 8) ? s.slots[1] : s.slots[0];
                        const uint32_t at = sl[u] + ((st8[u] >> 4) & 1) * 2;
                        rawq[u][0] = *reinterpret_cast<const uint4 *>(&tab[at]); rawq[u][1] = *reinterpret_cast<const uint4 *>(&tab[at + 1]);
                    }
                }
                if (!any) break;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!(st8[u] & 3)) {
                        const int hp = (st8[u] >> 4) & 1;
                        if (hp == 0) st8[u] = (st8[u] & ~32) | ((rawq[u][0].w & SLOT_OVF) ? 32 : 0);      // bit 5: the bucket's overflow flag (slot 0)
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            if (!(st8[u] & 3)) {
                                np++;
                                if (rawq[u][q].w == 0) st8[u] |= 1;
                                else if (rawq[u][q].x == hkx[u] && rawq[u][q].y == hky[u]) { st8[u] |= 2; c_sl[u] = sl[u] + hp * 2 + q; c_sst[u] = rawq[u][q].z; c_cw[u] = rawq[u][q].w; }
                            }
                        }
                        if (!(st8[u] & 3)) {
                            if (hp == 0) st8[u] |= 16;                                    // the second pair of the bucket
                            else if (!(st8[u] & 32)) st8[u] |= 1;                          // a full bucket without the flag ends an unsuccessful search
                            else { st8[u] &= ~16; sl[u] += 4; if (sl[u] >= (uint32_t)s.cap[0]) sl[u] = 0; }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if ((st8[u] & 2) && !(c_cw[u] & SLOT_DEAD)) {
                    if (c_cw[u] & SLOT_BIG) big[u] = true;
                    else {
                        cand[u] = true;
                        if (c_cw[u] & SLOT_EMB) {
                            // single-read bins: claimed reads (when the batch found several such bins) and the chain's own reads are weeded out by all lanes at once
                            if (__popcll(__ballot(true)) >= s.weedmin && ((claimed32[c_sst[u] >> 5] >> (c_sst[u] & 31u)) & 1u)) { cand[u] = false; atomicOr(reinterpret_cast<uint32_t *>(&((st8[u] & 8) ? s.slots[1] : s.slots[0])[c_sl[u]]) + 3, SLOT_DEAD); }
                            else if (grp_own_has(ownt, c_sst[u])) cand[u] = false;
                        }
                    }
                }
            }
        }
        GPH(3);
        // ---- the small bins the batch found, one after the other in priority order, each candidate tested by NW lanes of the group
        unsigned long long gm = 0, bigm = 0;                      // bit u G + lane
#pragma unroll
        for (int u = 0; u < U; u++) { gm |= (unsigned long long)gballot<G>(cand[u], g) << (u * G); bigm |= (unsigned long long)gballot<G>(big[u], g) << (u * G); }
        if (bigm) gm &= (bigm & (0ULL - bigm)) - 1ULL;             // behind a large bin the walk stops anyway
        uint32_t found = HARC_NONE; uint32_t fmeta = 0;           // fmeta: probe index in the batch | shift << 8 | direction << 16 | (distance == 0) << 17
        while (__ballot(live && gm != 0ULL && found == HARC_NONE)) {
#ifdef HARC_GRP_STATS
            gs_cand++;
#endif
            const bool on = live && gm != 0ULL && found == HARC_NONE;
            const int w = on ? (__ffsll((long long)gm) - 1) : 0;
            if (on) gm &= gm - 1ULL;
            uint32_t x_sst = c_sst[0], x_cw = c_cw[0];
#pragma unroll
            for (int u = 1; u < U; u++) if (w / G == u) { x_sst = c_sst[u]; x_cw = c_cw[u]; }      // (w is the same in all lanes of the group)
            const uint32_t o_sst = (uint32_t)__shfl((int)x_sst, g0 + (w % G), 64), o_cw = (uint32_t)__shfl((int)x_cw, g0 + (w % G), 64);
            uint2 pi = make_uint2(0, 0);
            if (on) pi = s_pinfo[base + w];
            const int o_l = (int)((pi.x >> 14) & 1);
            const uint32_t cntb = o_cw & SLOT_CNT_MASK;
            const bool emb = (o_cw & SLOT_EMB) != 0;
            const int bitoff = (int)(pi.y & 0xFFFF), i0 = bitoff >> 5, sh = bitoff & 31;
            const uint32_t *const mrow = s_mask + (pi.y >> 16);
            uint32_t i = on ? cntb : 0u, lead = 0, hit = HARC_NONE, rd = 0, ntest = 0; bool alltop = true, hd0 = false;
            while (__ballot(i > 0u && hit == HARC_NONE)) {
                const bool on2 = i > 0u && hit == HARC_NONE;
                uint32_t rid = o_sst;
                if (on2 && !emb) rid = (o_l ? s.ids[1] : s.ids[0])[o_sst + i - 1u];
                if (on2) i--;
                uint32_t cwd = 0xFFFFFFFFu, rdn = 0;
                if (on2) { cwd = claimed32[rid >> 5]; rdn = gl < NW ? reads32[(size_t)rid * NW + gl] : 0u; }      // claim bit and read words together: one trip
                const bool clm = ((cwd >> (rid & 31u)) & 1u) != 0;
                if (on2 && clm && alltop) lead++;
                if (on2 && !clm) alltop = false;
                // taken by this chain earlier in this super-round? (not in the frozen bitmap)
                const bool own = gballot<G>(ownreg == rid && gl < t, g) != 0u;
                const bool test = on2 && !clm && !own;
                if (__ballot(test)) {
                    uint32_t hdp = 0;
                    if (test && gl < NW) hdp = (uint32_t)__popc((__builtin_amdgcn_alignbit(rowF[i0 + gl + 1], rowF[i0 + gl], sh) ^ rdn) & mrow[gl]);
                    hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x118, 0xF, 0xF, true);
                    hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x114, 0xF, 0xF, true);
                    hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x112, 0xF, 0xF, true);
                    hdp += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hdp, 0x111, 0xF, 0xF, true);
                    const int hd = __shfl((int)hdp, g0 + 15, 64);                 // the first row of the group holds the NW lanes
                    if (test) { ntest++; rd = rdn; if (hd <= s.thresh) { hit = rid; hd0 = hd == 0; } }
                }
            }
            // hints only (the claim bitmap stays the truth): claimed reads at the top of a bin are never looked at again
            // (ds_bpermute reads nothing from a lane that is switched off: the slot travels while the whole wave is here)
            if (__ballot(on && lead)) {
                uint32_t x_sl = c_sl[0];
#pragma unroll
                for (int u = 1; u < U; u++) if (w / G == u) x_sl = c_sl[u];
                const uint32_t o_slot = (uint32_t)__shfl((int)x_sl, g0 + (w % G), 64);
                if (on && lead && gl == 0) {
                    uint32_t *cp = reinterpret_cast<uint32_t *>(&(o_l ? s.slots[1] : s.slots[0])[o_slot]) + 3;
                    if (lead == cntb) atomicOr(cp, SLOT_DEAD); else if (!emb) atomicMin(cp, (cntb - lead) | (o_cw & SLOT_OVF));
                }
            }
            ncs += ntest;
            if (on && hit != HARC_NONE) {
                found = hit; fmeta = (uint32_t)w | ((pi.x >> 16) << 8) | (((pi.x >> 13) & 1u) << 16) | (hd0 ? (1u << 17) : 0u);
                if (gl < NW) rdl[gl] = rd;                         // the accepted read, for updaterefcount
            }
        }
        GPH(4);
        // ---- what the batch means for the walk
        const bool hitg = live && found != HARC_NONE;
        bool nohit = false;
        if (live && !hitg) {
            if (bigm) { defer = true; live = false; }             // a bin of more than HARC_LARGEBIN reads comes first: the cooperative kernel makes this step
            else { base += U * G; if (base >= s.nprobe) nohit = true; }
        }
        if (__ballot(hitg)) {
            const int fj = (int)((fmeta >> 8) & 0xFFu), fdir = (int)((fmeta >> 16) & 1u);
            if (hitg) {
                nuse += (uint32_t)base + (fmeta & 0xFFu) + 1u;
                if (gl == 0) grp_own_insert(ownt, found);
                if (gl == t) { ownreg = found; ownmeta = (uint32_t)fj | ((uint32_t)fdir << 8); }
            }
            grp_sync();
            const bool lz = hitg && lazy && ((fmeta >> 17) & 1u);      // the read agrees with the consensus on the whole overlap: the new consensus is the read
            if (__ballot(lz)) {
                grows_from_read<W>(lz, rdl, L, fdir, gl, rowF, rowR);
                if (lz) { ptot += fj; if (gl == 0) pshift[pend] = (uint16_t)ptot; pend++; rows_ok = true; }
            }
            const bool up = hitg && !lz;
            if (__ballot(up)) {
                const bool fl = up && pend > 0;
                if (__ballot(fl)) { grp_sync(); gcons_flush(st, fl, pshift, pend, ptot, rowF, L, gl); if (fl) { pend = 0; ptot = 0; } }
                gcons_update(st, up, rdl, L, fdir, fj, gl);
                if (up) rows_ok = false;
            }
            if (hitg) { t++; base = 0; }
        }
        if (__ballot(nohit)) {
            // no candidate: go on from the chain's look-ahead seeds (highest id first, skipping what was claimed meanwhile)
            uint32_t sid = HARC_NONE;
            if (nohit) {
                nuse += (uint32_t)s.nprobe;
                if (spos < nsugg) {
                    const int idx = spos + gl;
                    uint32_t id = 0; bool okc = false;
                    if (idx < nsugg) {
                        id = __hip_atomic_load(&s.sugg[(size_t)c * s.nsugg_stride + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        okc = !((claimed32[id >> 5] >> (id & 31u)) & 1u) && !grp_own_has(ownt, id);
                    }
                    const uint32_t sm = gballot<G>(okc, g);
                    if (sm) { const int f = __ffs((int)sm) - 1; sid = (uint32_t)__shfl((int)id, g0 + f, 64); spos += f + 1; } else spos = nsugg;
                }
            }
            const bool took = nohit && sid != HARC_NONE;
            if (nohit && !took) { needseed = true; live = false; }
            if (took) {
                if (gl == 0) grp_own_insert(ownt, sid);
                if (gl == t) { ownreg = sid; ownmeta = (1u << 16) | ((uint32_t)spos << 24); }
                if (gl < NW) rdl[gl] = reads32[(size_t)sid * NW + gl];
            }
            grp_sync();
            gcons_reset(st, took, rdl, L, gl);                    // every count is replaced: what was pending is gone with the old consensus
            if (took) { rows_ok = false; pend = 0; ptot = 0; t++; base = 0; }
            grp_sync();
        }
        GPH(5);
        if (live && t >= s.S) live = false;
        // ---- groups whose walk is over: the bids of its steps, all at once (lane t of the group holds the read of step t), the counts back to HBM,
        //      statistics and header
        const bool anylive = HARC_GRP_WSYNC && __ballot(live) != 0ULL;
        const bool fin = have && !live && !anylive;
#ifdef HARC_GRP_STATS
        if (__ballot(fin)) gs_fin++;
        if (__ballot(hitg)) gs_hit++;
#endif
        if (__ballot(fin)) {
            if (fin && gl < t) { atomicMin(&s.bid[ownreg], ((uint32_t)gl << 20) | c); s.steps[(size_t)c * 64 + gl] = make_uint2(ownreg, ownmeta); }
            {
                const bool fl = fin && pend > 0;
                if (__ballot(fl)) { grp_sync(); gcons_flush(st, fl, pshift, pend, ptot, rowF, L, gl); }
            }
            uint32_t hfl = 0, hns = 0, hp0 = 0;                                              // flags, nsteps, pad0 as the take left them
            if (fin) { hfl = hdrl[2]; hns = hdrl[6]; hp0 = hdrl[7]; }
            const uint32_t par1 = (hfl & CH_PARITY) ? 0u : 1u;                              // the half that holds the state at the END of the super-round
            {
                const bool onb = fin && t > 0;
                if (__ballot(onb)) {
                    grp_sync();
                    uint4 *const B1 = s.cnt + ((size_t)par1 * s.K + (onb ? c : 0u)) * LP;
                    uint32_t mx;
                    const uint32_t w1 = gcons_store(st, onb, B1, L, gl, g, &mx);
                    if (onb) hp0 = (hp0 & ~(3u << (24 + 2 * par1))) | (w1 << (24 + 2 * par1));
                }
            }
            // slots inspected: the sum over the group
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) np += (uint32_t)__shfl_xor((int)np, o, 64);
            if (fin && gl == 0) {
                uint32_t *const cs = reinterpret_cast<uint32_t *>(&s.cstat[c]);              // (adds without an answer: nobody waits for them)
                atomicAdd(cs, np); atomicAdd(cs + 1, ncs); atomicAdd(cs + 2, nuse); atomicAdd(cs + 3, ncs);
                const uint32_t resume = hfl >> 16;
                uint32_t fl = hfl;
                fl = needseed ? (fl | CH_NEEDSEED) : (fl & ~CH_NEEDSEED);
                fl = defer ? (fl | CH_COOP) : (fl & ~CH_COOP);
                fl = (fl & 0xFFFFu & ~CH_WIDE) | (((defer && t == 0) ? resume : 0u) << 16);
                if (defer) atomicAdd(&s.coopcnt[c & (HARC_COOPCNT - 1)], 1ULL);
                uint2 *hp = reinterpret_cast<uint2 *>(&s.hdr[c]);                            // cur prev | flags mode | n_main n_sing | nsteps pad0
                hp[1] = make_uint2(fl, 0u);
                hp[3] = make_uint2((hns & 0xFFFF0000u) | (uint32_t)t, (hp0 & 0xFF00FFFFu) | (((uint32_t)spos & 0xFFu) << 16));
            }
            if (fin) have = false;
            grp_sync();
        }
        GPH(6);
    }
#ifdef HARC_GRP_STATS
    if (lane == 0 && s.dbg) {
        atomicAdd(&s.dbg[0], 1ULL); atomicAdd(&s.dbg[1], gs_slots); atomicAdd(&s.dbg[2], gs_live); atomicAdd(&s.dbg[3], gs_take); atomicAdd(&s.dbg[4], gs_fin); atomicAdd(&s.dbg[5], gs_cand); atomicAdd(&s.dbg[6], gs_hit);
        atomicAdd(&s.dbg[7], (unsigned long long)(wall_clock64() - gs_t0)); atomicMax(&s.dbg[8], (unsigned long long)(wall_clock64() - gs_t0));
        for (int k = 0; k < 7; k++) atomicAdd(&s.dbg[16 + k], gs_ph[k]);
    }
#endif
}
#undef GPH
#undef GC_T
#undef GC_CT
#undef GC_LP
