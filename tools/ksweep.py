# scratch: K sweep (compression vs throughput) -- python tools_ksweep.py N L G err "K1,K2,..."
import sys, time, lzma, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import harc_amd
from perf_probe_lib import synth
n, L, G, err = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
Ks = [int(x) for x in sys.argv[5].split(',')]
E = 8
reads = synth(n, L, G, err)
hasN = (reads == ord('N')).any(1)
clean = reads[~hasN].contiguous(); nn = reads[hasN].contiguous()
for K in Ks:
    p = harc_amd.default_params(L, num_thr=E, num_chains=K)
    h = harc_amd.HarcAmd(p)
    h.set_reads_ascii_device(clean.data_ptr(), clean.shape[0], L)
    h.set_nreads_ascii_device(nn.data_ptr(), nn.shape[0], L)
    t0 = time.time(); h.reorder(); t1 = time.time(); h.encode(); t2 = time.time()
    c = h.counters()
    raw = {}; xz = {}
    for k in ["S2_SEQ", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV"]:
        b = b"".join(h.stream(k, e) for e in range(E)); raw[k] = len(b); xz[k] = len(lzma.compress(b, preset=6))
    for k in ["S2_SINGLETON", "S2_INPUT_N"]:
        b = h.stream(k); raw[k] = len(b); xz[k] = len(lzma.compress(b, preset=6))
    print(f"K={c.chains} rounds={c.rounds} reorder={t1-t0:.3f}s encode={t2-t1:.3f}s unmatched={c.unmatched} conflicts={c.conflicts} "
          f"sing_aligned={c.aligned_singletons} N_aligned={c.aligned_N} contigs={c.contigs} seq_bases={c.seq_bases} raw_total={sum(raw.values())} xz_total={sum(xz.values())} xz={xz}", flush=True)
    h.close()
