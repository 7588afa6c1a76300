# scratch perf probe (not part of the product): python tools_perf_probe.py N L G err K E
import sys, time, torch, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import harc_amd

def synth(n, L, G, err, seed=1, dev='cuda'):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    genome = lut[torch.randint(0, 4, (G,), generator=g, device=dev)]
    out = torch.empty((n, L), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGTN", b"TGCAN"): comp[a] = b
    CH = 4_000_000
    for s in range(0, n, CH):
        m = min(CH, n - s)
        st = torch.randint(0, G - L, (m,), generator=g, device=dev)
        r = genome[st[:, None] + ar[None, :]]
        if err > 0:
            e = torch.rand((m, L), generator=g, device=dev) < err
            isN = e & (torch.rand((m, L), generator=g, device=dev) < 0.25)
            sub = e & ~isN
            code = torch.searchsorted(lut, r)
            nc = (code + torch.randint(1, 4, (m, L), generator=g, device=dev)) % 4
            r = torch.where(sub, lut[nc], r)
            r = torch.where(isN, torch.full_like(r, ord('N')), r)
        odd = (torch.arange(s, s + m, device=dev) % 2) == 1
        rc = comp[r.flip(1).long()]
        r = torch.where(odd[:, None], rc, r)
        out[s:s + m] = r
    return out

n, L, G, err, K, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
prof = int(sys.argv[7]) if len(sys.argv) > 7 else 0
S = int(sys.argv[8]) if len(sys.argv) > 8 else 0
reads = synth(n, L, G, err)
hasN = (reads == ord('N')).any(1)
clean = reads[~hasN].contiguous(); nn = reads[hasN].contiguous()
del reads
torch.cuda.synchronize()
print("clean", clean.shape[0], "N", nn.shape[0], flush=True)
p = harc_amd.default_params(L, num_thr=E, num_chains=K, profile=prof, num_steps=S)
h = harc_amd.HarcAmd(p)
h.set_reads_ascii_device(clean.data_ptr(), clean.shape[0], L)
h.set_nreads_ascii_device(nn.data_ptr(), nn.shape[0], L)
for it in range(2):
    t0 = time.time(); h.reorder(); t1 = time.time(); h.encode(); t2 = time.time()
    c = h.counters()
    print(f"iter {it}: reorder {t1-t0:.3f}s encode {t2-t1:.3f}s  -> {n/(t2-t0)/1e6:.2f} Mreads/s", flush=True)
    if it == 1: print({k: (round(v,2) if isinstance(v,float) else v) for k, v in c.as_dict().items() if k in ("chains","rounds","unmatched","conflicts","probes","candidates","useful_probes","index_ms","chain_ms","encode_ms")}, flush=True)
sizes = {k: sum(len(h.stream(k, e)) for e in range(E)) for k in ["S2_SEQ", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV"]}
sizes.update({k: len(h.stream(k)) for k in ["S2_ORDER", "S2_SINGLETON", "S2_INPUT_N"]})
print(sizes)
