# scratch: repeat-heavy genome (interspersed repeat copies + poly-A runs) -> big dictionary bins
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import harc_amd
dev='cuda'
g = torch.Generator(device=dev); g.manual_seed(5)
G, n, L = 6_300_000, 3_300_000, 100
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
genome = lut[torch.randint(0, 4, (G,), generator=g, device=dev)]
rep = lut[torch.randint(0, 4, (300,), generator=g, device=dev)]
ncopy = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pos = torch.randint(0, G - 400, (ncopy,), generator=g, device=dev)
for p in pos.tolist(): genome[p:p+300] = rep
for p in torch.randint(0, G - 400, (200,), generator=g, device=dev).tolist(): genome[p:p+150] = ord('A')
st = torch.randint(0, G - L, (n,), generator=g, device=dev)
reads = genome[st[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
p = harc_amd.default_params(L, num_thr=8)
h = harc_amd.HarcAmd(p)
torch.cuda.synchronize()          # the library works on its own stream: inputs must be complete
h.set_reads_ascii_device(reads.data_ptr(), n, L)
sig_in = h.reads_signature_device(reads.data_ptr(), n, L)
for it in range(2):
    t0 = time.time(); h.reorder(); t1 = time.time(); h.encode(); t2 = time.time()
    c = h.counters()
    print(f"copies={ncopy} iter {it}: reorder {t1-t0:.3f}s encode {t2-t1:.3f}s -> {n/(t2-t0)/1e6:.2f} Mreads/s rounds={c.rounds} cands={c.candidates} probes={c.probes} unmatched={c.unmatched} conflicts={c.conflicts} chain_ms={c.chain_ms:.0f} bigbins2={c.bins_over_maxsearch}", flush=True)
print("roundtrip", h.decode_signature() == sig_in)
