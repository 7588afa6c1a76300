import torch
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

