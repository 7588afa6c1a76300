# PCIe-inclusive rate: ASCII reads in (pageable / pinned) HOST memory -> all stage-II streams in host memory, workload c2
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import harc_amd, bench
n, L, G, err, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda", 0)
reads = bench.synth_reads(n, L, G, err, 1000, dev)
hasN = (reads == ord("N")).any(1)
clean = reads[~hasN].contiguous().cpu()
withN = reads[hasN].contiguous().cpu()
del reads
for pinned in (False, True):
    c_h = clean.pin_memory() if pinned else clean
    n_h = withN.pin_memory() if pinned else withN
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8))
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.set_reads_ascii(C.cast(c_h.numpy().ctypes.data, C.c_char_p), c_h.shape[0], L)
        h.set_nreads_ascii(C.cast(n_h.numpy().ctypes.data, C.c_char_p), n_h.shape[0], L)
        t1 = time.perf_counter()
        h.reorder(); h.encode()
        t2 = time.perf_counter()
    print(f"{'pinned' if pinned else 'pageable'} host ASCII: upload+pack {1e3*(t1-t0):.1f} ms, reorder+encode {1e3*(t2-t1):.1f} ms -> {n/(t2-t0)/1e6:.1f} Mreads/s PCIe-inclusive")
    h.close()
