# a FASTQ file larger than 4 GiB through the file-level entry points: compressfq (GPU ingest) -> decoder, order-independent check
import sys, os, time, subprocess, shutil, numpy as np, torch
sys.path.insert(0, '.')
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
L, G, err = 100, int(n * 100 / 26), 0.005
dev = torch.device("cuda", 0)
reads = bench.synth_reads(n, L, G, err, 1000, dev).cpu().numpy()
d = "/dev/shm/harc_e2e_big"; shutil.rmtree(d, ignore_errors=True); os.makedirs(os.path.join(d, "output"))
fq = os.path.join(d, "x.fastq")
rec = np.empty((n, 2 * L + 16), dtype=np.uint8)
ids = np.char.zfill(np.arange(n).astype(str), 8)
rec[:, 0:3] = np.frombuffer(b"@T.", dtype=np.uint8); rec[:, 3:11] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n, 8); rec[:, 11] = 10
rec[:, 12:12 + L] = reads; rec[:, 12 + L] = 10; rec[:, 13 + L] = ord('+'); rec[:, 14 + L] = 10; rec[:, 15 + L:15 + 2 * L] = ord('H'); rec[:, 15 + 2 * L] = 10
rec.tofile(fq); del rec
print("fastq bytes", os.path.getsize(fq), flush=True)
w = np.random.RandomState(1).randint(1, 1 << 62, size=L, dtype=np.int64).astype(np.uint64)
def sig(a):                                                       # order-independent: sum over lines of a position-weighted sum
    s = np.uint64(0)
    for i in range(0, a.shape[0], 1 << 20):
        with np.errstate(over="ignore"):
            s += (a[i:i + (1 << 20)].astype(np.uint64) * w[None, :]).sum(dtype=np.uint64)
    return int(s)
want = sig(reads); del reads
t0 = time.time()
subprocess.check_call(["harc_amd/harc_amd_stage", "compressfq", d, str(L), fq, "8", "0"], stdout=subprocess.DEVNULL)
t1 = time.time()
os.remove(fq)
subprocess.check_call(["harc_amd/harc_amd_stage", "decoder", d, "0", "8"], stdout=subprocess.DEVNULL)
t2 = time.time()
out = np.fromfile(os.path.join(d, "output", "output.dna"), dtype=np.uint8).reshape(-1, L + 1)
ok = out.shape[0] == n and bool((out[:, L] == 10).all()) and sig(out[:, :L]) == want
print(f"compressfq {t1-t0:.2f}s ({n/(t1-t0)/1e6:.1f} Mreads/s from the file), decoder {t2-t1:.2f}s, round trip {'ok' if ok else 'FAILED'} on {out.shape[0]} reads", flush=True)
shutil.rmtree(d, ignore_errors=True)
sys.exit(0 if ok else 1)
