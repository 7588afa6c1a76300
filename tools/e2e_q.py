# -q without -p at a size beyond the tests': output.quality / output.id of the streamed path (the FASTQ ingested in 1 GiB pieces and streamed again
# once per bin of output, as for a file larger than HBM) against the resident path, byte for byte.   python tools/e2e_q.py [n_reads] [bin_bytes]
import sys, os, time, shutil, hashlib, numpy as np, torch
sys.path.insert(0, '.')
import bench, harc_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
binb = sys.argv[2] if len(sys.argv) > 2 else str(1 << 30)
L = 100
reads = bench.synth_reads(n, L, int(n * 100 / 26), 0.005, 7, torch.device("cuda", 0)).cpu().numpy()
rs = np.random.RandomState(3)
idw = rs.randint(0, 3, n)                                           # ids of 9, 10 or 11 characters
root = "/dev/shm/harc_e2e_q"; shutil.rmtree(root, ignore_errors=True); os.makedirs(root)
fq = os.path.join(root, "x.fastq")
with open(fq, "wb") as f:
    CH = 2_000_000
    for a in range(0, n, CH):
        b = min(n, a + CH); m = b - a
        q = (rs.randint(0, 40, (m, L)) + 35).astype(np.uint8)
        q[reads[a:b] == ord("N")] = ord("#")
        ids = np.char.zfill(np.arange(a, b).astype(str), 8)
        out = bytearray()
        recs = [b"@" + ids[i].encode() + b"x" * int(idw[a + i]) + b"\n" + reads[a + i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(m)]
        f.write(b"".join(recs))
print("FASTQ %.1f GB, %d reads" % (os.path.getsize(fq) / 1e9, n), flush=True)
res = {}
for mode, env in (("resident", {"HARC_AMD_Q_STREAM": "0"}), ("streamed", {"HARC_AMD_Q_STREAM": "1", "HARC_AMD_Q_BIN": binb, "HARC_AMD_TRACE": "1"})):
    d = os.path.join(root, mode); os.makedirs(os.path.join(d, "output"))
    for k in ("HARC_AMD_Q_STREAM", "HARC_AMD_Q_BIN", "HARC_AMD_TRACE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.time()
    harc_amd.compress_fastq(fq, d, L, num_thr=4, num_chains=0, preserve_quality=True)
    dt = time.time() - t0
    h = {}
    for name in ("output.quality", "output.id", "read_order.bin"):
        m = hashlib.md5()
        with open(os.path.join(d, "output", name), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                m.update(blk)
        h[name] = (m.hexdigest(), os.path.getsize(os.path.join(d, "output", name)))
    res[mode] = h
    print(mode, "%.1f s" % dt, h, flush=True)
    shutil.rmtree(d)
ok = res["resident"] == res["streamed"]
print("streamed == resident:", "ok" if ok else "FAILED")
shutil.rmtree(root, ignore_errors=True)
sys.exit(0 if ok else 1)
