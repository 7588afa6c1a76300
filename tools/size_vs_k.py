# stream sizes (xz -6 of every stage-II stream, the stage-III proxy) of the reference at -t 8 / -t 64 and of this build at several
# chain counts, on the bench workload:  python tools/size_vs_k.py [workload]   (GPU box; oracle/_ref must be prebuilt)
import sys, os, time, subprocess, shutil, numpy as np, torch
sys.path.insert(0, '.')
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, L, G, err, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
reads = bench.synth_reads(n, L, G, err, 1000, dev).cpu().numpy()
d = "/dev/shm/harc_svk"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
fq = os.path.join(d, "x.fastq")
rec = np.empty((n, 2 * L + 16), dtype=np.uint8)
ids = np.char.zfill(np.arange(n).astype(str), 8)
rec[:, 0:3] = np.frombuffer(b"@T.", dtype=np.uint8); rec[:, 3:11] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n, 8); rec[:, 11] = 10
rec[:, 12:12 + L] = reads; rec[:, 12 + L] = 10; rec[:, 13 + L] = ord('+'); rec[:, 14 + L] = 10; rec[:, 15 + L:15 + 2 * L] = ord('H'); rec[:, 15 + 2 * L] = 10
rec.tofile(fq)
del rec, reads


def size(od):
    names = sorted(f for f in os.listdir(od) if f.startswith(("read_seq", "read_pos", "read_noise", "read_noisepos", "read_rev", "read_singleton", "input_N", "read_meta")))
    tot = 0
    for f in names:
        tot += int(subprocess.check_output("xz -6 -T8 -c %s | wc -c" % os.path.join(od, f), shell=True))
    return tot


def fresh():
    shutil.rmtree(os.path.join(d, "output"), ignore_errors=True); os.makedirs(os.path.join(d, "output"))


ref = "oracle/_ref"
res = []
for t in (8, 64):
    if not os.path.exists(f"{ref}/reorder_L{L}_t{t}.out"):
        continue
    fresh()
    t0 = time.time()
    subprocess.check_call([f"{ref}/preprocess.out", fq, d, "False", "False", str(L)], stdout=subprocess.DEVNULL)
    log = subprocess.check_output([os.path.abspath(f"{ref}/reorder_L{L}_t{t}.out"), d], cwd=d, text=True)
    subprocess.check_call([os.path.abspath(f"{ref}/encoder_L{L}_t{t}.out"), d], cwd=d, stdout=subprocess.DEVNULL)
    dt = time.time() - t0
    um = [l for l in log.splitlines() if "unmatched" in l]
    res.append((f"reference -t {t}", size(os.path.join(d, "output")), dt, um[0] if um else ""))
    print(res[-1], flush=True)
for div in (4096, 2048, 1024, 512, 256):
    fresh()
    k = max(1, int(n * 0.88) // div)
    t0 = time.time()
    out = subprocess.check_output(["harc_amd/harc_amd_stage", "compressfq", d, str(L), fq, "8", str(k)], text=True)
    dt = time.time() - t0
    um = [l for l in out.splitlines() if "unmatched" in l]
    res.append((f"this build, K={k} (one chain per ~{div} reads)", size(os.path.join(d, "output")), dt, um[0] if um else ""))
    print(res[-1], flush=True)
base = res[0][1]
for r in res:
    print("%-48s %12d B  %+6.1f %% vs %s   %6.2f s   %s" % (r[0], r[1], 100.0 * (r[1] - base) / base, res[0][0], r[2], r[3]))
shutil.rmtree(d, ignore_errors=True)
