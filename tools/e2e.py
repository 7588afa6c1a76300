# end-to-end: FASTQ file on local disk/tmpfs -> stage-II stream files, GPU ingest vs host preprocess
import sys, os, time, subprocess, shutil, numpy as np, torch
sys.path.insert(0, '.')
import bench
n, L, G, err, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda", 0)
reads = bench.synth_reads(n, L, G, err, 1000, dev).cpu().numpy()
d = "/dev/shm/harc_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
fq = os.path.join(d, "x.fastq")
t0 = time.time()
rec = np.empty((n, 2 * L + 16), dtype=np.uint8)          # "@T.nnnnnnnn\n" + read + "\n+\n" + quality + "\n"
ids = np.char.zfill(np.arange(n).astype(str), 8)
rec[:, 0:3] = np.frombuffer(b"@T.", dtype=np.uint8); rec[:, 3:11] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n, 8); rec[:, 11] = 10
rec[:, 12:12 + L] = reads; rec[:, 12 + L] = 10; rec[:, 13 + L] = ord('+'); rec[:, 14 + L] = 10; rec[:, 15 + L:15 + 2 * L] = ord('H'); rec[:, 15 + 2 * L] = 10
rec.tofile(fq); print("fastq bytes", os.path.getsize(fq), "written in %.1fs" % (time.time() - t0), flush=True)
stage = "harc_amd/harc_amd_stage"
for mode in ("compressfq", "host", "compressfq"):
    shutil.rmtree(os.path.join(d, "output"), ignore_errors=True); os.makedirs(os.path.join(d, "output"))
    t0 = time.time()
    if mode == "compressfq":
        subprocess.check_call([stage, "compressfq", d, str(L), fq, "8", "0"], stdout=subprocess.DEVNULL)
    else:
        subprocess.check_call([stage, "preprocess", d, str(L), fq], stdout=subprocess.DEVNULL); t1 = time.time()
        subprocess.check_call([stage, "compress", d, str(L), "8", "0"], stdout=subprocess.DEVNULL)
        print("   host preprocess %.2fs + compress (files) %.2fs" % (t1 - t0, time.time() - t1))
    dt = time.time() - t0
    print(f"{mode}: {dt:.2f}s wall -> {n/dt/1e6:.2f} Mreads/s end to end (process start, file read, GPU, stream files written)", flush=True)
shutil.rmtree(d, ignore_errors=True)
