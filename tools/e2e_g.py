# ./harc -c -g <ranks> -p end to end at a size beyond the tests', all ranks on the box's ONE GPU over the file transport (RCCL refuses two
# ranks on a device): a FASTQ of n reads -> archive -> ./harc -d -p -> the input file's reads in their order.  Both multi-GPU modes.
#   python tools/e2e_g.py [n_reads] [ranks] [mode[:VAR=value[,VAR=value]] ...]
import sys, os, time, subprocess, shutil, numpy as np, torch
sys.path.insert(0, '.')
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, G, err = 100, int(n * 100 / 26), 0.005
dev = torch.device("cuda", 0)
reads = bench.synth_reads(n, L, G, err, 1000, dev).cpu().numpy()
torch.cuda.empty_cache()
root = "/dev/shm/harc_e2e_g"; shutil.rmtree(root, ignore_errors=True); os.makedirs(root)
rec = np.empty((n, 2 * L + 16), dtype=np.uint8)
ids = np.char.zfill(np.arange(n).astype(str), 8)
rec[:, 0:3] = np.frombuffer(b"@T.", dtype=np.uint8); rec[:, 3:11] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n, 8); rec[:, 11] = 10
rec[:, 12:12 + L] = reads; rec[:, 12 + L] = 10; rec[:, 13 + L] = ord('+'); rec[:, 14 + L] = 10; rec[:, 15 + L:15 + 2 * L] = ord('H'); rec[:, 15 + 2 * L] = 10
want = np.empty((n, L + 1), dtype=np.uint8); want[:, :L] = reads; want[:, L] = 10
del reads
ok_all = True
for spec in (sys.argv[3:] or ["bucket", "replicate"]):
    mode, _, extra = spec.partition(":")
    d = os.path.join(root, mode); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    fq = os.path.join(d, "x.fastq"); rec.tofile(fq)
    env = dict(os.environ, HARC_AMD_MG_MODE=mode, HARC_AMD_XPORT="mailbox", HARC_AMD_SHARE_DEVICE="0", HARC_AMD_MAILBOX_TIMEOUT="600", HARC_AMD_STAGE3="none")
    env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
    t0 = time.time()
    r = subprocess.run(["./harc", "-c", fq, "-p", "-t", "4"] + ([] if mode == "single" else ["-g", str(ranks)]), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t1 = time.time()
    if r.returncode != 0:
        print(spec, "compress FAILED\n", r.stdout[-int(os.environ.get("E2E_TAIL", "2000")):]); ok_all = False; continue
    os.remove(fq)
    arc = os.path.join(d, "x.harc")
    r = subprocess.run(["./harc", "-d", arc, "-p"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t2 = time.time()
    if r.returncode != 0:
        print(spec, "decompress FAILED\n", r.stdout[-2000:]); ok_all = False; continue
    got = np.fromfile(os.path.join(d, "x.dna.d"), dtype=np.uint8)
    ok = got.size == want.size and bool((got.reshape(-1, L + 1) == want).all())
    ok_all &= ok
    gflag = "" if mode == "single" else f" -g {ranks}"
    print(f"{spec}: ./harc -c{gflag} -p {t1-t0:.1f}s, archive {os.path.getsize(arc)/1e6:.1f} MB (raw streams), ./harc -d -p {t2-t1:.1f}s, "
          f"{n} reads back in file order: {'ok' if ok else 'FAILED'}", flush=True)
    shutil.rmtree(d, ignore_errors=True)
shutil.rmtree(root, ignore_errors=True)
sys.exit(0 if ok_all else 1)
