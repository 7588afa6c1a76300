#!/usr/bin/env python3
"""bench.py -- Mreads/s of HARC reorder+encode (100 bp) on 1/2/4/8 MI355X, with the roofline of the dominant kernel and
the reference's CPU path timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|c2r|c2d|mini ...] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over the synthetic batch: index build + chaining (reorder.cpp:277-703) + encoding
(encoder.cpp:154-616), from 2-bit packed reads resident in HBM to every stage-II stream resident in host memory.
Default workload: c3 = BASELINE.json configs[2] (350 M x 100 bp error-free, 11.3x), the configuration the metric is quoted on.
N>1: every rank owns a batch of the same size drawn from a genome N times larger (weak scaling); inside every step the reads
are sharded by a canonical-minimizer bucket with ONE all-to-all over xGMI -- harc_amd_shard_exchange: RCCL send/recv inside
libharc_amd.so, packed read + u32 global id per read -- then each GPU chains and encodes its shard independently
(BASELINE.json north_star; DESIGN.md "Multi-GPU").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = None  # imported in main(), AFTER the launcher decision: the process that starts the ranks never touches the GPU


def _need_torch():
    """tools and tests import this module for its generator: torch comes in on first use"""
    global torch
    if torch is None:
        import torch as _t
        torch = _t


# ---------------------------------------------------------------------------------------------- launcher (python bench.py --gpus N)
def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as a fresh child
    (python -m torch.distributed.run --nproc-per-node N bench.py ...), relay rank 0's JSON line and leave with the child's exit
    code.  Nothing in this process has touched the GPU (torch is not even imported yet), and nothing is re-exec'ed.  The child
    runs in its own process group; when it outlives --launch-timeout the whole group is killed and the exit code is 124."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # RCCL between processes: dmabuf IPC (before any HIP call of a rank)
    env["HARC_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    line = [None]

    def descendants(pid):
        """the process ids below `pid` (torch.distributed.run puts its workers into sessions of their own, so the process group alone would
        miss them): psutil when it is there, else a walk over /proc"""
        try:
            import psutil
            try:
                return [pr.pid for pr in psutil.Process(pid).children(recursive=True)]
            except psutil.Error:
                return []
        except ImportError:
            kids = {}
            for d in os.listdir("/proc"):
                if d.isdigit():
                    try:
                        with open(f"/proc/{d}/stat") as f:
                            kids.setdefault(int(f.read().rsplit(")", 1)[1].split()[1]), []).append(int(d))
                    except (OSError, ValueError, IndexError):
                        pass
            out, todo = [], [pid]
            while todo:
                for k in kids.get(todo.pop(), []):
                    out.append(k); todo.append(k)
            return out

    def end_tree():
        """exactly the processes this call started: the launcher child and whatever descends from it.  A fresh kill, never a re-exec."""
        if child.poll() is not None:
            return
        tree = descendants(child.pid)
        child.terminate()                                         # torch.distributed.run ends its workers on SIGTERM
        try:
            child.wait(timeout=15)
        except subprocess.TimeoutExpired:
            pass
        for pid in tree:
            try:
                os.kill(pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        child.wait()

    class _Signalled(BaseException):
        pass

    def on_signal(signum, frame):
        # Ctrl-C, a harness timeout, SIGTERM: the ranks sit in sessions of their own and would run on as orphans on the GPUs.  The handler only
        # raises: the tree is ended by the try / finally below, OUTSIDE signal context (the main thread may be inside Popen.wait()'s lock when the
        # signal lands; terminating and waiting for the child from in here could deadlock on it)
        got[0] = signum
        raise _Signalled()
    got = [0]
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)

    def reader():
        for raw in child.stdout:
            txt = raw.decode("utf-8", "replace").strip()
            if txt.startswith("{") and '"metric"' in txt:
                line[0] = txt
            elif txt:
                print(txt, file=sys.stderr)
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    try:
        try:
            rc = child.wait(timeout=args.launch_timeout)
        except subprocess.TimeoutExpired:
            print(f"bench.py: the {args.gpus} ranks did not finish within {args.launch_timeout} s: ending process {child.pid} and its descendants", file=sys.stderr)
            end_tree()
            rc = 124
        except _Signalled:
            for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                signal.signal(sg, signal.SIG_IGN)                 # one tree kill is under way: a second signal changes nothing
            print(f"bench.py: signal {got[0]}: ending process {child.pid} and its descendants", file=sys.stderr)
            rc = 128 + got[0]
    finally:
        end_tree()                                                # every way out (a signal, an exception in this process): the same tree kill, once the wait is over
    if got[0]:
        sys.exit(rc)
    th.join(timeout=10)
    if rc == 0 and line[0] is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        rc = 1
    if line[0] is not None and rc == 0:
        print(line[0])
        sys.stdout.flush()
    sys.exit(rc)


class Watchdog:
    """A rank that waits for a peer inside a collective (ncclGroupEnd, a barrier, the communicator bootstrap) waits forever when the
    peer died or never came.  Every phase that can block on a peer runs under `with wd.phase(name, seconds)`: a daemon thread ends
    THIS process with exit code 86 when the phase overruns (the C calls release the GIL), torch.distributed.run then ends the other
    ranks and the launcher returns non-zero.  Never a re-exec."""

    def __init__(self):
        self.deadline = None
        self.name = ""
        self.lock = threading.Lock()
        t = threading.Thread(target=self.run, daemon=True)
        t.start()

    def run(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                d, nm = self.deadline, self.name
            if d is not None and time.monotonic() > d:
                sys.stderr.write(f"bench.py watchdog: rank {os.environ.get('RANK', '0')} stuck in '{nm}': leaving with exit code 86\n")
                sys.stderr.flush()
                os._exit(86)

    def phase(self, name, seconds):
        wd = self

        class P:
            def __enter__(self_):
                with wd.lock:
                    wd.deadline, wd.name = time.monotonic() + seconds, name

            def __exit__(self_, *a):
                with wd.lock:
                    wd.deadline = None
                return False
        return P()


def launch_only(args):
    """--launch-only (CPU, gloo): what the launcher and the watchdog do, without a GPU -- every rank joins a gloo group, the world size is
    all-reduced and rank 0 prints a result line; --hang-rank R makes rank R sleep instead of joining the barrier (the watchdog of the
    others must end the run with a non-zero exit code)."""
    import torch.distributed as dist
    wd = Watchdog()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    with wd.phase("init_process_group", args.watchdog):
        dist.init_process_group("gloo")
    import torch as th
    t = th.ones(1, dtype=th.int64)
    if args.hang_rank == rank:
        time.sleep(10 * args.watchdog + 60)
    with wd.phase("all_reduce", args.watchdog):
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch-only", "n_gpus": int(t.item()), "world_env": world, "launched": os.environ.get("HARC_BENCH_LAUNCHED") == "1"}))
        sys.stdout.flush()
    dist.destroy_process_group()

WORKLOADS = {
    # name: (reads per GPU, read length, genome bp per GPU, error rate, description == BASELINE.json config)
    "c2": (3_300_000, 100, 6_300_000, 0.005, "configs[1] stand-in: 3.3M x 100bp, 6.3 Mbp i.i.d. genome (~52x), 0.5% substitutions (1/4 N), odd reads RC"),
    "c2r": (3_300_000, 100, 6_300_000, 0.005, "c2 with a repeat-spiked genome: 2000 copies of one 300-bp element and 200 poly-A runs of 150 bp (hot dictionary bins)"),
    "c2d": (3_300_000, 100, 6_300_000, 0.005, "c2 with a diverged repeat family: 10000 copies of one 300-bp element, each with 12 % of its bases substituted, 300 poly-A runs and 300 (CA)n runs of 150 bp"),
    "c3": (350_000_000, 100, 3_100_000_000, 0.0, "configs[2] stand-in: 350M x 100bp error-free, 3.1 Gbp i.i.d. genome (11.3x), odd reads RC"),
    "c3s": (50_000_000, 100, 443_000_000, 0.0, "configs[2] at 1/7 scale: 50M x 100bp error-free, 443 Mbp i.i.d. genome (11.3x)"),
    "c1": (1_000_000, 100, 35_000_000, 0.0, "configs[0] stand-in: 1M x 100bp error-free, 35 Mbp i.i.d. genome (2.9x)"),
    "c4": (810_000_000, 101, 3_100_000_000, 0.01, "configs[3] stand-in on ONE GPU: 810M x 101bp, 3.1 Gbp i.i.d. genome (26x), 1% substitutions (1/4 N), odd reads RC"),
    "c4s": (50_000_000, 101, 191_000_000, 0.01, "configs[3] at 1/16 scale: 50M x 101bp, 191 Mbp i.i.d. genome (26x), 1% substitutions (1/4 N)"),
    "c5s": (250_000_000, 150, 194_000_000, 0.01, "configs[4] at 1/16 scale: 250M x 150bp, 194 Mbp i.i.d. genome (193x), 1% substitutions (1/4 N), -p (pack_order inside the step)"),
    "c5g": (500_000_000, 150, 388_000_000, 0.01, "configs[4]: ONE GPU's share of the 8-GPU run, 500M x 150bp, 388 Mbp i.i.d. genome (193x), 1% substitutions (1/4 N: 31 % of the reads carry an N), -p (pack_order inside the step)"),
    "c3r": (350_000_000, 100, 3_100_000_000, 0.0, "configs[2] with a human-like repeat content: 350M x 100bp error-free on a 3.1 Gbp genome of which 10 % is one diverged family of 300-mers (1 M copies, 12 %), 2 % a family of 6-kb elements (10 000 copies, 5 %), plus 30 000 poly-A and (CA)n runs and 2000 tandem arrays of a 171-mer"),
    "c4r": (810_000_000, 101, 3_100_000_000, 0.01, "configs[3] with a human-like repeat content (the shape of ERR194146): 810M x 101bp, 1 % errors, on a 3.1 Gbp genome of which 10 % is one diverged family of 300-mers (1 M copies, 12 %), 2 % a family of 6-kb elements (10 000 copies, 5 %), plus 30 000 poly-A and (CA)n runs and 2000 tandem arrays of a 171-mer"),
    "c3sd": (50_000_000, 100, 443_000_000, 0.005, "50M x 100bp on a 443 Mbp genome with a diverged repeat family: 100000 copies of one 300-bp element (12 % divergence), 3000 poly-A and 3000 (CA)n runs"),
    "c3m": (4_000_000, 100, 35_400_000, 0.0, "configs[2] at 1/88 scale: 4M x 100bp error-free, 35.4 Mbp i.i.d. genome (11.3x)"),
    "mini": (200_000, 100, 400_000, 0.005, "smoke-sized: 200k x 100bp, 0.4 Mbp genome"),
}


def synth_chunks(n, L, G, err, seed, dev, spike=None):
    """yields [m, L] uint8 ASCII reads, 4 M at a time: uniform starts on an i.i.d. genome, substitutions (a quarter become N, as
    gen_fastq_noRC.cpp:67-71), odd reads reverse-complemented (gen_fastq.cpp:105-113)."""
    _need_torch()
    g = torch.Generator(device=dev)
    g.manual_seed(4242 + G)                                      # the genome is the same on every rank
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    genome = torch.empty(G, dtype=torch.uint8, device=dev)
    for s in range(0, G, 1 << 28):
        m = min(1 << 28, G - s)
        genome[s:s + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
    if isinstance(spike, dict):
        # a human-like repeat content (what ERR194146 has and an i.i.d. genome lacks): interspersed families -- `fam`: (copies, element length,
        # per-copy divergence) each, every copy diverged from its family's element on its own and dropped into a slot of its own --, homopolymer
        # and dinucleotide runs, and tandem arrays of a short monomer (`sat`: (arrays, monomer length, monomers per array, per-monomer divergence))
        for ncopy, flen, div in spike.get("fam", []):
            rep = torch.randint(0, 4, (flen,), generator=g, device=dev, dtype=torch.uint8)
            slot_bp = flen + 100
            pos = torch.randperm(G // slot_bp, generator=g, device=dev)[:ncopy] * slot_bp
            CHK = max(1, (1 << 26) // flen)                        # copies per piece: the index tensor stays below a GiB
            ar_f = torch.arange(flen, device=dev)
            for a in range(0, int(pos.shape[0]), CHK):
                pp = pos[a:a + CHK]
                cp = rep[None, :].repeat(pp.shape[0], 1)
                if div > 0:
                    mut = torch.rand((pp.shape[0], flen), generator=g, device=dev) < div
                    cp = torch.where(mut, (cp + torch.randint(1, 4, (pp.shape[0], flen), generator=g, device=dev, dtype=torch.uint8)) % 4, cp)
                genome[(pp[:, None] + ar_f[None, :]).reshape(-1)] = lut[cp.reshape(-1).long()]
                del cp
        ar150 = torch.arange(150, device=dev)
        if spike.get("polya", 0):
            pa = torch.randint(0, G - 400, (spike["polya"],), generator=g, device=dev)
            genome[(pa[:, None] + ar150[None, :]).reshape(-1)] = ord("A")
        if spike.get("ca", 0):
            pc = torch.randint(0, (G - 400) // 2, (spike["ca"],), generator=g, device=dev) * 2      # even starts: two runs that overlap write the same bases (a scatter with clashing values is not reproducible)
            ca = torch.tensor(list(b"CA" * 75), dtype=torch.uint8, device=dev)
            genome[(pc[:, None] + ar150[None, :]).reshape(-1)] = ca.repeat(pc.shape[0])
        if spike.get("sat"):
            narr, mlen, nmono, sdiv = spike["sat"]
            mono = torch.randint(0, 4, (mlen,), generator=g, device=dev, dtype=torch.uint8)
            alen = mlen * nmono
            ps = torch.randperm(G // (alen + 100), generator=g, device=dev)[:narr] * (alen + 100)
            cp = mono[None, :].repeat(int(ps.shape[0]) * nmono, 1)
            mut = torch.rand(cp.shape, generator=g, device=dev) < sdiv
            cp = torch.where(mut, (cp + torch.randint(1, 4, cp.shape, generator=g, device=dev, dtype=torch.uint8)) % 4, cp)
            genome[(ps[:, None] + torch.arange(alen, device=dev)[None, :]).reshape(-1)] = lut[cp.reshape(-1).long()]
            del cp
    elif spike:                                                  # repeats and low-complexity runs: hot dictionary bins
        ncopy, div, npolya, nstr = spike
        rep = torch.randint(0, 4, (300,), generator=g, device=dev)
        pos = torch.randperm(G // 400, generator=g, device=dev)[:ncopy] * 400     # distinct 400-bp slots: copies never overlap
        cp = rep[None, :].repeat(ncopy, 1)
        if div > 0:                                              # every copy diverged from the element on its own
            mut = torch.rand((ncopy, 300), generator=g, device=dev) < div
            cp = torch.where(mut, (cp + torch.randint(1, 4, (ncopy, 300), generator=g, device=dev)) % 4, cp)
        genome[(pos[:, None] + torch.arange(300, device=dev)[None, :]).reshape(-1)] = lut[cp.reshape(-1)]
        for p in torch.randint(0, G - 400, (npolya,), generator=g, device=dev).tolist():
            genome[p:p + 150] = ord("A")
        ca = torch.tensor(list(b"CA" * 75), dtype=torch.uint8, device=dev)
        for p in torch.randint(0, G - 400, (nstr,), generator=g, device=dev).tolist():
            genome[p:p + 150] = ca
    g.manual_seed(seed)                                          # the reads differ per rank
    ar = torch.arange(L, device=dev)
    CH = 4_000_000
    for s in range(0, n, CH):
        m = min(CH, n - s)
        st = torch.randint(0, G - L, (m,), generator=g, device=dev)
        r = genome[st[:, None] + ar[None, :]]
        if err > 0:
            e = torch.rand((m, L), generator=g, device=dev) < err
            isN = e & (torch.rand((m, L), generator=g, device=dev) < 0.25)
            code = torch.searchsorted(lut, r)
            nc = (code + torch.randint(1, 4, (m, L), generator=g, device=dev)) % 4
            r = torch.where(e & ~isN, lut[nc], r)
            r = torch.where(isN, torch.full_like(r, ord("N")), r)
        odd = (torch.arange(s, s + m, device=dev) % 2) == 1
        r = torch.where(odd[:, None], comp[r.flip(1).long()], r)
        yield r
    del genome


def synth_reads(n, L, G, err, seed, dev, spike=None):
    _need_torch()
    return torch.cat(list(synth_chunks(n, L, G, err, seed, dev, spike)))


# workload -> (copies of a 300-bp element, per-copy divergence, poly-A runs, (CA)n runs) put into the genome
SPIKES = {"c2r": (2000, 0.0, 200, 0), "c2d": (10000, 0.12, 300, 300), "c3sd": (100000, 0.12, 3000, 3000),
          # the human-like repeat content of the BASELINE-sized repeat workloads: a tenth of the genome in one diverged family of short elements (an Alu-like
          # 300-mer, a million copies, 12 % apart from the element), 2 % in a family of long ones (6 kb x 10 000, 5 %), poly-A and (CA)n runs, and tandem
          # arrays of a 171-base monomer (2000 arrays x 18 monomers, 2 % apart: the centromeric kind of bin that holds tens of thousands of reads)
          "c4r": {"fam": [(1_000_000, 300, 0.12), (10_000, 6000, 0.05)], "polya": 30_000, "ca": 30_000, "sat": (2000, 171, 18, 0.02)},
          "c3r": {"fam": [(1_000_000, 300, 0.12), (10_000, 6000, 0.05)], "polya": 30_000, "ca": 30_000, "sat": (2000, 171, 18, 0.02)}}


def scale_spike(spike, f):
    """the repeat content of a genome f times the size (side legs on a sample: same fractions of the genome, same copy-number RATIOS would need the
    same genome; what is kept is the share of the genome every kind of repeat takes)"""
    if not isinstance(spike, dict):
        return spike
    out = {"fam": [(max(1, int(n * f)), fl, d) for n, fl, d in spike.get("fam", [])], "polya": max(1, int(spike.get("polya", 0) * f)), "ca": max(1, int(spike.get("ca", 0) * f))}
    if spike.get("sat"):
        a, m, k, d = spike["sat"]
        out["sat"] = (max(1, int(a * f)), m, k, d)
    return out


def install_synthetic(h, n, L, G, err, seed, dev, spike=None):
    """the synthetic workload -> the context h, 4 M reads at a time so that config-3/4-sized sets never exist as ASCII: clean reads are
    packed to 2 bits per base (k_pack2) as they are made, reads with N are kept as text for stage II.  Returns the order-independent
    signature [count, sum, xor of 64-bit read hashes] of all reads, taken on the way (for the round-trip check after a run)."""
    _need_torch()
    Wd = (2 * L + 63) // 64
    packed = torch.empty((n, Wd), dtype=torch.int64, device=dev)
    nclean, nparts = 0, []
    sig = [0, 0, 0]

    def acc(t):
        c3 = h.reads_signature_device(t.data_ptr(), t.shape[0], L)
        sig[0] += c3[0]; sig[1] = (sig[1] + c3[1]) % (1 << 64); sig[2] ^= c3[2]
    for r in synth_chunks(n, L, G, err, seed, dev, spike):
        hasN = (r == ord("N")).any(1)
        cl = r[~hasN].contiguous()
        wn = r[hasN].contiguous()
        torch.cuda.synchronize()                                  # libharc_amd works on its own stream: its inputs must be complete
        if cl.shape[0]:
            h.pack_reads_device(cl.data_ptr(), cl.shape[0], cl.stride(0), packed[nclean:].data_ptr())
            acc(cl)
            nclean += cl.shape[0]
        if wn.shape[0]:
            acc(wn)
            nparts.append(wn)
        del r, hasN, cl, wn
    withN = torch.cat(nparts) if nparts else torch.empty((0, L), dtype=torch.uint8, device=dev)
    del nparts
    packed = packed[:nclean]
    torch.cuda.synchronize()
    # this rank's slice of the job is installed ONCE (the library keeps its own copy): on one GPU it is the whole input, on N GPUs the
    # exchange inside every step starts from it
    h.set_reads_packed_device(packed.data_ptr(), nclean)
    h.set_nreads_ascii_device(withN.data_ptr(), withN.shape[0], L)
    del packed, withN
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                      # hand torch's cached blocks back: the library allocates with hipMalloc
    return sig


def stage_files(reads_np, d):
    """[n, L] uint8 reads -> the input files of the reference's reorder.out / encoder.out under <d>/output"""
    import numpy as np
    hasN = (reads_np == ord("N")).any(1)
    nl = np.full((reads_np.shape[0], 1), 10, dtype=np.uint8)
    lines = np.concatenate([reads_np, nl], axis=1)
    od = os.path.join(d, "output")
    os.makedirs(od)
    lines[~hasN].tofile(os.path.join(od, "input_clean.dna"))
    lines[hasN].tofile(os.path.join(od, "input_N.dna"))
    np.array([int((~hasN).sum())], dtype=np.uint32).tofile(os.path.join(od, "numreads.bin"))
    return od


def run_reference(reads_np, L, thr):
    """reorder.out + encoder.out of the REAL reference (oracle/_ref, built from /root/reference by oracle/build_ref.sh) at -t thr on
    these reads -> (wall seconds, {stage-II file: bytes})"""
    refdir = os.path.join(ROOT, "oracle", "_ref")
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmp, prefix="harc_cpu_") as d:
        od = stage_files(reads_np, d)
        t0 = time.time()
        subprocess.check_call([os.path.join(refdir, f"reorder_L{L}_t{thr}.out"), d], cwd=d, stdout=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(refdir, f"encoder_L{L}_t{thr}.out"), d], cwd=d, stdout=subprocess.DEVNULL)
        dt = time.time() - t0
        files = {}
        for f in os.listdir(od):
            if f.startswith(("read_seq", "read_pos", "read_noise", "read_rev.txt.", "read_singleton", "read_meta")) or f == "input_N.dna":
                files[f] = open(os.path.join(od, f), "rb").read()
    return dt, files


def cpu_baseline(reads_np, L, desc):
    """The reference itself timed on this box's host cores on a bounded sample of the workload; falls back to the C port
    (oracle/liboracle.so, one core) when the prebuilt reference is absent."""
    ncpu = os.cpu_count() or 1
    refdir = os.path.join(ROOT, "oracle", "_ref")
    thr = max([t for t in (1, 8, 16, 32, 64) if t <= ncpu and os.path.exists(os.path.join(refdir, f"reorder_L{L}_t{t}.out"))], default=0)
    n_sample = reads_np.shape[0]
    runs = []
    if thr:
        for _ in range(3):                                        # one sample moved 0.41 .. 0.52 Mreads/s between runs of the same leg: three, the median
            runs.append(run_reference(reads_np, L, thr)[0])
        dt = sorted(runs)[1]
        kind, cores = "reference", thr
    else:
        from tests import oracle_lib as ol
        o = ol.load()
        tmp = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=tmp, prefix="harc_cpu_") as d:
            stage_files(reads_np, d)
            t0 = time.time()
            assert o.harc_oracle_reorder(d.encode(), L, 1, 1, None, None) == 0
            assert o.harc_oracle_encoder(d.encode(), L, 1, None, None) == 0
            dt = time.time() - t0
        kind, cores = "port", 1
    return {"value": round(n_sample / dt / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": kind,
            "sample": f"{desc}, reorder+encode wall {dt:.1f}s" + (f" (median of {len(runs)} runs: " + ", ".join(f"{x:.1f}" for x in runs) + f" s), ./harc -t {thr} stage programs" if thr else ", scalar C port"),
            "runs_s": [round(x, 2) for x in runs]}


def xz_size(blobs):
    import lzma
    return sum(len(lzma.compress(b, preset=6)) for b in blobs)


def gpu_run_sample(harc_amd, reads_t, L, dev_index, shards, **kw):
    """one untimed reorder+encode of a sample on a fresh context -> (context, counters, seconds)"""
    hasN = (reads_t == ord("N")).any(1)
    cl, wn = reads_t[~hasN].contiguous(), reads_t[hasN].contiguous()
    p = harc_amd.default_params(L, num_thr=shards, device=dev_index, **kw)
    h = harc_amd.HarcAmd(p)
    torch.cuda.synchronize()
    h.set_reads_ascii_device(cl.data_ptr(), cl.shape[0], L)
    h.set_nreads_ascii_device(wn.data_ptr(), wn.shape[0], L)
    h.reorder(); h.encode()                                        # warm-up: pool growth, first-launch costs
    t0 = time.perf_counter()
    h.reorder(); h.encode()
    dt = time.perf_counter() - t0
    return h, h.counters(), dt


def stage2_blobs(h, shards):
    out = []
    for e in range(shards):
        for sid in ("S2_SEQ", "S2_SEQ_TAIL", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV", "S2_REV_TAIL"):
            out.append(h.stream(sid, e))
    for sid in ("S2_SINGLETON", "S2_SINGLETON_TAIL", "S2_INPUT_N", "S2_META"):
        out.append(h.stream(sid))
    return out


L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md "L2 (per XCD)": 8 x 4 MiB, ~34.5 TB/s aggregate


def chain_kernels_roofline(steps_alg, useful, cands_seq, dense_ms, dense_launches, coop):
    """The two kernels of the chain phase priced apart (a `frac` above 1 on the repeat workloads came from counters that covered both kernels
    over a time that covered one).  dense = k_steps' main form: random accesses into tables, bitmaps and reads -> HBM.  coop = k_steps<COOP>: walks that reach a
    dictionary bin of more than 16 reads; its up-to-maxsearch Hamming tests per probe (reorder.cpp:540-556) read bin-ordered MIRRORS of the reads that the
    scans of one super-round share out of L2 (16 concurrent walks per CU over a few thousand hot bins) -> priced against L2 bandwidth, not HBM.
    Algorithmic bytes per kernel, SURVEY.md 8(d)'s chain step: 49 B per read + 16 B per sequential dictionary lookup + 36 B per sequential candidate, each
    from the kernel's OWN counters; the reads are shared out by the steps each kernel walked.  coop = dict(ms, launches, useful, cands_seq, steps, dense_steps)."""
    walked = coop["steps"] + coop["dense_steps"]
    share_c = (coop["steps"] / walked) if walked else 0.0
    out = {}
    d_bytes = 49.0 * steps_alg * (1.0 - share_c) + 16.0 * (useful - coop["useful"]) + 36.0 * (cands_seq - coop["cands_seq"])
    c_bytes = 49.0 * steps_alg * share_c + 16.0 * coop["useful"] + 36.0 * coop["cands_seq"]
    for name, kern, nbytes, ms, launches, bound, peak, why in (
            ("dense", "k_steps (main form)", d_bytes, dense_ms, dense_launches, "hbm", 8000.0, "random accesses into tables, bitmaps and reads"),
            ("coop", "k_steps<COOP> (walks into bins of more than 16 reads)", c_bytes, coop["ms"], coop["launches"], "l2", L2_PEAK_GBS,
             "candidate scans stream the bin-ordered mirrors of the reads, shared by the walks of a super-round out of L2")):
        if not launches or ms <= 0:
            continue
        ach = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"kernel": kern, "bound": bound, "peak": peak, "unit": "GB/s", "achieved": round(ach, 2), "frac": round(ach / peak, 5), "ms_total": round(ms, 2),
                     "launches": int(launches), "avg_launch_us": round(ms / launches * 1e3, 2), "alg_bytes_per_launch": round(nbytes / launches, 1), "why_this_ceiling": why}
    if out:
        worst = min(out, key=lambda k: out[k]["frac"])
        longest = max(out, key=lambda k: out[k]["ms_total"])
        out["furthest_from_its_ceiling"] = f"{worst}: {out[worst]['frac']:.3f} of {out[worst]['bound'].upper()} peak"
        out["most_time"] = f"{longest}: {out[longest]['ms_total']:.0f} ms"
    return out


def end_to_end_leg(harc_amd, dev_index, dev, shards, n=100_000_000, L=100, cover=26.0, err=0.005):
    """FASTQ FILE -> raw stream FILES (SURVEY.md 8(d) "end-to-end, reported separately"; preprocess.cpp:81-121 + harc:50-69), outside the timed region:
    a synthetic FASTQ of n reads (ids @T.<i>, quality all 'H' as gen_fastq_noRC.cpp writes them) in /dev/shm -> harc_amd_compress_fastq_files_ex -> the stage-II
    files under <dir>/output, wall clock around the ONE library call with the library's own breakdown; then the decode side (harc:188: decoder.out's
    replacement) files -> output.dna, and the multiset signature of the decoded file against the reads written.  No stage III (bsc / 7z): the path ends where
    the reference's stage programs end."""
    import numpy as np
    import shutil
    root = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    rec_bytes = 2 * L + 17
    free = shutil.disk_usage(root).free
    n = int(min(n, free * 0.30 / rec_bytes))                      # input + outputs + output.dna must fit what /dev/shm has left
    d = tempfile.mkdtemp(dir=root, prefix="harc_e2e_")
    hs = harc_amd.HarcAmd(harc_amd.default_params(L, device=dev_index))          # only for the signature kernel
    try:
        fq = os.path.join(d, "x.fastq")
        G = max(4 * L, int(n * L / cover))
        t0 = time.perf_counter()
        sig = [0, 0, 0]
        at = 0
        pw = torch.tensor([10 ** (8 - k) for k in range(9)], dtype=torch.int64, device=dev)
        with open(fq, "wb") as f:
            for r in synth_chunks(n, L, G, err, 4321, dev, None):
                m = r.shape[0]
                torch.cuda.synchronize()
                c3 = hs.reads_signature_device(r.data_ptr(), m, L)
                sig[0] += c3[0]; sig[1] = (sig[1] + c3[1]) % (1 << 64); sig[2] ^= c3[2]
                rec = torch.empty((m, rec_bytes), dtype=torch.uint8, device=dev)
                rec[:, 0] = ord("@"); rec[:, 1] = ord("T"); rec[:, 2] = ord(".")
                idx = torch.arange(at, at + m, device=dev, dtype=torch.int64)
                rec[:, 3:12] = ((idx[:, None] // pw[None, :]) % 10 + 48).to(torch.uint8)
                rec[:, 12] = 10; rec[:, 13:13 + L] = r; rec[:, 13 + L] = 10; rec[:, 14 + L] = ord("+"); rec[:, 15 + L] = 10
                rec[:, 16 + L:16 + 2 * L] = ord("H"); rec[:, 16 + 2 * L] = 10
                f.write(memoryview(rec.cpu().numpy()).cast("B"))
                at += m
                del rec, idx, r
        torch.cuda.empty_cache()
        t_gen = time.perf_counter() - t0
        fsz = os.path.getsize(fq)
        os.makedirs(os.path.join(d, "output"))
        t0 = time.perf_counter()
        harc_amd.compress_fastq(fq, d, L, num_thr=shards, num_chains=0, device=dev_index)
        t_c = time.perf_counter() - t0
        lap = harc_amd.last_fastq_timing()
        os.remove(fq)
        out_bytes = sum(os.path.getsize(os.path.join(d, "output", x)) for x in os.listdir(os.path.join(d, "output")))
        t0 = time.perf_counter()
        harc_amd.decoder(d, shards, device=dev_index)
        t_d = time.perf_counter() - t0
        # the decoded file's reads against the reads written: multiset signature, the file through the GPU 256 MB at a time
        dn = os.path.join(d, "output", "output.dna")
        dsig = [0, 0, 0]
        rows = (256 << 20) // (L + 1)
        with open(dn, "rb") as f:
            while True:
                b = f.read(rows * (L + 1))
                if not b:
                    break
                t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev).view(-1, L + 1)
                torch.cuda.synchronize()
                c3 = hs.reads_signature_device(t.data_ptr(), t.shape[0], L + 1)
                dsig[0] += c3[0]; dsig[1] = (dsig[1] + c3[1]) % (1 << 64); dsig[2] ^= c3[2]
                del t
        other = max(0.0, lap["total"] - lap["context_and_pool"] - lap["ingest"] - lap["reorder"] - lap["encode"] - lap["stream_files"])
        return {"value": round(n / t_c / 1e6, 2), "unit": "Mreads/s, FASTQ file -> raw stage-II stream files (one call of harc_amd_compress_fastq_files_ex, wall clock)",
                "reads": n, "readlen": L, "fastq_bytes": fsz, "stream_file_bytes": out_bytes, "wall_s": round(t_c, 3),
                "breakdown_s": {"context_and_device_pool": round(lap["context_and_pool"], 3),
                                "ingest_file_to_packed_reads": round(lap["ingest"], 3),
                                "ingest_of_which_waiting_for_the_file_readers": round(lap["ingest_waiting_for_file_readers"], 3),
                                "ingest_of_which_device_passes": round(lap["ingest_device_passes"], 3),
                                "reorder": round(lap["reorder"], 3), "encode_with_d2h": round(lap["encode"], 3), "stream_files_written": round(lap["stream_files"], 3),
                                "small_files_and_rest": round(other, 3), "outside_the_library_call_to_its_own_clock": round(t_c - lap["total"], 3)},
                "file_read_GBps": round(fsz / max(1e-9, lap["ingest"]) / 1e9, 2),
                "decode": {"value": round(n / t_d / 1e6, 2), "unit": "Mreads/s, stream files -> output.dna (harc_amd_decoder_files)", "wall_s": round(t_d, 3)},
                "roundtrip": {"ok": bool(tuple(dsig) == tuple(sig)), "reads_decoded": int(dsig[0]), "reads_in": int(sig[0])},
                "input": f"{n} x {L} bp at {cover:g}x on an i.i.d. genome, {err:g} substitutions (a quarter N), ids @T.<i>, quality all H; in {root}",
                "fastq_generation_s": round(t_gen, 1)}
    finally:
        hs.close()
        shutil.rmtree(d, ignore_errors=True)


def run_other_config(harc_amd, name, dev_index, dev, shards, steps=2):
    """one of the other BASELINE configurations on this GPU: 1 warm-up (pool growth) + `steps` timed reorder + encode (+ pack_order) passes,
    timed like the main line (reads resident in HBM -> streams in pinned host memory), then the round trip on all reads"""
    n, L, G, err, desc = WORKLOADS[name]
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=shards, device=dev_index, profile=1))
    try:
        sig_in = install_synthetic(h, n, L, G, err, 1000, dev, SPIKES.get(name))
        po = name in ("c5s", "c5g")

        def step():
            h.reorder(); h.encode()
            if po:
                h.pack_order()
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = h.counters()
        dsig = h.decode_signature()
        return {"value": round(n * steps / dt / 1e6, 3), "unit": "Mreads/s", "steps": steps, "warmup": 1, "ms_per_step": round(dt / steps * 1e3, 2), "description": desc,
                "phases_ms_last_step": {"index": round(c.index_ms, 2), "chain": round(c.chain_ms, 2), "encode": round(c.encode_ms, 2)},
                "k_steps_avg_launch_us": round(c.propose_ms / max(1, c.propose_launches) * 1e3, 2), "rounds": int(c.rounds), "chains": int(c.chains),
                "chain_kernels": chain_kernels_roofline(c.n_clean, c.useful_probes, c.candidates_seq, c.propose_ms, c.propose_launches,
                                                        dict(ms=c.coop_ms, launches=c.coop_launches, useful=c.coop_useful_probes, cands_seq=c.coop_candidates_seq, steps=c.coop_steps, dense_steps=c.dense_steps)),
                "roundtrip": {"ok": bool(tuple(dsig) == tuple(sig_in)), "reads_decoded": int(dsig[0]), "reads_in": int(sig_in[0])},
                "device_bytes_peak": int(c.device_bytes_peak)}
    finally:
        h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("HARC_BENCH_WORKLOAD", "c3"))
    ap.add_argument("--chains", type=int, default=0)
    ap.add_argument("--super-steps", type=int, default=0, help="steps per super-round (num_steps; 0 = the library's choice: 16; 64 for one chain; 32 from 16 385 chains on, and from 2048 chains on inputs that are not low-coverage ones, where the index holds next to no large bins)")
    ap.add_argument("--shards", type=int, default=8, help="num_thr of the reference = encoder shards per GPU (harc:195 default 8)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs (reference baseline, size comparison) and the exact-mode sample")
    ap.add_argument("--no-prime", action="store_true", help="with --warmup 0: do NOT make the untimed pass that allocates the context's device pool and pinned buffers (the first timed step then pays for them: cold-start numbers)")
    ap.add_argument("--no-side-legs", action="store_true", help="N>1: skip rank 0's GPU side legs after the timed region (n1_equivalent, size_vs_1gpu) -- the peers wait for them in the final barrier")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the bucket exchange even at world size 1 (exercises the N>1 path on one GPU)")
    ap.add_argument("--cpu-sample", type=int, default=0)
    ap.add_argument("--no-other-configs", action="store_true", help="default workload on one GPU: skip the side lines for configs[2] with repeats (c3r, 1 timed step), configs[3] (c4) and configs[4]'s share (c5g), 2 timed steps each, after the main line's timed region")
    ap.add_argument("--end-to-end", type=int, default=0, help="run the end-to-end leg (FASTQ file in /dev/shm -> stream files -> output.dna) on this many reads whatever the workload (default: 100 M reads with the default workload)")
    ap.add_argument("--mg-mode", default="bucket", choices=["bucket", "replicate"],
                    help="N>1: bucket = minimizer-bucket shard + one all-to-all, independent shards (north_star; larger archives); replicate = design (R): "
                         "all-gather of the reads, chains partitioned over the GPUs, one all-gather of the walked steps per super-round -- every GPU "
                         "holds the whole job (pick a workload of which N batches fit one GPU, e.g. c3s) and the archive is byte-identical to one GPU's")
    ap.add_argument("--via-launcher", action="store_true", help="start the ranks through the launcher even for --gpus 1 (checks that the launcher costs nothing)")
    ap.add_argument("--launch-timeout", type=float, default=3300.0, help="launcher: seconds after which the ranks' process group is killed")
    ap.add_argument("--watchdog", type=float, default=600.0, help="seconds a rank may sit in one collective phase before it ends itself (exit code 86)")
    ap.add_argument("--launch-only", action="store_true", help="CPU rehearsal of launcher + watchdog over gloo (tests)")
    ap.add_argument("--hang-rank", type=int, default=-1, help="with --launch-only: this rank never joins the collective")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.via_launcher):
        launch_ranks(args, [a for a in sys.argv[1:] if a != "--via-launcher"])      # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start the ranks with --nproc-per-node {args.gpus} (or let bench.py do it: no WORLD_SIZE)")
    if args.launch_only:
        return launch_only(args)
    global torch
    import torch  # noqa: E402  (device memory, streams and torch.distributed only)
    wd = Watchdog()

    # stdout carries the JSON line and nothing else: whatever C libraries print (RCCL's version banner, the library's counters) goes to
    # stderr from here on
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL between processes needs dmabuf IPC on this driver (before the first HIP call)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < max(1, world if "LOCAL_RANK" in os.environ else 1):      # counting devices does not initialise the GPU
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPUs are visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libharc_amd has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with wd.phase("init_process_group", args.watchdog):
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=max(args.watchdog, 1800.0) + 60))    # nccl == RCCL on ROCm

    import harc_amd
    n, L, G, err, desc = WORKLOADS[args.workload]
    spike = SPIKES.get(args.workload)
    # weak scaling: same reads per GPU, genome (and so coverage) per GPU constant; every rank samples the WHOLE genome
    # a minimizer-bucket shard is already fragmented into islands of ~10 overlapping reads: twice as many chains cost +0.3 % (8 GPUs) to
    # +1.2 % (2 GPUs) of consensus bases there and save 38 % of the chain time (tools/shard_sim.py); one GPU keeps one chain per 2048 reads
    rpc = 1024 if (dist is not None and args.mg_mode == "bucket") else 0
    p = harc_amd.default_params(L, num_thr=args.shards, num_chains=args.chains, device=local, profile=1, num_steps=args.super_steps, reads_per_chain=rpc)
    h = harc_amd.HarcAmd(p)
    Wd = (2 * L + 63) // 64
    sig_in = install_synthetic(h, n, L, G * world, err, 1000 + rank, dev, spike)
    if dist is not None:
        from harc_amd import multigpu
        with wd.phase("communicator bootstrap", args.watchdog):
            multigpu.init_comm(h, dist, dev)                      # ncclUniqueId from the library, broadcast over the process group
            sig_in = list(multigpu.allreduce_signature(dist, tuple(sig_in), dev))    # the whole job's reads, BEFORE any exchange
    with_pack_order = args.workload in ("c5s", "c5g")

    replicate = dist is not None and args.mg_mode == "replicate"

    def step():
        if replicate:
            h.replicate_exchange()                                # design (R): all-gather of the reads; reorder() partitions the chains
        elif dist is not None:
            h.shard_exchange()                                    # bucket -> ONE all-to-all(v) over xGMI (RCCL inside the library) -> this GPU's shard
        h.reorder()
        h.encode()
        if with_pack_order:
            h.pack_order()                                        # -p: pack_order.cpp:20-77 on read_order.bin

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The first pass of a context allocates its device pool (tens of GB of hipMalloc at configs[2]: 0.1 ... 3 s on this box) and its pinned output
    # buffers; every later pass reuses them (tests/test_gpu_config_size.py::test_second_run_of_a_context_allocates_nothing_more).  With
    # --warmup 0 that allocation would sit inside the first timed step: one untimed priming pass is made then, and the line says so.
    priming = 1 if (args.warmup == 0 and not args.no_prime) else 0
    for _ in range(args.warmup + priming):
        with wd.phase("warm-up step", args.watchdog):
            step()
    agg = dict(propose_ms=0.0, launches=0, steps_alg=0, useful=0, cands=0, cands_seq=0, probes=0, rounds=0, coop_ms=0.0, coop_launches=0, coop_useful=0, coop_cseq=0, coop_steps=0, dense_steps=0)
    with wd.phase("barrier before the timed region", args.watchdog):
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        with wd.phase("timed step", args.watchdog):
            step()
        c = h.counters()
        agg["propose_ms"] += c.propose_ms
        agg["launches"] += c.propose_launches
        agg["cands"] += c.candidates
        agg["cands_seq"] += c.candidates_seq
        agg["probes"] += c.probes
        agg["useful"] += c.useful_probes
        agg["rounds"] += c.rounds
        agg["steps_alg"] += c.n_clean
        agg["coop_ms"] += c.coop_ms; agg["coop_launches"] += c.coop_launches; agg["coop_useful"] += c.coop_useful_probes
        agg["coop_cseq"] += c.coop_candidates_seq; agg["coop_steps"] += c.coop_steps; agg["dense_steps"] += c.dense_steps
    with wd.phase("barrier after the timed region", args.watchdog):
        barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        with wd.phase("max over ranks", args.watchdog):
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
    c = h.counters()
    # round trip at full size, outside the timed region: decode the streams of the last step on the GPU (decoder.cpp:90-169
    # restated in verify.hip) and compare the multiset signature (count, sum, xor of 64-bit read hashes) with the inputs'.  N GPUs:
    # the signatures of all ranks' INPUT slices (taken before the first exchange) against the signatures of all ranks' decoded
    # shards -- a read lost, duplicated or altered anywhere between the slice and the streams (the all-to-all included) shows.
    dsig = h.decode_signature()
    seq_bases_total, reads_total = int(c.seq_bases), n
    s2_env = os.environ.get("HARC_AMD_S2_PART", "1")
    s2_part = (world > 1 and s2_env != "0") or (replicate and s2_env == "2")      # design (R): stage II partitioned over the ranks by encoder shard (the default from two ranks on; 2: also at world 1, so that its collectives run)
    if replicate and not s2_part:
        reads_total = n * world                                   # every rank decoded the WHOLE job: its signature alone must equal all ranks' inputs
    elif dist is not None:
        with wd.phase("round-trip signature", args.watchdog):
            dsig = multigpu.allreduce_signature(dist, dsig, dev)
            sb = torch.tensor([int(c.seq_bases)], dtype=torch.int64, device=dev)
            dist.all_reduce(sb)
            seq_bases_total, reads_total = int(sb.item()), n * world
    roundtrip = {"ok": bool(tuple(dsig) == tuple(sig_in)), "reads_decoded": int(dsig[0]), "reads_in": int(sig_in[0]),
                 "check": "multiset signature (count, sum, xor of 64-bit read hashes) of the GPU-decoded streams of all ranks == the input reads of all ranks"}
    total_reads = n * world * args.steps
    value = total_reads / dt / 1e6

    # roofline of the dominant kernel k_steps (DESIGN.md "Kernels"): algorithmic bytes of SURVEY.md 8(d)'s chain step
    #   49 B/read (removals 40 + claim 2 + outputs 7) + 16 B per dictionary lookup of a strictly sequential scan (L-bar, counted by
    #   the kernel as `useful_probes`) + 36 B per candidate that scan would test (C-bar, `candidates_seq`).  The speculative lookups and
    #   candidates of the 64-lane batches are NOT counted; `candidates_per_read_speculative` shows them.
    kernels = chain_kernels_roofline(agg["steps_alg"], agg["useful"], agg["cands_seq"], agg["propose_ms"], agg["launches"],
                                     dict(ms=agg["coop_ms"], launches=agg["coop_launches"], useful=agg["coop_useful"], cands_seq=agg["coop_cseq"], steps=agg["coop_steps"], dense_steps=agg["dense_steps"]))
    # the top-level block is the MAIN kernel's on every workload, from its own counters over its own launches (no cooperative launch on inputs without large bins)
    walked = agg["coop_steps"] + agg["dense_steps"]
    share_d = (agg["dense_steps"] / walked) if walked else 1.0
    alg_bytes = 49.0 * agg["steps_alg"] * share_d + 16.0 * (agg["useful"] - agg["coop_useful"]) + 36.0 * (agg["cands_seq"] - agg["coop_cseq"])
    launches = max(1, agg["launches"])
    avg_ms = agg["propose_ms"] / launches
    achieved = (alg_bytes / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"kernel": "k_steps", "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": None,
                "avg_launch_us": round(avg_ms * 1e3, 2), "launches": launches,
                "alg_bytes_per_launch": round(alg_bytes / launches, 1),
                "Lbar_sequential_lookups_per_read": round(agg["useful"] / max(1, agg["steps_alg"]), 2),
                "Cbar_sequential_candidates_per_read": round(agg["cands_seq"] / max(1, agg["steps_alg"]), 3),
                "slots_inspected_per_read": round(agg["probes"] / max(1, agg["steps_alg"]), 2),
                "candidates_per_read_speculative": round(agg["cands"] / max(1, agg["steps_alg"]), 3),
                "Gslots_per_s": round(agg["probes"] / (agg["propose_ms"] * 1e-3) / 1e9, 2) if agg["propose_ms"] > 0 else None,
                "note": "random-access regime: 16-B slots and 32-B reads fetched as >=64-B sectors; see DESIGN.md",
                "kernels": kernels}
    try:                                                          # HBM bytes per launch from the committed PMC passes of the same command
        tr = json.load(open(os.path.join(ROOT, "profiles", "k_steps_traffic.json"))).get(args.workload)
        if tr and world == 1:
            # the counters were taken by a separate run: they describe THIS kernel only when that run loaded a library built from the same sources
            roofline["traffic_build_id"] = tr.get("build_id")
            roofline["traffic_stale"] = tr.get("build_id") != harc_amd.build_id()
            roofline["traffic"] = round((tr["fetch_kb_per_launch"] + tr["write_kb_per_launch"]) * 1024.0, 1)
            roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / max(1.0, roofline["alg_bytes_per_launch"]), 2)
            roofline["traffic_source"] = "NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this command taken by the builder, profiles/k_steps_traffic.json (" + tr.get("profile", "") + ")"
            # the random-access roofline: L2 misses of the PMC pass per launch over the launch time measured live, against the ceiling for requests
            # of the same shape -- one small load per random 64-byte sector, 64 GiB, eight waves per SIMD -- measured by two programs in one call
            # (tools/micro/gups sector, tools/micro/lat2 indep: 48.5 G/s, 64.0 B fetched per miss; profiles/r05/random_access_ceiling.txt)
            ceil = json.load(open(os.path.join(ROOT, "profiles", "random_access_ceiling.json")))
            if tr.get("tcc_miss") and avg_ms > 0:
                req = tr["tcc_miss"] / (avg_ms * 1e-3) / 1e9
                peak = ceil["sector"]["G_requests_per_s"]
                roofline["random_access"] = {"achieved": round(req, 2), "peak": peak, "unit": "G sector requests/s (L2 misses, 64 B fetched each)",
                                             "frac": round(req / peak, 3), "l2_misses_per_launch": tr["tcc_miss"], "l2_misses_source": "from the builder's separate PMC run (as traffic); the launch time is this run's",
                                             "request_shape": "one 4/8-byte load per random 64-byte sector (bitmap word, claim word, read dword, slot pair)",
                                             "peak_source": "tools/micro/gups sector 64 = 47.4-48.7, tools/micro/lat2 indep 64 = 45.4-49.4 G/s in one call (profiles/r05/random_access_ceiling.txt); rounds 2-4 quoted 26.3 = two dwordx4 loads per lane, a wider request than the kernel makes"}
            # ... and, once the bitmap in front of the tables keeps most probes away from them, instruction issue: vector instructions of
            # the PMC pass x 4 cycles (one wave64 instruction on a 16-lane SIMD) over the SIMD cycles of the launch measured live
            if tr.get("sq_insts_valu") and avg_ms > 0:
                simd_cycles = 256 * 4 * 2.4e9 * avg_ms * 1e-3
                roofline["issue"] = {"valu_insts_per_launch": tr["sq_insts_valu"], "salu_insts_per_launch": tr.get("sq_insts_salu"),
                                     "valu_busy_frac": round(tr["sq_insts_valu"] * 4.0 / simd_cycles, 3),
                                     "wave_cycles_waiting_frac": round(tr["sq_wait_any"] / max(1.0, tr["sq_wave_cycles"]), 3),
                                     "note": "instruction counts from the builder's separate PMC run (as traffic), the launch time is this run's; 1024 SIMDs at 2.4 GHz peak clock (the launch runs at ~2.1 GHz), 4 cycles per instruction (tools/micro/valu_rate: 2.9-5.3 measured): a lower bound of the busy fraction"}
    except Exception:
        pass
    out = {
        "metric": "Mreads/s reorder+encode, 100 bp", "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "warmup_effective": args.warmup + priming, "allocation_priming_passes": priming, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64 (2-bit packed bases, XOR+popcount)", "data": "synthetic",
        "config": {"workload": args.workload, "description": desc, "reads_per_gpu": n, "readlen": L, "genome_bp": G * world,
                   "error_rate": err, "chains_per_gpu": int(c.chains), "reads_per_chain": rpc or 2048, "encoder_shards_per_gpu": args.shards,
                   "schedule": "throughput mode: deterministic K-chain x S-step schedule of DESIGN.md (lossless, == CPU oracle byte for byte; "
                               "bytes differ from the reference's -t 1, which is num_chains = 1: see exact_mode)",
                   "parallelism": "single GPU" if world == 1 and dist is None else (f"design (R) x{world}: reads all-gathered, index replicated, chains partitioned, one all-gather of the walked steps per super-round; stage II partitioned by encoder shard with one all-reduce(min) of the claims; the ranks' stream files are the single-GPU archive" if replicate else f"minimizer-bucket shard x{world}, one RCCL all-to-all (8W+4 B per read) inside every step")},
        "roofline": roofline,
        "stage2_mode": ("partitioned by encoder shard over the ranks, one all-reduce(min) of the claims" if (replicate and s2_part) else "replicated on every rank" if replicate else "independent per shard" if dist is not None else "single GPU"),
        "build_id": harc_amd.build_id(),
        "roundtrip": roundtrip,
        "phases_ms_last_step": {"index": round(c.index_ms, 2), "chain": round(c.chain_ms, 2), "encode": round(c.encode_ms, 2)},
        "counters_last_step": {"unmatched": int(c.unmatched), "singletons_aligned": int(c.aligned_singletons), "N_aligned": int(c.aligned_N),
                               "rounds": int(c.rounds), "conflicts": int(c.conflicts), "contigs": int(c.contigs),
                               "seq_bases": int(c.seq_bases), "device_bytes_peak": int(c.device_bytes_peak)},
    }
    if dist is not None:
        # every N>1 line carries what ONE GPU does with its own batch, unsharded (no exchange, the single-GPU schedule): the N = 1 equivalent
        # measured in this very run on rank 0, outside the timed region
        out["per_gpu_value"] = round(value / world, 3)
        if rank == 0 and not args.no_side_legs:
            h.shard_reset()
            h.reorder(); h.encode()                               # warm-up at the unsharded size (pool growth)
            t1 = time.perf_counter()
            h.reorder(); h.encode()
            d1 = time.perf_counter() - t1
            c1 = h.counters()
            out["n1_equivalent"] = {"value": round(n / d1 / 1e6, 3), "unit": "Mreads/s", "what": "rank 0's own batch, unsharded, on one GPU (same reads per chain as in the multi-GPU run)",
                                    "ms_per_step": round(d1 * 1e3, 3), "seq_bases_per_read": round(int(c1.seq_bases) / max(1, n), 3)}
            out["scaling_vs_n1_equivalent"] = round(value / max(1e-9, out["n1_equivalent"]["value"]), 3)
    h.close()
    del h
    # ---- bounded side legs on rank 0 (outside the timed region): a sample of the same generator at the same coverage and error rate
    if rank == 0 and (not args.no_cpu or (dist is not None and not args.no_side_legs)):
        ns = args.cpu_sample or min(n, 3_300_000 if err > 0 or G // max(1, n) < 10 else 1_000_000)
        if isinstance(spike, dict):                               # the BASELINE-sized repeat workloads: a sample genome with the same SHARES of every kind of repeat
            ns = args.cpu_sample or 3_300_000
            Gs = max(L * 4, int(G * (ns / n))); sspike = scale_spike(spike, ns / n)
        elif spike:
            Gs, sspike = G, spike                                 # repeat-spiked genomes are not scaled: same genome, fewer reads would change the coverage
            ns = n
        else:
            Gs, sspike = max(L * 4, int(G * (ns / n))), None
        sample = synth_reads(ns, L, Gs, err, 999, dev, sspike)
        sdesc = f"{ns} reads of the same generator on a {Gs} bp genome (same coverage and error rate)"
        if (world > 1 or args.force_dist) and not replicate:
            # the price of bucket sharding in compressed size: consensus bases of the sample compressed as `world` minimizer
            # buckets, one after the other on this GPU with the sharded run's parameters, over the consensus bases of the same sample unsharded
            nb = max(2, world)
            h1, c1, _ = gpu_run_sample(harc_amd, sample, L, local, args.shards)
            base1 = int(c1.seq_bases); h1.close()
            hasN = (sample == ord("N")).any(1)
            cl = sample[~hasN].contiguous()
            pk = torch.empty((cl.shape[0], Wd), dtype=torch.int64, device=dev)
            bk = torch.empty((cl.shape[0],), dtype=torch.int32, device=dev)
            hb = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=args.shards, device=local, reads_per_chain=1024))
            torch.cuda.synchronize()
            hb.pack_reads_device(cl.data_ptr(), cl.shape[0], cl.stride(0), pk.data_ptr())
            hb.bucket_reads_device(pk.data_ptr(), cl.shape[0], nb, bk.data_ptr())
            sharded = 0
            for b in range(nb):
                sel = pk[bk == b].contiguous()
                torch.cuda.synchronize()
                hb.set_reads_packed_device(sel.data_ptr(), sel.shape[0])
                hb.set_nreads_ascii_device(0, 0, L)
                hb.reorder(); hb.encode()
                sharded += int(hb.counters().seq_bases)
            hb.close()
            out["size_vs_1gpu"] = {"value": round(sharded / max(1, base1), 3), "unit": "consensus bases, bucket-sharded / unsharded",
                                   "buckets": nb, "sample": sdesc + ", clean reads only",
                                   "seq_bases_per_read_this_run": round(seq_bases_total / max(1, reads_total), 3)}
            del pk, bk, cl
        if world == 1 and not args.no_cpu:
            sample_np = sample.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample_np, L, sdesc)
            # exact mode: num_chains = 1 is the reference at -t 1 byte for byte (tests/test_gpu_parity.py); its speed on a bounded sample
            ne = min(ns, 200_000)
            esample = synth_reads(ne, L, max(L * 4, int(Gs * (ne / ns))), err, 998, dev, None)
            he, ce, dte = gpu_run_sample(harc_amd, esample, L, local, 1, num_chains=1)
            out["exact_mode"] = {"value": round(ne / dte / 1e6, 4), "unit": "Mreads/s", "num_chains": 1, "num_thr": 1,
                                 "sample": f"{ne} reads, same coverage; streams byte-identical to the reference at -t 1"}
            he.close()
            # ... and the program it is identical to, on the same sample and this box's host: the reference's reorder + encoder at -t 1
            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"reorder_L{L}_t1.out")):
                dtr, _ = run_reference(esample.cpu().numpy(), L, 1)
                out["exact_mode"]["reference_t1"] = {"value": round(ne / dtr / 1e6, 4), "unit": "Mreads/s", "cores": 1, "sample": "the same reads"}
                out["exact_mode"]["vs_reference_t1"] = round(dtr / dte, 1)
            del esample
            # compressed size against the reference's default (-t 8), xz -6 of every stage-II stream as the stand-in for bsc / 7z
            refdir = os.path.join(ROOT, "oracle", "_ref")
            nz = min(ns, 1_000_000)
            if os.path.exists(os.path.join(refdir, f"reorder_L{L}_t8.out")):
                if isinstance(spike, dict): zs = synth_reads(nz, L, max(L * 4, int(G * (nz / n))), err, 997, dev, scale_spike(spike, nz / n))
                else: zs = sample[:nz].contiguous() if sspike else synth_reads(nz, L, max(L * 4, int(Gs * (nz / ns))), err, 997, dev, None)
                # the sample is run with the workload's reads per chain (a 1 M-read sample at the automatic chain count would be cut into
                # shorter chains than the 350 M-read workload is): what the workload's schedule costs in compressed size
                rpc_eff = max(1.0, float(c.n_clean) / max(1, int(c.chains)))
                zclean = int((~(zs == ord("N")).any(1)).sum())
                hz, cz, _ = gpu_run_sample(harc_amd, zs, L, local, 8, num_chains=max(1, int(round(zclean / rpc_eff))))
                ours = xz_size(stage2_blobs(hz, 8)); hz.close()
                _, rf = run_reference(zs.cpu().numpy(), L, 8)
                theirs = xz_size(list(rf.values()))
                out["size_vs_reference_t8"] = {"value": round(ours / max(1, theirs), 4), "unit": "xz -6 bytes of all stage-II streams, this build (default schedule) / reference -t 8",
                                               "ours_bytes": ours, "reference_bytes": theirs, "sample": f"{nz} reads, same coverage, one chain per {rpc_eff:.0f} reads as in the timed workload"}
        del sample
    # ---- the other BASELINE configurations that fit one GPU, on the same line (outside the timed region, after the context of the main
    #      workload is gone): configs[3] on ONE GPU (c4: 810 M x 101 bp, 1 % errors) and one GPU's share of configs[4] (c5g: 500 M x 150 bp, 31 %
    #      of the reads with N, pack_order inside the step) -- one warm-up + 2 timed steps each, round trip checked on all reads; bounded in wall time
    if rank == 0 and world == 1 and dist is None and args.workload == "c3" and not args.no_other_configs and not args.no_cpu:
        out["other_configs"] = {}
        t_other = time.perf_counter()
        # (round 6: c3r first -- configs[2] on a genome with a human-like repeat content, the input shape of the real configs[1] / [3]; one warm-up + ONE timed
        #  step of ~1.9 s, so that the repeat-bearing rate is the driver's number too and not only the builder's; behind them the two small configurations, configs[1]'s
        #  and configs[0]'s stand-ins, ten steps of 16 / 8 ms each)
        for name in ("c3r", "c4", "c5g", "c2", "c1"):
            if time.perf_counter() - t_other > 150.0:            # the side lines took long on this box: the line must not
                out["other_configs"][name] = {"skipped": "wall-time bound of the side lines reached"}
                continue
            try:
                out["other_configs"][name] = run_other_config(harc_amd, name, local, dev, args.shards, steps=1 if name == "c3r" else 10 if name in ("c1", "c2") else 2)
            except Exception as e:                                # a side line never takes the main line down
                out["other_configs"][name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        out["other_configs"]["wall_s"] = round(time.perf_counter() - t_other, 1)
    # ---- end to end, FASTQ file -> stream files (and back), on the same line, outside the timed region
    if rank == 0 and world == 1 and dist is None and ((args.workload == "c3" and not args.no_cpu and not args.no_other_configs) or args.end_to_end):
        t_e = time.perf_counter()
        try:
            out["end_to_end"] = end_to_end_leg(harc_amd, local, dev, args.shards, n=args.end_to_end or 100_000_000)
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)[:300]}
        out["end_to_end"]["leg_wall_s"] = round(time.perf_counter() - t_e, 1)
    if dist is not None:
        with wd.phase("final barrier (rank 0 runs the side legs meanwhile)", max(args.watchdog, 1800.0)):
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        real_stdout.write(json.dumps(out) + "\n")                 # the ONE JSON line: the only thing on stdout
        real_stdout.flush()


if __name__ == "__main__":
    main()
