"""CPU model of one multi-GPU run (test infrastructure): the exchange plan of tests/bucket_ref.shard_plan + the oracle on every
shard + the rank-part files harc_amd_compress_fastq_shard_files leaves under output/.shard/ (csrc/ingest.hip), so that
harc_amd_merge_shard_files (host code) and the decoders can be checked without a GPU, and the GPU ranks against the model."""
import os
import shutil

import numpy as np

from tests import oracle_lib as ol
from tests.bucket_ref import shard_plan


def slices_of(reads_arr, world):
    """file-order slices the way the ranks cut a FASTQ of equal-sized records: record boundaries at/after rank * size // world"""
    n = reads_arr.shape[0]
    return [reads_arr[n * r // world: n * (r + 1) // world] for r in range(world)]


def lines(arr):
    n, L = arr.shape
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = arr
    out[:, L] = 10
    return out.tobytes()


def oracle_shard(oracle, plan_r, L, E, K, S, workdir):
    """the oracle on one shard -> dict of its files, order streams mapped to global ids"""
    os.makedirs(workdir, exist_ok=True)
    base = ol.stage_dir(workdir, {"input_clean.dna": lines(plan_r["clean"]), "input_N.dna": lines(plan_r["withN"]),
                                   "numreads.bin": np.array([plan_r["clean"].shape[0]], dtype=np.uint32).tobytes()})
    um = (np.ctypeslib.ctypes.c_uint32 * 1)()
    assert oracle.harc_oracle_reorder(base.encode(), L, K, S, um, None) == 0
    ms, mn = (np.ctypeslib.ctypes.c_uint32 * 1)(), (np.ctypeslib.ctypes.c_uint32 * 1)()
    assert oracle.harc_oracle_encoder(base.encode(), L, E, ms, mn) == 0
    f = ol.read_dir(base)
    o = np.frombuffer(f["read_order.bin"], dtype=np.uint32)
    on = np.frombuffer(f["read_order_N_pe.bin"], dtype=np.uint32)
    f["read_order.bin"] = plan_r["gid"][o].astype(np.uint32).tobytes() if o.size else b""
    f["read_order_N_pe.bin"] = plan_r["ngid"][on].astype(np.uint32).tobytes() if on.size else b""
    f["_counters"] = (int(um[0]), int(ms[0]), int(mn[0]))
    return f


def write_rank_parts(od, rank, E, L, files, slice_arr, rec_off):
    """what rank `rank` writes: its shard family with suffix rank*E + e, and its parts under .shard/"""
    sd = os.path.join(od, ".shard")
    os.makedirs(sd, exist_ok=True)
    for e in range(E):
        for stem in ["read_seq.txt", "read_pos.txt", "read_noise.txt", "read_noisepos.txt", "read_rev.txt"]:
            open(os.path.join(od, f"{stem}.{rank * E + e}"), "wb").write(files[f"{stem}.{e}"])
        for stem in ["read_seq.txt", "read_rev.txt"]:
            open(os.path.join(od, f"{stem}.{rank * E + e}.tail"), "wb").write(files[f"{stem}.{e}.tail"])
    US = (4 * len(files["read_singleton.txt"]) + len(files["read_singleton.txt.tail"])) // L
    UN = len(files["input_N.dna"]) // (L + 1)
    o, on = files["read_order.bin"], files["read_order_N_pe.bin"]
    hasN = (slice_arr == ord("N")).any(1)
    parts = {"order_a": o[:len(o) - 4 * US], "order_u": o[len(o) - 4 * US:], "orderN_a": on[:len(on) - 4 * UN], "orderN_u": on[len(on) - 4 * UN:],
             "singleton": files["read_singleton.txt"], "singleton_tail": files["read_singleton.txt.tail"], "input_N": files["input_N.dna"],
             "order_N": (np.nonzero(hasN)[0] + rec_off).astype(np.uint32).tobytes()}
    for k, v in parts.items():
        open(os.path.join(sd, f"{k}.{rank}"), "wb").write(v)
    um, ms, mn = files["_counters"]
    nclean = int((~hasN).sum())
    open(os.path.join(sd, f"stats.{rank}"), "w").write(f"{L} {slice_arr.shape[0]} {nclean} {int(hasN.sum())} {um} {ms} {mn} 0 0\n")


def model_run(oracle, reads_arr, L, world, E, K, S, workdir):
    """-> (basedir with the UNMERGED output/ of a `world`-rank run, plan, per-rank oracle files)"""
    sl = slices_of(reads_arr, world)
    plan = shard_plan(sl, L, world)
    base = os.path.join(str(workdir), "job")
    od = os.path.join(base, "output")
    os.makedirs(od)
    per_rank = []
    rec_off = 0
    for r in range(world):
        f = oracle_shard(oracle, plan[r], L, E, K, S, os.path.join(str(workdir), f"rank{r}"))
        write_rank_parts(od, r, E, L, f, sl[r], rec_off)
        rec_off += sl[r].shape[0]
        per_rank.append(f)
    return base, plan, per_rank
