"""Multi-GPU archive contract on the CPU: the rank parts of a 2- and 3-rank run (modelled with the oracle, tests/shard_model.py)
merged by harc_amd_merge_shard_files (host code of libharc_amd.so) must decode -- with the oracle's decoder.cpp restatement and,
with -p, with the REAL reference's unpack_order / decoder_preserve / merge_N (oracle/_ref) -- to the input file."""
import os
import subprocess

import numpy as np
import pytest

from tests import gen, shard_model
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.parametrize("world,E,n,L,err", [(2, 1, 6000, 100, 0.01), (3, 1, 5001, 100, 0.02), (2, 2, 4000, 63, 0.01), (2, 1, 3000, 100, 0.0)])
def test_merged_archive_decodes_to_the_input(world, E, n, L, err, oracle, tmp_path):
    import harc_amd
    arr = gen.reads_array(31 + world, n, L, 12 * n, err=err)
    base, plan, per_rank = shard_model.model_run(oracle, arr, L, world, E, 7, 16, tmp_path)
    # every read of the job is in exactly one shard
    assert sum(p["clean"].shape[0] + p["withN"].shape[0] for p in plan) == n
    assert sorted(np.concatenate([p["gid"] for p in plan]).tolist()) == list(range(int(sum(p["clean"].shape[0] for p in plan))))
    harc_amd.merge_shards(base, world)
    od = os.path.join(base, "output")
    assert not os.path.exists(os.path.join(od, ".shard"))
    got = ol.read_dir(base)
    total_clean = int((~(arr == ord("N")).any(1)).sum())
    assert np.frombuffer(got["numreads.bin"], dtype=np.uint32)[0] == total_clean
    assert got["read_meta.txt"] == b"%d\n" % L
    # read_order.bin is a permutation of the clean ids, read_order_N_pe.bin of the N ids, read_order_N.bin the records with N
    assert sorted(np.frombuffer(got["read_order.bin"], dtype=np.uint32).tolist()) == list(range(total_clean))
    assert sorted(np.frombuffer(got["read_order_N_pe.bin"], dtype=np.uint32).tolist()) == list(range(n - total_clean))
    assert np.frombuffer(got["read_order_N.bin"], dtype=np.uint32).tolist() == np.nonzero((arr == ord("N")).any(1))[0].tolist()
    # singleton stream re-packed across the joints
    sing = b"".join(f["read_singleton.txt"] for f in per_rank)
    if all(len(f["read_singleton.txt.tail"]) == 0 for f in per_rank):
        assert got["read_singleton.txt"] == sing
    # (1) plain decode: the multiset of the input
    assert oracle.harc_oracle_decoder(base.encode(), world * E) == 0
    dec = open(os.path.join(od, "output.dna"), "rb").read()
    assert sorted(dec.split()) == sorted(bytes(r) for r in arr)
    os.remove(os.path.join(od, "output.dna"))
    # (2) -p: pack_order, then the reference's own chain restores the file order
    dp = os.path.join(REF, "decoder_preserve_L%d_e%d.out" % (L, world * E))
    if not os.path.exists(dp):
        pytest.skip("oracle/_ref has no decoder_preserve for L=%d, %d shards" % (L, world * E))
    assert oracle.harc_oracle_pack_order(base.encode()) == 0
    for exe in ("unpack_order.out", os.path.basename(dp), "merge_N.out"):
        subprocess.run([os.path.join(REF, exe), base], check=True, stdout=subprocess.DEVNULL)
    assert open(os.path.join(od, "output.dna"), "rb").read() == shard_model.lines(arr)


def test_merge_refuses_a_missing_rank(oracle, tmp_path):
    import harc_amd
    arr = gen.reads_array(5, 2000, 100, 20000, err=0.01)
    base, _, _ = shard_model.model_run(oracle, arr, 100, 2, 1, 3, 16, tmp_path)
    os.remove(os.path.join(base, "output", ".shard", "stats.1"))
    with pytest.raises(harc_amd.HarcAmdError) as e:
        harc_amd.merge_shards(base, 2)
    assert "rank 1" in str(e.value)
