"""BASELINE.json's configurations at FULL size inside the GPU suite (run with -m gpu), through the C-ABI:

  configs[0]  1 M x 100 bp, 35 Mbp (2.9x)        exact mode (num_chains = 1) == the md5 of every file of the REFERENCE at -t 1
  configs[1]  3.3 M x 100 bp, 52x, 0.5 % errors  exact mode == the reference at -t 1 (tests/golden/md5_*.json, oracle/make_goldens.py);
                                                 throughput mode against the oracle: tests/test_gpu_config_size.py
  configs[2]  350 M x 100 bp error-free          throughput mode: round trip, two runs of one context and every forced kernel variant give
                                                 the same stream digest (harc_amd_stream_digest: every stage-II stream folded on the device)
  configs[3]  810 M x 101 bp, 1 % errors         one GPU (187 GB): round trip, digest equal between two runs
  configs[4]  one GPU's share, 500 M x 150 bp,   round trip, digest and packed order equal between two runs
              31 % N reads, -p (pack_order)

At these sizes the oracle cannot follow (it walks one chain step at a time on one core); what holds at any size is (i) losslessness -- the
multiset signature of the GPU-decoded streams equals the inputs' --, (ii) determinism -- the schedule is a function of the input alone, so two
runs, and two execution variants of the same schedule, must agree in every byte --, (iii) the counters.  The one race this code has had
(k_reseed_mg) showed only at 100 M reads and more."""
import os
import sys
import zlib

import numpy as np
import pytest

from tests import gen
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(workload, seed=1000, **kw):
    """a context holding bench.py's workload `workload` (the generator of the headline benchmark) -> (context, input signature, workload tuple)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import harc_amd
    n, L, G, err, _ = bench.WORKLOADS[workload]
    dev = torch.device("cuda", 0)
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8, num_chains=0, stream_digest=1, **kw))
    sig = bench.install_synthetic(h, n, L, G, err, seed, dev, bench.SPIKES.get(workload))
    return h, tuple(sig), (n, L, G, err)


def _run(h, pack_order=False):
    h.reorder(); h.encode()
    if pack_order:
        h.pack_order()
    c = h.counters()
    return h.stream_digest(), {k: getattr(c, k) for k in ("n_clean", "n_N", "n_main", "n_singleton", "unmatched", "aligned_singletons", "aligned_N", "chains", "rounds",
                                                         "conflicts", "contigs", "seq_bases", "bins_over_maxsearch")}


def _free():
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def test_configs2_350M_x_100bp(monkeypatch):
    """configs[2]: 350 M x 100 bp error-free on one GPU -- what bench.py times.  Lossless; two runs of one context agree; every execution variant
    a run of this size can take (lane-serial / wave-uniform scan of the small bins, k_reseed by one or by 64 workgroups, bitmap by atomics or
    by tiles, column counts applied per step or per run of agreeing steps) produces the same digest."""
    for k in ("HARC_AMD_SEQ", "HARC_AMD_RESEED_MG", "HARC_AMD_S1BLOOM_TILED", "HARC_AMD_LAZY", "HARC_AMD_SPEC"):
        monkeypatch.delenv(k, raising=False)
    h, sig_in, (n, L, G, err) = _ctx("c3")
    try:
        d0, c0 = _run(h)
        assert c0["n_clean"] == n and c0["n_main"] + c0["n_singleton"] == n and c0["chains"] == 65536
        assert h.decode_signature() == sig_in, "configs[2]: the decoded streams are not the input reads"
        d1, c1 = _run(h)
        assert d1 == d0 and c1 == c0, "configs[2]: two runs of one context differ"
        for env in ({"HARC_AMD_SEQ": "0"}, {"HARC_AMD_SEQ": "1"}, {"HARC_AMD_RESEED_MG": "0"}, {"HARC_AMD_S1BLOOM_TILED": "0"}, {"HARC_AMD_LAZY": "0"}, {"HARC_AMD_SPEC": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            dv, cv = _run(h)
            for k in env:
                monkeypatch.delenv(k)
            assert dv == d0 and cv == c0, f"configs[2]: variant {env} differs"
    finally:
        h.close()
        _free()


def test_configs3_810M_x_101bp_one_gpu():
    """configs[3] (ERR194146-size: 810 M x 101 bp, 26x, 1 % substitutions, 22 % of the reads with an N) on ONE GPU"""
    h, sig_in, (n, L, G, err) = _ctx("c4")
    try:
        d0, c0 = _run(h)
        assert c0["n_clean"] + c0["n_N"] == n and c0["n_main"] + c0["n_singleton"] == c0["n_clean"]
        assert c0["aligned_N"] > 0.5 * c0["n_N"], c0                  # at 26x most reads with N find their place in a contig
        assert h.decode_signature() == sig_in, "configs[3]: the decoded streams are not the input reads"
        d1, c1 = _run(h)
        assert d1 == d0 and c1 == c0, "configs[3]: two runs of one context differ"
    finally:
        h.close()
        _free()


def test_leftover_emission_is_stable_at_50M(monkeypatch):
    """a first list-based form of the kernel that writes the unaligned singletons / N reads (k_left_emit) differed between two launches on the
    same inputs at 50 M reads; the form that shipped (k_left_emit_w) and the pass over all candidates it replaced (HARC_AMD_LEFT_ALL=1) are held here to
    ONE digest, twice each, on 50 M x 101 bp with 1 % errors (c4s: 11 M reads with N, a tenth of them left unaligned)"""
    monkeypatch.delenv("HARC_AMD_LEFT_ALL", raising=False)
    h, sig_in, (n, L, G, err) = _ctx("c4s")
    try:
        d0, c0 = _run(h)
        assert h.decode_signature() == sig_in
        assert _run(h) == (d0, c0)
        monkeypatch.setenv("HARC_AMD_LEFT_ALL", "1")
        assert _run(h) == (d0, c0)
        assert _run(h) == (d0, c0)
    finally:
        h.close()
        _free()


@pytest.mark.parametrize("workload", ["c3r", "c4r"])
def test_human_like_repeats_at_baseline_size(workload, monkeypatch):
    """configs[2] / configs[3] with the repeat content the real data sets have and an i.i.d. genome lacks (SURVEY.md 8d "add a repeat-spiked variant";
    bench.py SPIKES: a tenth of the genome in a diverged family of 300-mers, a family of 6-kb elements, poly-A / (CA)n runs, tandem arrays of a 171-mer):
    bins of thousands of reads, cooperative walks, bins above maxsearch in stage II -- at 350 M / 810 M reads.  Lossless; two runs agree; the two scans of
    the small bins (HARC_AMD_SEQ=0/1), which a run of this kind chooses between by measuring, give the same digest."""
    monkeypatch.delenv("HARC_AMD_SEQ", raising=False)
    h, sig_in, (n, L, G, err) = _ctx(workload)
    try:
        d0, c0 = _run(h)
        assert c0["n_clean"] + c0["n_N"] == n and c0["n_main"] + c0["n_singleton"] == c0["n_clean"]
        assert h.decode_signature() == sig_in, f"{workload}: the decoded streams are not the input reads"
        # ... and so do the two forms of stage II's window passes over the bins above maxsearch (an event per lane -- the default -- and a wave per event)
        for env in ({"HARC_AMD_SEQ": "0"}, {"HARC_AMD_SEQ": "1"}, {"HARC_AMD_S2_BLOCK": "0", "HARC_AMD_S2_RANGE": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            dv, cv = _run(h)
            for k in env:
                monkeypatch.delenv(k)
            assert dv == d0 and cv == c0, f"{workload}: variant {env} differs"
    finally:
        h.close()
        _free()


def test_configs4_share_500M_x_150bp_pack_order():
    """configs[4]: what ONE of the 8 GPUs gets -- 500 M x 150 bp, 193x, 1 % substitutions (31 % N reads), -p: pack_order inside the run"""
    h, sig_in, (n, L, G, err) = _ctx("c5g")
    try:
        d0, c0 = _run(h, pack_order=True)
        assert c0["n_clean"] + c0["n_N"] == n
        assert h.decode_signature() == sig_in, "configs[4] share: the decoded streams are not the input reads"
        po = h.stream("P_ORDER")
        p0 = (zlib.crc32(po), len(po), h.stream("P_ORDER_TAIL"))
        del po
        nord = c0["n_clean"]                                           # read_order.bin: one entry per clean read (pack_order.cpp:20-77)
        numbits = int(nord).bit_length()
        assert p0[1] == 8 + (nord // 32) * numbits * 4 and len(p0[2]) == (nord % 32) * 4
        d1, c1 = _run(h, pack_order=True)
        po = h.stream("P_ORDER")
        p1 = (zlib.crc32(po), len(po), h.stream("P_ORDER_TAIL"))
        del po
        assert d1 == d0 and c1 == c0 and p1 == p0, "configs[4] share: two runs of one context differ"
    finally:
        h.close()
        _free()


def test_decoder_of_a_shard_beyond_2_32_text_bytes(tmp_path):
    """`decoder.out` (decoder.cpp:90-169) on ONE shard of 43 M reads x 100 bp: its text is 4.34 G characters, a thread each -- beyond the 2^32 work-items a
    one-dimensional grid holds on this platform (devutil.h: such a launch was cut short without an error; ./harc -d of a configs[2]-sized archive made with
    -t 8 has shards of this size).  The decoded file must be the input multiset."""
    import shutil
    import tempfile
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import harc_amd
    n, L = 43_000_000, 100
    assert n * (L + 1) > (1 << 32)
    dev = torch.device("cuda", 0)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (24 << 30) else None
    h = harc_amd.HarcAmd(harc_amd.default_params(L))
    with tempfile.TemporaryDirectory(dir=shm or str(tmp_path), prefix="harc_bigshard_") as td:
        od = os.path.join(td, "output"); os.makedirs(od)
        sig_in = [0, 0, 0]
        with open(os.path.join(od, "input_clean.dna"), "wb") as f:
            for r in bench.synth_chunks(n, L, 380_000_000, 0.0, 4321, dev):
                lines = torch.full((r.shape[0], L + 1), 10, dtype=torch.uint8, device=dev)
                lines[:, :L] = r
                torch.cuda.synchronize()
                c3 = h.reads_signature_device(lines.data_ptr(), lines.shape[0], L + 1)
                sig_in[0] += c3[0]; sig_in[1] = (sig_in[1] + c3[1]) % (1 << 64); sig_in[2] ^= c3[2]
                f.write(lines.cpu().numpy().tobytes())
                del lines, r
        open(os.path.join(od, "input_N.dna"), "wb").close()
        np.array([n], dtype=np.uint32).tofile(os.path.join(od, "numreads.bin"))
        torch.cuda.empty_cache()
        harc_amd.compress(td, L, num_thr=1, num_chains=0)
        harc_amd.decoder(td, 1)
        out = os.path.join(od, "output.dna")
        assert os.path.getsize(out) == n * (L + 1)
        sig = [0, 0, 0]
        CH = 4_000_000 * (L + 1)
        with open(out, "rb") as f:
            while True:
                b = f.read(CH)
                if not b:
                    break
                t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
                torch.cuda.synchronize()
                c3 = h.reads_signature_device(t.data_ptr(), len(b) // (L + 1), L + 1)
                sig[0] += c3[0]; sig[1] = (sig[1] + c3[1]) % (1 << 64); sig[2] ^= c3[2]
                del t
    h.close()
    _free()
    assert tuple(sig) == tuple(sig_in), "the decoded file of a 43 M-read shard is not the input multiset"


@pytest.mark.parametrize("case", ol.md5_cases())
def test_exact_mode_at_config_size(case, tmp_path):
    """north_star's "bit-exact vs CPU" at configs[0] / configs[1] size: num_chains = 1, num_thr = 1 against the md5 of every file the REFERENCE
    writes at -t 1 (reorder.cpp:455-689, encoder.cpp:154-616).  At 1 M / 3.3 M reads the bitmaps in front of the tables, bucket overflow and the
    large-bin machinery are live, which the <= 5 k-read fixtures of tests/golden/*.tar.xz do not reach."""
    import tempfile
    import harc_amd
    meta, arr = ol.load_md5_case(case)
    L = arr.shape[1]
    hasN = (arr == ord("N")).any(1)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm, prefix="harc_exact_") as td:
        base = ol.stage_dir(td, {"input_clean.dna": gen.lines_of(arr[~hasN]), "input_N.dna": gen.lines_of(arr[hasN]),
                                 "numreads.bin": np.array([int((~hasN).sum())], dtype=np.uint32).tobytes()})
        del arr
        assert not ol.md5_mismatches(ol.read_dir(base), meta["stage1"], ["input_clean.dna", "input_N.dna", "numreads.bin"]), "inputs differ from the reference's preprocess.out"
        harc_amd.reorder(base, L, num_chains=1)
        bad = ol.md5_mismatches(ol.read_dir(base), meta["stage1"], ol.STAGE1_FILES)
        assert not bad, f"{case}: stage I of exact mode differs from the reference at -t 1: {bad}"
        harc_amd.encoder(base, L, num_thr=1)
        bad = ol.md5_mismatches(ol.read_dir(base), meta["stage2"], ol.stage2_files(1))
        assert not bad, f"{case}: stage II of exact mode differs from the reference at -t 1: {bad}"
