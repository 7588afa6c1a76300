"""Parity in the regime every BASELINE-sized run uses (run with -m gpu): MORE THAN 16 384 CHAINS -- k_reseed_mg, the dense SPEC kernel, 32 steps
per super-round by the index rule, the back-off of chains that keep losing bids on repeat-rich input -- the HIP path against the CPU oracle, every
stage-I and stage-II file byte for byte, at K = 20 000 and K = 65 536 (the cap, what configs[2] / [3] / [4] run with) x S = 16 / 32 / the library's own
choice, on (i) 2 M clean reads at 11x (configs[2]'s coverage) and (ii) 1 M repeat-rich reads (diverged copies of a repeat, poly-A runs).

What the regime stresses in the reference: seeding (reorder.cpp:476-491: K chains start K reads apart -- 15 to 100 reads per chain here, so most chains
run out and are reseeded again and again from the one descending cursor, reorder.cpp:650-688) and the claim race (reorder.cpp:545-552) that the
super-round arbitration replaces.

The oracle runs (eight of them, 10-40 s each on one core) are made side by side in threads before the first comparison: ctypes releases the GIL and the
oracle keeps no global state."""
import concurrent.futures as cf

import pytest

from tests import gen
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
L, E = 100, 8
INPUTS = {
    # name: (maker, steps per super-round the index rule picks from 16 385 chains on)
    "clean2M": (lambda: gen.lines_of(gen.reads_array_big(51, 2_000_000, L, 17_700_000, err=0.0)), 32),
    "rich1M": (lambda: gen.reads_text_lowcomplexity(99, 1_000_000, L, 2_500_000, n_repeat=1500, n_polya=12, err=0.004), 16),
}
KS = (20000, 65536)
SS = (16, 32)


@pytest.fixture(scope="module")
def oracle_runs(oracle, tmp_path_factory):
    """({input: its preprocessed files}, {(input, K, S): (stage-I files, stage-II files)}) for every combination, the oracle's runs side by side"""
    root = tmp_path_factory.mktemp("many_chains_oracle")
    txts = {name: mk() for name, (mk, _) in INPUTS.items()}

    def one(key):
        name, K, S = key
        d = root / f"{name}_{K}_{S}"
        d.mkdir()
        base = ol.stage_dir(d, {})
        txt = txts[name]
        assert oracle.harc_oracle_preprocess(txt, len(txt), L, base.encode()) == 0
        if key[1:] == (KS[0], SS[0]):                              # one copy of the inputs per input, not per run
            ins[name] = {k: v for k, v in ol.read_dir(base).items() if k in ("input_clean.dna", "numreads.bin", "input_N.dna")}
        assert oracle.harc_oracle_reorder(base.encode(), L, K, S, None, None) == 0
        s1 = {f: v for f, v in ol.read_dir(base).items() if f in ol.STAGE1_FILES}
        assert oracle.harc_oracle_encoder(base.encode(), L, E, None, None) == 0
        s2 = {f: v for f, v in ol.read_dir(base).items() if f in ol.stage2_files(E)}
        for f in (d / "output").iterdir():                        # the files are in memory now: the directory would hold ~300 MB per run
            f.unlink()
        return key, (s1, s2)
    ins = {}
    keys = [(name, K, S) for name in INPUTS for K in KS for S in SS]
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        runs = dict(ex.map(one, keys))
    return ins, runs


def _diff(name, a, b):
    if a == b:
        return None
    n = min(len(a), len(b))
    first = next((i for i in range(n) if a[i] != b[i]), n)
    return f"{name}: len {len(a)} vs {len(b)}, first difference at byte {first}: {a[first:first+16]!r} vs {b[first:first+16]!r}"


@pytest.mark.parametrize("S", [16, 32, 0])
@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("name", list(INPUTS))
def test_more_than_16384_chains_match_oracle(name, K, S, oracle_runs, tmp_path):
    """S = 0: the library's choice from the index (stage1_run_w: 32 where the bins of more than 16 reads hold less than 2 % of N entries, else 16) must be the
    oracle's run with that S -- a function of the input alone"""
    import harc_amd
    S_eff = S if S else INPUTS[name][1]
    s1, s2 = oracle_runs[1][(name, K, S_eff)]
    base = ol.stage_dir(tmp_path, oracle_runs[0][name])
    harc_amd.reorder(base, L, num_chains=K, num_steps=S)
    got = ol.read_dir(base)
    errs = [d for d in (_diff(f, got.get(f, b"<missing>"), s1[f]) for f in ol.STAGE1_FILES) if d]
    assert not errs, f"{name} K={K} S={S}: stage I vs oracle\n" + "\n".join(errs)
    harc_amd.encoder(base, L, num_thr=E)
    got = ol.read_dir(base)
    errs = [d for d in (_diff(f, got.get(f, b"<missing>"), s2[f]) for f in ol.stage2_files(E)) if d]
    assert not errs, f"{name} K={K} S={S} E={E}: stage II vs oracle\n" + "\n".join(errs)
