"""Builds libharc_amd.so / harc_amd_stage in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(force=False, jobs=4):
    so = os.path.join(HERE, "libharc_amd.so")
    if force and os.path.exists(so):
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}"])
    if not os.path.exists(so):
        raise RuntimeError("libharc_amd.so was not produced")
    return so
