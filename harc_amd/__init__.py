"""harc_amd -- MI355X (gfx950) implementation of HARC's reorder + encode hot path.

The product is the C-ABI shared library ``libharc_amd.so`` (include/harc_amd.h, sources in harc_amd/csrc).
This package is the thin ctypes binding used by the tests and bench.py; it mirrors the reference's stage
programs (``reorder.out`` / ``encoder.out`` / ``pack_order.out`` <basedir>, harc:65-69,112) and exposes the in-memory API.
There is no CPU fallback: importing works anywhere, every compute call needs a gfx950 device.
"""
from .api import (HarcAmd, HarcAmdError, Params, Counters, default_params, lib, lib_path, reorder, encoder, compress,
                  pack_order, preprocess, decoder, compress_fastq, STREAMS, comm_get_id, compress_fastq_shard, merge_shards, build_id, last_fastq_timing, build_has)
from ._build import build

__all__ = ["HarcAmd", "HarcAmdError", "Params", "Counters", "default_params", "lib", "lib_path", "reorder", "encoder",
           "compress", "pack_order", "preprocess", "decoder", "compress_fastq", "build", "STREAMS", "comm_get_id", "compress_fastq_shard",
           "merge_shards", "build_id", "last_fastq_timing", "build_has"]
