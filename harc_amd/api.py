"""ctypes binding of include/harc_amd.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class HarcAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"harc_amd error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """harc_amd_params == src/config.h macros (harc:52-63)"""
    _fields_ = [("readlen", C.c_int32), ("num_thr", C.c_int32), ("num_chains", C.c_int32), ("maxmatch", C.c_int32),
                ("thresh", C.c_int32), ("thresh_s", C.c_int32), ("maxsearch", C.c_int32), ("dict_start", C.c_int32 * 2),
                ("dict_end", C.c_int32 * 2), ("device", C.c_int32), ("profile", C.c_int32), ("num_steps", C.c_int32), ("reads_per_chain", C.c_int32), ("decode_memory_gb", C.c_int32), ("stream_digest", C.c_int32), ("table_slots_per_read", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("n_clean", C.c_uint64), ("n_N", C.c_uint64), ("n_main", C.c_uint64), ("n_singleton", C.c_uint64),
                ("unmatched", C.c_uint64), ("aligned_singletons", C.c_uint64), ("aligned_N", C.c_uint64),
                ("chains", C.c_uint64), ("rounds", C.c_uint64), ("probes", C.c_uint64), ("candidates", C.c_uint64),
                ("conflicts", C.c_uint64), ("propose_launches", C.c_uint64), ("propose_ms", C.c_double),
                ("index_ms", C.c_double), ("chain_ms", C.c_double), ("encode_ms", C.c_double), ("total_ms", C.c_double),
                ("contigs", C.c_uint64), ("seq_bases", C.c_uint64), ("bins_over_maxsearch", C.c_uint64),
                ("device_bytes_peak", C.c_uint64), ("useful_probes", C.c_uint64), ("candidates_seq", C.c_uint64),
                ("coop_launches", C.c_uint64), ("coop_ms", C.c_double), ("coop_useful_probes", C.c_uint64), ("coop_candidates_seq", C.c_uint64),
                ("coop_candidates", C.c_uint64), ("coop_steps", C.c_uint64), ("dense_steps", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


STREAMS = dict(S1_ORDER=0, S1_FLAG=1, S1_POS=2, S1_RC=3, S1_ORDER_SINGLETON=4, S1_DNA=5, S1_DNA_SINGLETON=6,
               S2_SEQ=10, S2_SEQ_TAIL=11, S2_POS=12, S2_NOISE=13, S2_NOISEPOS=14, S2_REV=15, S2_REV_TAIL=16,
               S2_ORDER=20, S2_ORDER_N_PE=21, S2_SINGLETON=22, S2_SINGLETON_TAIL=23, S2_INPUT_N=24, S2_META=25,
               P_ORDER=30, P_ORDER_TAIL=31, IN_ORDER_N=40)

_lib = None


def lib_path():
    return os.environ.get("HARC_AMD_LIB") or os.path.join(HERE, "libharc_amd.so")      # override: experiment builds of the same sources


def lib():
    """Load libharc_amd.so; fails loudly when the HIP extension has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise HarcAmdError(-2, f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C harc_amd/csrc`; harc_amd has no CPU path")
    l = C.CDLL(p)
    PP = C.POINTER(Params)
    ctx = C.c_void_p
    l.harc_amd_default_params.argtypes = [C.c_int32, PP]
    l.harc_amd_create.argtypes = [PP, C.POINTER(ctx)]
    l.harc_amd_destroy.argtypes = [ctx]
    l.harc_amd_destroy.restype = None
    l.harc_amd_last_error.restype = C.c_char_p
    l.harc_amd_set_reads_ascii.argtypes = [ctx, C.c_char_p, C.c_uint32, C.c_uint32]
    l.harc_amd_set_reads_ascii_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32]
    l.harc_amd_set_reads_packed_device.argtypes = [ctx, C.c_void_p, C.c_uint32]
    l.harc_amd_set_nreads_ascii.argtypes = [ctx, C.c_char_p, C.c_uint32, C.c_uint32]
    l.harc_amd_set_nreads_ascii_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32]
    l.harc_amd_set_stage1_streams.argtypes = [ctx, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32,
                                              C.c_char_p, C.c_char_p, C.c_uint32]
    l.harc_amd_pack_reads_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    l.harc_amd_bucket_reads_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    l.harc_amd_partition_reads_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    for f in ("harc_amd_reorder", "harc_amd_encode", "harc_amd_pack_order"):
        getattr(l, f).argtypes = [ctx]
    l.harc_amd_get_stream.argtypes = [ctx, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    l.harc_amd_get_counters.argtypes = [ctx, C.POINTER(Counters)]
    for f in ("harc_amd_reorder_files", "harc_amd_encoder_files", "harc_amd_compress_files", "harc_amd_pack_order_files"):
        getattr(l, f).argtypes = [PP, C.c_char_p]
    l.harc_amd_preprocess_files.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
    l.harc_amd_decoder_files.argtypes = [PP, C.c_char_p, C.c_int32]
    l.harc_amd_compress_fastq_files_ex.argtypes = [PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]
    l.harc_amd_build_has.argtypes = [C.c_char_p]
    l.harc_amd_last_fastq_timing.argtypes = [C.POINTER(C.c_double), C.c_int32]
    l.harc_amd_decoder_preserve_files.argtypes = [PP, C.c_char_p, C.c_int32]
    l.harc_amd_compress_fastq_files.argtypes = [PP, C.c_char_p, C.c_char_p]
    l.harc_amd_set_fastq_device.argtypes = [ctx, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    l.harc_amd_decode_signature.argtypes = [ctx, C.POINTER(C.c_uint64)]
    l.harc_amd_input_signature.argtypes = [ctx, C.POINTER(C.c_uint64)]
    l.harc_amd_reads_signature_device.argtypes = [ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    l.harc_amd_comm_get_id.argtypes = [C.c_char_p, C.c_size_t]
    l.harc_amd_comm_init.argtypes = [ctx, C.c_char_p, C.c_size_t, C.c_int32, C.c_int32]
    l.harc_amd_comm_init_mailbox.argtypes = [ctx, C.c_char_p, C.c_int32, C.c_int32]
    l.harc_amd_comm_barrier.argtypes = [ctx]
    l.harc_amd_comm_destroy.argtypes = [ctx]
    l.harc_amd_shard_exchange.argtypes = [ctx, C.POINTER(C.c_uint64)]
    l.harc_amd_shard_reset.argtypes = [ctx]
    l.harc_amd_replicate_exchange.argtypes = [ctx, C.POINTER(C.c_uint64)]
    l.harc_amd_compress_fastq_shard_files.argtypes = [PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p]
    l.harc_amd_merge_shard_files.argtypes = [C.c_char_p, C.c_int32]
    l.harc_amd_stream_digest.argtypes = [ctx, C.POINTER(C.c_uint64)]
    l.harc_amd_selftest_launch.argtypes = [ctx, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    l.harc_amd_build_id.restype = C.c_char_p
    _lib = l
    return l


def _check(rc):
    if rc != 0:
        raise HarcAmdError(rc, lib().harc_amd_last_error().decode(errors="replace"))


def default_params(readlen, num_thr=8, num_chains=0, device=0, profile=0, num_steps=0, reads_per_chain=0, stream_digest=0, table_slots_per_read=0):
    p = Params()
    _check(lib().harc_amd_default_params(readlen, C.byref(p)))
    p.num_thr, p.num_chains, p.device, p.profile, p.num_steps = num_thr, num_chains, device, profile, num_steps
    p.reads_per_chain = reads_per_chain
    p.stream_digest = stream_digest
    p.table_slots_per_read = table_slots_per_read      # 0: the library chooses (4, 3 or 2 by free memory); include/harc_amd.h
    return p


def build_id():
    """sha256 of the kernel sources the loaded library was built from"""
    return lib().harc_amd_build_id().decode()


# ---- the reference's stage programs (file contract)
def reorder(basedir, readlen, num_chains=1, **kw):
    """== `reorder.out <basedir>` (src/reorder.cpp:100-131)"""
    p = default_params(readlen, num_chains=num_chains, **kw)
    _check(lib().harc_amd_reorder_files(C.byref(p), os.fsencode(basedir)))


def encoder(basedir, readlen, num_thr=1, **kw):
    """== `encoder.out <basedir>` (src/encoder.cpp:108-152)"""
    p = default_params(readlen, num_thr=num_thr, **kw)
    _check(lib().harc_amd_encoder_files(C.byref(p), os.fsencode(basedir)))


def compress(basedir, readlen, num_thr=1, num_chains=1, **kw):
    """harc:65-69 fused: stage I -> stage II in HBM"""
    p = default_params(readlen, num_thr=num_thr, num_chains=num_chains, **kw)
    _check(lib().harc_amd_compress_files(C.byref(p), os.fsencode(basedir)))


def preprocess(fastq, basedir, readlen):
    """== `preprocess.out <fastq> <basedir> False False <readlen>` (src/preprocess.cpp:22-137): the N split"""
    _check(lib().harc_amd_preprocess_files(os.fsencode(fastq), os.fsencode(basedir), readlen))


def compress_fastq(fastq, basedir, readlen, num_thr=1, num_chains=1, preserve_order=False, preserve_quality=False, **kw):
    """harc:50-69 in one call, FASTQ parsed on the GPU; preserve_quality (-q) also writes output.quality / output.id
    (preprocess.cpp:61-118, reorder_quality.cpp)"""
    p = default_params(readlen, num_thr=num_thr, num_chains=num_chains, **kw)
    _check(lib().harc_amd_compress_fastq_files_ex(C.byref(p), os.fsencode(fastq), os.fsencode(basedir), int(preserve_order), int(preserve_quality)))


def build_has(feature):
    """True when the loaded library was built with the optional part `feature` ("grp", "test_transport", "experiments")"""
    return bool(lib().harc_amd_build_has(feature.encode()))


def last_fastq_timing():
    """seconds of the last compress_fastq of this process, by phase (include/harc_amd.h: harc_amd_last_fastq_timing)"""
    t = (C.c_double * 8)()
    _check(lib().harc_amd_last_fastq_timing(t, 8))
    names = ("context_and_pool", "ingest", "ingest_waiting_for_file_readers", "ingest_device_passes", "reorder", "encode", "stream_files", "total")
    return {k: float(v) for k, v in zip(names, t)}


COMM_ID_BYTES = 128


def comm_get_id():
    """rank 0 of a multi-GPU run: the bytes every rank hands to HarcAmd.comm_init (== ncclGetUniqueId)"""
    b = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().harc_amd_comm_get_id(b, COMM_ID_BYTES))
    return b.raw


def compress_fastq_shard(fastq, basedir, readlen, world, rank, comm_spec, num_thr=1, num_chains=0, preserve_order=False,
                         preserve_quality=False, **kw):
    """one rank of `./harc -c -g <world>`: slice of the FASTQ -> exchange -> this GPU's shard files"""
    kw.setdefault("reads_per_chain", 1024)
    p = default_params(readlen, num_thr=num_thr, num_chains=num_chains, **kw)
    _check(lib().harc_amd_compress_fastq_shard_files(C.byref(p), os.fsencode(fastq), os.fsencode(basedir), int(preserve_order),
                                                     int(preserve_quality), world, rank, os.fsencode(comm_spec)))


def merge_shards(basedir, world):
    """after every rank has finished: the whole-job files of the archive (host code)"""
    _check(lib().harc_amd_merge_shard_files(os.fsencode(basedir), world))


def decoder(basedir, num_thr_e, device=0, preserve_order=False, memory_gb=0):
    """== `decoder.out <basedir> <num_thr> <num_thr_e>` (src/decoder.cpp:44-172): output/output.dna;
    preserve_order: the -p chain unpack_order + decoder_preserve + merge_N (harc:183-185)"""
    p = default_params(100, device=device)
    p.decode_memory_gb = memory_gb
    f = lib().harc_amd_decoder_preserve_files if preserve_order else lib().harc_amd_decoder_files
    _check(f(C.byref(p), os.fsencode(basedir), num_thr_e))


def pack_order(basedir, readlen=100, **kw):
    """== `pack_order.out <basedir>` (src/pack_order.cpp:11-77)"""
    p = default_params(readlen, **kw)
    _check(lib().harc_amd_pack_order_files(C.byref(p), os.fsencode(basedir)))


class HarcAmd:
    """In-memory API: one context = one HIP device + stream."""

    def __init__(self, params):
        self.params = params
        self._ctx = C.c_void_p()
        _check(lib().harc_amd_create(C.byref(params), C.byref(self._ctx)))

    def close(self):
        if self._ctx:
            lib().harc_amd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reads_ascii(self, buf, n, stride):
        _check(lib().harc_amd_set_reads_ascii(self._ctx, buf, n, stride))

    def set_reads_ascii_device(self, dptr, n, stride):
        _check(lib().harc_amd_set_reads_ascii_device(self._ctx, C.c_void_p(dptr), n, stride))

    def set_fastq_device(self, dptr, nbytes):
        nrec = C.c_uint64(0)
        _check(lib().harc_amd_set_fastq_device(self._ctx, C.c_void_p(dptr), nbytes, C.byref(nrec)))
        return nrec.value

    def set_reads_packed_device(self, dptr, n):
        _check(lib().harc_amd_set_reads_packed_device(self._ctx, C.c_void_p(dptr), n))

    def set_nreads_ascii(self, buf, n, stride):
        _check(lib().harc_amd_set_nreads_ascii(self._ctx, buf, n, stride))

    def set_nreads_ascii_device(self, dptr, n, stride):
        _check(lib().harc_amd_set_nreads_ascii_device(self._ctx, C.c_void_p(dptr), n, stride))

    def pack_reads_device(self, d_ascii, n, stride, d_out):
        _check(lib().harc_amd_pack_reads_device(self._ctx, C.c_void_p(d_ascii), n, stride, C.c_void_p(d_out)))

    def partition_reads_device(self, d_packed, n, n_buckets, d_out, d_counts):
        _check(lib().harc_amd_partition_reads_device(self._ctx, C.c_void_p(d_packed), n, n_buckets, C.c_void_p(d_out), C.c_void_p(d_counts)))

    def bucket_reads_device(self, d_packed, n, n_buckets, d_out):
        _check(lib().harc_amd_bucket_reads_device(self._ctx, C.c_void_p(d_packed), n, n_buckets, C.c_void_p(d_out)))

    def comm_init(self, comm_id, world, rank):
        """join the RCCL communicator of the run (comm_id from comm_get_id() on rank 0)"""
        _check(lib().harc_amd_comm_init(self._ctx, comm_id, len(comm_id), world, rank))

    def comm_init_mailbox(self, directory, world, rank):
        _check(lib().harc_amd_comm_init_mailbox(self._ctx, os.fsencode(directory), world, rank))

    def comm_barrier(self):
        _check(lib().harc_amd_comm_barrier(self._ctx))

    def comm_destroy(self):
        _check(lib().harc_amd_comm_destroy(self._ctx))

    def shard_exchange(self):
        """bucket the context's own reads, ONE all-to-all(v); -> info tuple (see include/harc_amd.h)"""
        info = (C.c_uint64 * 8)()
        _check(lib().harc_amd_shard_exchange(self._ctx, info))
        return tuple(int(x) for x in info)

    def replicate_exchange(self):
        """design (R): all-gather every rank's slice; reorder() then partitions the chains over the ranks and every rank ends with the
        single-GPU result of the whole job; -> info tuple (see include/harc_amd.h)"""
        info = (C.c_uint64 * 8)()
        _check(lib().harc_amd_replicate_exchange(self._ctx, info))
        return tuple(int(x) for x in info)

    def shard_reset(self):
        """the stages read this context's own slice again (no exchange); results of the last run go"""
        _check(lib().harc_amd_shard_reset(self._ctx))

    def reorder(self):
        _check(lib().harc_amd_reorder(self._ctx))

    def encode(self):
        _check(lib().harc_amd_encode(self._ctx))

    def pack_order(self):
        _check(lib().harc_amd_pack_order(self._ctx))

    def decode_signature(self):
        """(count, sum, xor) of the reads decoded on the GPU from this context's stage-II streams"""
        sig = (C.c_uint64 * 3)()
        _check(lib().harc_amd_decode_signature(self._ctx, sig))
        return tuple(int(x) for x in sig)

    def input_signature(self):
        sig = (C.c_uint64 * 3)()
        _check(lib().harc_amd_input_signature(self._ctx, sig))
        return tuple(int(x) for x in sig)

    def reads_signature_device(self, d_ascii, n, stride):
        sig = (C.c_uint64 * 3)()
        _check(lib().harc_amd_reads_signature_device(self._ctx, C.c_void_p(d_ascii), n, stride, sig))
        return tuple(int(x) for x in sig)

    def stream(self, name, shard=0):
        ptr, ln = C.c_void_p(), C.c_size_t()
        _check(lib().harc_amd_get_stream(self._ctx, STREAMS[name], shard, C.byref(ptr), C.byref(ln)))
        return C.string_at(ptr, ln.value) if ln.value else b""

    def selftest_launch(self, n):
        """(visited, sum of indices mod 2^64) of a thread-per-item launch over n items through the library's launch geometry"""
        v, x = C.c_uint64(0), C.c_uint64(0)
        _check(lib().harc_amd_selftest_launch(self._ctx, n, C.byref(v), C.byref(x)))
        return int(v.value), int(x.value)

    def stream_digest(self):
        """four 64-bit words over the stage-II streams of the last encode(), folded on the device (params.stream_digest = 1)"""
        d = (C.c_uint64 * 4)()
        _check(lib().harc_amd_stream_digest(self._ctx, d))
        return tuple(int(x) for x in d)

    def counters(self):
        c = Counters()
        _check(lib().harc_amd_get_counters(self._ctx, C.byref(c)))
        return c
