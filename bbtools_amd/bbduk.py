"""
Python binding over the C ABI (include/bbduk_gpu.h, include/bbduk_host.h).

Names follow the reference: `BBDuk(args)` takes the bbduk.sh key=value string (bbduk/BBDukParser.java),
`load_refs` / `build_index` play BBDukLoader.loadIndex (bbduk/BBDukLoader.java:82-103), and
`ktrim` / `kfilter` are the batch forms of BBDukProcessorS.ktrim / countSetKmers plus the k-mer stage of
processList (bbduk/BBDukProcessorS.java:948-1093).  There is no CPU fallback: if libbbduk_hip.so is
missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

from ._build import lib_path

OK = 0
MODE_KFILTER, MODE_KTRIM_R, MODE_KTRIM_L, MODE_KMASK, MODE_KTRIM_TIPS, MODE_KSPLIT = 0, 1, 2, 3, 4, 5
FLAG_DISCARDED, FLAG_REMOVED = 1, 2
NCOUNTERS = 16
HOOK_FORCE_TILE, HOOK_BUCKET_BITS, HOOK_LDS_BITS, HOOK_TIMING_MASK, HOOK_BIG_LAYOUT, HOOK_PAIR_SCAN, HOOK_SEED_LAYOUT, HOOK_BIG_LOAD = 1, 2, 3, 4, 5, 6, 7, 8      # include/bbduk_test_hooks.h
COUNTER_NAMES = ["readsIn", "basesIn", "readsKTrimmed", "basesKTrimmed", "readsKFiltered", "basesKFiltered",
                 "readsOutu", "basesOutu", "readsOutm", "basesOutm"]
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data")


class BBDukError(RuntimeError):
    pass


class Params(C.Structure):          # struct bbduk_params
    _fields_ = [("abi_version", C.c_int32), ("mode", C.c_int32), ("k", C.c_int32), ("mink", C.c_int32),
                ("rcomp", C.c_int32), ("forbidNs", C.c_int32), ("minlen", C.c_int32), ("minlen2", C.c_int32),
                ("middleMask", C.c_int64), ("qhdist", C.c_int32), ("qhdist2", C.c_int32),
                ("maxBadKmers", C.c_int32), ("minReadLength", C.c_int32), ("minLenFraction", C.c_float),
                ("removePairsIfEitherBad", C.c_int32), ("trimPad", C.c_int32), ("ktrimExclusive", C.c_int32),
                ("restrictLeft", C.c_int32), ("restrictRight", C.c_int32), ("skipR1", C.c_int32),
                ("skipR2", C.c_int32), ("numScaffolds", C.c_int32), ("device", C.c_int32),
                ("trimPairsEvenly", C.c_int32), ("qSkip", C.c_int32), ("speed", C.c_int32),
                ("minKmerFraction", C.c_float), ("minCoveredFraction", C.c_float), ("kbig", C.c_int32), ("findBestMatch", C.c_int32),
                ("kmaskFullyCovered", C.c_int32), ("trimFailuresTo1bp", C.c_int32), ("reserved0", C.c_int32)]


class SynthParams(C.Structure):     # struct bbduk_synth_params
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_int32), ("ins_min", C.c_int32), ("ins_max", C.c_int32),
                ("adapter1_len", C.c_int32), ("adapter2_len", C.c_int32), ("adapter1", C.c_char_p),
                ("adapter2", C.c_char_p), ("sub_rate_q32", C.c_uint32), ("n_rate_q32", C.c_uint32),
                ("contam_frac_q32", C.c_uint32), ("contam_len", C.c_int64), ("contam", C.c_char_p)]


# every symbol include/bbduk_gpu.h and include/bbduk_host.h declare
GPU_SYMBOLS = ["bbduk_abi_version", "bbduk_create", "bbduk_destroy", "bbduk_last_error", "bbduk_upload_table_way",
               "bbduk_upload_pairs", "bbduk_finalize_table", "bbduk_build_table_device", "bbduk_build_table_device_edits", "bbduk_table_size", "bbduk_table_bytes",
               "bbduk_table_lookup", "bbduk_ktrim_batch", "bbduk_kfilter_batch", "bbduk_ktrim_batch_device",
               "bbduk_kfilter_batch_device", "bbduk_pack_bases_host", "bbduk_pack_bases_device", "bbduk_ktrim_batch_packed",
               "bbduk_kfilter_batch_packed", "bbduk_ktrim_batch_packed_device", "bbduk_kfilter_batch_packed_device", "bbduk_kmask_batch", "bbduk_kmask_batch_device", "bbduk_ktrimtips_batch", "bbduk_ktrimtips_batch_device", "bbduk_ksplit_batch", "bbduk_kfilter_batch_matches", "bbduk_kfilter_batch_matches_device", "bbduk_ksplit_batch_device", "bbduk_kmask_batch_packed_device", "bbduk_ktrimtips_batch_packed_device", "bbduk_fastq_ingest_device", "bbduk_fastq_write_device", "bbduk_fastq_write_masked_device", "bbduk_device_malloc", "bbduk_device_free", "bbduk_pinned_malloc", "bbduk_pinned_free",
               "bbduk_copy_to_device", "bbduk_copy_from_device", "bbduk_device_memset", "bbduk_stream_create", "bbduk_stream_destroy", "bbduk_stream_synchronize", "bbduk_copy_async", "bbduk_kernel_time_ms", "bbduk_counters_len", "bbduk_get_counters", "bbduk_reset_counters",
               "bbduk_synth_generate_device", "bbduk_synth_generate_host", "bbduk_synth_pair_inserts",
               "bbduk_comm_preload", "bbduk_comm_unique_id", "bbduk_comm_create", "bbduk_comm_create_local", "bbduk_comm_destroy", "bbduk_comm_size",
               "bbduk_allreduce_counters", "bbduk_allreduce_counters_device", "bbduk_allreduce_counters_local",
               "bbduk_test_hook", "bbduk_table_spilled", "bbduk_table_line_histogram", "bbduk_table_layout", "bbduk_build_begin", "bbduk_build_add_device", "bbduk_build_end"]
HOST_SYMBOLS = ["bbduk_host_parse", "bbduk_host_destroy", "bbduk_host_add_ref", "bbduk_host_load_fasta",
                "bbduk_host_load_refs", "bbduk_host_build_index", "bbduk_host_index_pairs",
                "bbduk_host_num_scaffolds", "bbduk_host_scaffold_info", "bbduk_host_num_refs", "bbduk_host_ref_info", "bbduk_host_params", "bbduk_host_upload_index", "bbduk_host_build_on_device"]

_lib = None
_lib_override = None


def use_library(path):
    """Experiments only (profiles/ab_*.py): bind to another build of the same sources, e.g. bbtools_amd/ab_tsw.so.  Must be
    called before the first lib(); the product path never calls it."""
    global _lib_override
    if _lib is not None:
        raise BBDukError("library already loaded")
    _lib_override = path


class FastqResult(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_reads", "total_bases", "consumed1", "consumed2", "first_bad_read")]


class FastqBatch:
    """Device-resident result of fastq_ingest_device: torch tensors (HBM) + the host-side counts."""
    pass


def fastq_ingest_device(d_text1, d_text2=None, is_final=True, max_reads=None, device=0, stream_ptr=0):
    """Raw FASTQ text in HBM (uint8 torch tensors, one or two files) -> FastqBatch with lines1/lines2 (int64 line offsets),
    offsets (int64 base offsets), codes/undef (packed boundary format), n, total_bases, consumed (per text)."""
    import torch
    ns = 2 if d_text2 is not None else 1
    nb1 = d_text1.numel(); nb2 = d_text2.numel() if ns == 2 else 0
    if max_reads is None:
        max_reads = (max(nb1, nb2) // 4 + 1) * ns          # a record has at least four bytes
    cap_bases = (nb1 + nb2) // 2 + 32
    kw = dict(device=d_text1.device)
    fb = FastqBatch()
    fb.lines1 = torch.empty(4 * (max_reads // ns) + 1, dtype=torch.int64, **kw)
    fb.lines2 = torch.empty(4 * (max_reads // ns) + 1, dtype=torch.int64, **kw) if ns == 2 else None
    fb.offsets = torch.empty(max_reads + 1, dtype=torch.int64, **kw)
    fb.codes = torch.empty(cap_bases // 16 + 8, dtype=torch.int32, **kw)
    fb.undef = torch.empty(cap_bases // 32 + 8, dtype=torch.int32, **kw)
    res = FastqResult()
    rc = lib().bbduk_fastq_ingest_device(d_text1.data_ptr(), nb1, d_text2.data_ptr() if ns == 2 else None, nb2, int(bool(is_final)),
                                         max_reads, cap_bases, fb.lines1.data_ptr(), fb.lines2.data_ptr() if ns == 2 else None,
                                         fb.offsets.data_ptr(), fb.codes.data_ptr(), fb.undef.data_ptr(), device, stream_ptr, C.byref(res))
    fb.rc = rc; fb.n = res.n_reads; fb.total_bases = res.total_bases; fb.consumed = (res.consumed1, res.consumed2)
    fb.first_bad_read = res.first_bad_read
    if rc != 0:
        raise BBDukError("bbduk_fastq_ingest_device rc=%d (first bad read %d)" % (rc, res.first_bad_read))
    return fb


def fastq_write_device(d_text1, fb, d_left, d_right, d_flags, want_removed, d_out, d_text2=None, device=0, stream_ptr=0):
    """Selected reads of an ingested batch back to FASTQ text in d_out (uint8 torch tensor); returns the byte count."""
    nbytes = C.c_int64(0)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = lib().bbduk_fastq_write_device(p(d_text1), p(fb.lines1), p(d_text2), p(fb.lines2), fb.n, p(d_left), p(d_right), p(d_flags),
                                        int(bool(want_removed)), p(d_out), d_out.numel(), device, stream_ptr, C.byref(nbytes))
    if rc != 0:
        raise BBDukError("bbduk_fastq_write_device rc=%d (needs %d bytes)" % (rc, nbytes.value))
    return nbytes.value


def pack_bases_host(bases):
    """ASCII bases -> (codes uint32[(n+15)/16], undef uint32[(n+31)/32]) of the packed boundary format."""
    bases = np.ascontiguousarray(bases, np.uint8)
    n = len(bases)
    codes = np.zeros((n + 15) // 16, np.uint32); undef = np.zeros((n + 31) // 32, np.uint32)
    rc = lib().bbduk_pack_bases_host(bases.ctypes.data if n else None, n, codes.ctypes.data if n else None, undef.ctypes.data if n else None)
    if rc != 0:
        raise RuntimeError("bbduk_pack_bases_host failed (%d)" % rc)
    return codes, undef


def pack_bases_device(d_bases, d_codes, d_undef, device=0, stream_ptr=0):
    rc = lib().bbduk_pack_bases_device(d_bases.data_ptr(), d_bases.numel(), d_codes.data_ptr(), d_undef.data_ptr(), device, stream_ptr)
    if rc != 0:
        raise RuntimeError("bbduk_pack_bases_device failed (%d)" % rc)


def lib():
    """Loads libbbduk_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _lib_override or lib_path()
    if not os.path.exists(path):
        raise BBDukError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for this path)" % path)
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that this library
    # binds to the HIP runtime torch already loaded: one runtime per process, device pointers interchangeable.
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.bbduk_abi_version.restype = C.c_int
    L.bbduk_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.bbduk_destroy.argtypes = [vp]
    L.bbduk_last_error.restype = C.c_char_p
    L.bbduk_last_error.argtypes = [vp]
    L.bbduk_upload_table_way.argtypes = [vp, i32, i32, vp, vp, i64, vp, vp, i64]
    L.bbduk_upload_pairs.argtypes = [vp, vp, vp, i64]
    L.bbduk_finalize_table.argtypes = [vp]
    L.bbduk_table_size.restype = i64
    L.bbduk_table_size.argtypes = [vp]
    L.bbduk_table_bytes.restype = i64
    L.bbduk_table_bytes.argtypes = [vp]
    L.bbduk_table_lookup.argtypes = [vp, vp, i64, vp]
    for f in (L.bbduk_ktrim_batch, L.bbduk_kfilter_batch):
        f.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    L.bbduk_kmask_batch.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
    L.bbduk_ktrimtips_batch.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
    L.bbduk_ksplit_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.bbduk_kfilter_batch_matches.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, i32, vp, vp, vp]
    L.bbduk_kfilter_batch_matches_device.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.bbduk_ksplit_batch_device.argtypes = [vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.bbduk_ktrimtips_batch_device.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp, vp]
    L.bbduk_kmask_batch_device.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp, vp]
    for f in (L.bbduk_ktrim_batch_device, L.bbduk_kfilter_batch_device):
        f.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp]
    L.bbduk_pack_bases_host.argtypes = [vp, i64, vp, vp]
    L.bbduk_pack_bases_device.argtypes = [vp, i64, vp, vp, i32, vp]
    for f in (L.bbduk_ktrim_batch_packed, L.bbduk_kfilter_batch_packed):
        f.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp]
    for f in (L.bbduk_ktrim_batch_packed_device, L.bbduk_kfilter_batch_packed_device):
        f.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp]
    L.bbduk_kmask_batch_packed_device.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp, vp]
    L.bbduk_ktrimtips_batch_packed_device.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, vp, vp]
    L.bbduk_fastq_ingest_device.argtypes = [vp, i64, vp, i64, i32, i64, i64, vp, vp, vp, vp, vp, i32, vp, C.POINTER(FastqResult)]
    L.bbduk_fastq_write_device.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, i64, i32, vp, C.POINTER(i64)]
    L.bbduk_fastq_write_masked_device.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp, i32, vp, i64, i32, vp, C.POINTER(i64)]
    L.bbduk_kernel_time_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.bbduk_counters_len.argtypes = [vp]
    L.bbduk_get_counters.argtypes = [vp, vp, i32]
    L.bbduk_reset_counters.argtypes = [vp]
    L.bbduk_synth_generate_device.argtypes = [C.POINTER(SynthParams), i64, i64, vp, vp, i32, vp]
    L.bbduk_synth_generate_host.argtypes = [C.POINTER(SynthParams), i64, i64, vp, vp]
    L.bbduk_synth_pair_inserts.argtypes = [C.POINTER(SynthParams), i64, i64, vp]
    L.bbduk_host_parse.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
    L.bbduk_host_destroy.argtypes = [vp]
    L.bbduk_host_destroy.restype = None
    L.bbduk_host_add_ref.argtypes = [vp, C.c_char_p, i64]
    L.bbduk_host_load_fasta.argtypes = [vp, C.c_char_p]
    L.bbduk_host_load_refs.argtypes = [vp, C.c_char_p]
    L.bbduk_host_build_index.restype = i64
    L.bbduk_host_build_index.argtypes = [vp]
    L.bbduk_host_index_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.bbduk_host_num_scaffolds.argtypes = [vp]
    L.bbduk_host_scaffold_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64)]
    L.bbduk_host_num_refs.argtypes = [vp]
    L.bbduk_host_ref_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i32)]
    L.bbduk_host_params.argtypes = [vp, i32, C.POINTER(Params)]
    L.bbduk_host_upload_index.argtypes = [vp, vp]
    L.bbduk_host_build_on_device.argtypes = [vp, vp]
    L.bbduk_build_table_device.argtypes = [vp, vp, vp, i32, i32, i32]
    L.bbduk_build_table_device_edits.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32]
    L.bbduk_pinned_malloc.argtypes = [i64, C.POINTER(vp)]
    L.bbduk_pinned_free.argtypes = [vp]
    L.bbduk_test_hook.argtypes = [vp, i32, i64]
    L.bbduk_table_line_histogram.argtypes = [vp, vp]
    L.bbduk_table_layout.argtypes = [vp]
    L.bbduk_table_spilled.restype = i64
    L.bbduk_table_spilled.argtypes = [vp]
    L.bbduk_build_begin.argtypes = [vp, i64, i32, i32]
    L.bbduk_build_add_device.argtypes = [vp, vp, vp, i32, i32]
    L.bbduk_build_end.argtypes = [vp]
    L.bbduk_comm_unique_id.argtypes = [vp]
    L.bbduk_comm_create.argtypes = [vp, i32, i32, vp]
    L.bbduk_comm_create_local.argtypes = [C.POINTER(vp), i32]
    L.bbduk_comm_destroy.argtypes = [vp]
    L.bbduk_comm_size.argtypes = [vp]
    L.bbduk_allreduce_counters.argtypes = [vp]
    L.bbduk_allreduce_counters_device.argtypes = [vp, vp, vp]
    L.bbduk_allreduce_counters_local.argtypes = [C.POINTER(vp), i32]
    _lib = L
    return L


class HostIndex:
    """BBDukParser + BBDukLoader/BBDukIndexMod roles (C++ host mirror).  No GPU needed."""

    def __init__(self, args: str):
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().bbduk_host_parse(args.encode(), C.byref(self.h), err, len(err))
        if rc != OK:
            raise BBDukError("bbduk_host_parse: %s" % err.value.decode())
        self.args = args
        self.built = False

    def close(self):
        if getattr(self, "h", None):
            lib().bbduk_host_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_ref(self, seq: bytes):
        if lib().bbduk_host_add_ref(self.h, seq, len(seq)) != OK:
            raise BBDukError("bbduk_host_add_ref failed")

    def load_fasta(self, path: str) -> int:
        n = lib().bbduk_host_load_fasta(self.h, path.encode())
        if n < 0:
            raise BBDukError("cannot load %s" % path)
        return n

    def load_refs(self, resource_dir: str = DATA_DIR) -> int:
        n = lib().bbduk_host_load_refs(self.h, resource_dir.encode())
        if n < 0:
            raise BBDukError("cannot load references named by ref=/literal=")
        return n

    def build_index(self) -> int:
        n = lib().bbduk_host_build_index(self.h)
        if n < 0:
            raise BBDukError("bbduk_host_build_index failed")
        self.built = True
        return n

    def pairs(self):
        pk, pv, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        if lib().bbduk_host_index_pairs(self.h, C.byref(pk), C.byref(pv), C.byref(n)) != OK:
            raise BBDukError("index not built")
        if n.value == 0:
            return np.empty(0, np.int64), np.empty(0, np.int32)
        keys = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_int64)), (n.value,)).copy()
        vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_int32)), (n.value,)).copy()
        return keys, vals

    @property
    def num_scaffolds(self) -> int:
        return lib().bbduk_host_num_scaffolds(self.h)

    def scaffold_info(self, sid: int):
        """(name, length) of scaffold sid (1-based), as BBDukLoader records them."""
        name = C.c_char_p(); ln = C.c_int64()
        if lib().bbduk_host_scaffold_info(self.h, sid, C.byref(name), C.byref(ln)) != OK:
            raise BBDukError("bbduk_host_scaffold_info failed")
        return name.value.decode(), ln.value

    def params(self, device: int = 0) -> Params:
        p = Params()
        if lib().bbduk_host_params(self.h, device, C.byref(p)) != OK:
            raise BBDukError("bbduk_host_params failed")
        return p


def pack_reads(reads):
    """list[bytes] -> (bases uint8[], offsets int64[n+1]): the batch layout of the boundary."""
    offsets = np.zeros(len(reads) + 1, np.int64)
    if reads:
        offsets[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offsets


class BBDukGpu:
    """Device handle: the k-mer map resident in HBM + the batch operators."""

    def __init__(self, params: Params):
        self.h = C.c_void_p()
        self.params = params
        rc = lib().bbduk_create(C.byref(params), C.byref(self.h))
        if rc != OK:
            raise BBDukError("bbduk_create failed rc=%d (no GPU visible, or unsupported parameters)" % rc)

    def close(self):
        if getattr(self, "h", None):
            lib().bbduk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            raise BBDukError("%s rc=%d: %s" % (what, rc, lib().bbduk_last_error(self.h).decode()))

    def test_hook(self, which: int, value: int):
        """include/bbduk_test_hooks.h: tests and experiments only."""
        self._check(lib().bbduk_test_hook(self.h, which, value), "test_hook")

    # ---- table
    def upload_pairs(self, keys: np.ndarray, values: np.ndarray):
        keys = np.ascontiguousarray(keys, np.int64); values = np.ascontiguousarray(values, np.int32)
        self._check(lib().bbduk_upload_pairs(self.h, keys.ctypes.data, values.ctypes.data, len(keys)), "upload_pairs")

    def upload_table_way(self, way, prime, keys, values, vkeys, vvals):
        keys = np.ascontiguousarray(keys, np.int64); values = np.ascontiguousarray(values, np.int32)
        vkeys = np.ascontiguousarray(vkeys, np.int64); vvals = np.ascontiguousarray(vvals, np.int32)
        self._check(lib().bbduk_upload_table_way(self.h, way, prime, keys.ctypes.data, values.ctypes.data, len(keys),
                                                 vkeys.ctypes.data if len(vkeys) else None,
                                                 vvals.ctypes.data if len(vvals) else None, len(vkeys)),
                    "upload_table_way")

    def finalize_table(self):
        self._check(lib().bbduk_finalize_table(self.h), "finalize_table")

    # streaming device-side build: whole scaffolds already in HBM, chunk by chunk
    def build_begin(self, max_keys: int, hdist: int = 0, hdist2: int = 0):
        self._check(lib().bbduk_build_begin(self.h, max_keys, hdist, hdist2), "build_begin")

    def build_add_device(self, d_refs, ref_offsets, first_id: int):
        ref_offsets = np.ascontiguousarray(ref_offsets, np.int64)
        self._check(lib().bbduk_build_add_device(self.h, d_refs.data_ptr(), ref_offsets.ctypes.data, len(ref_offsets) - 1, first_id), "build_add_device")

    def build_end(self):
        self._check(lib().bbduk_build_end(self.h), "build_end")

    @property
    def table_size(self) -> int:
        return lib().bbduk_table_size(self.h)

    @property
    def table_bytes(self) -> int:
        return lib().bbduk_table_bytes(self.h)

    @property
    def table_spilled(self) -> int:
        return lib().bbduk_table_spilled(self.h)

    @property
    def table_layout(self) -> int:
        """0 cache-resident, 1 big, 2 seed; + 4 with a cache-resident twin (include/bbduk_test_hooks.h)"""
        return lib().bbduk_table_layout(self.h)

    def line_histogram(self) -> np.ndarray:
        out = np.zeros(33, np.int64)
        self._check(lib().bbduk_table_line_histogram(self.h, out.ctypes.data), "line_histogram")
        return out

    def table_lookup(self, keys: np.ndarray) -> np.ndarray:
        keys = np.ascontiguousarray(keys, np.int64)
        out = np.empty(len(keys), np.int32)
        self._check(lib().bbduk_table_lookup(self.h, keys.ctypes.data, len(keys), out.ctypes.data), "table_lookup")
        return out

    # ---- host-buffer operators
    def _host_op(self, fn, what, bases, offsets, paired):
        bases = np.ascontiguousarray(bases, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        self._check(fn(self.h, bases.ctypes.data if len(bases) else None, offsets.ctypes.data, n, int(paired),
                       a.ctypes.data, ids.ctypes.data, fl.ctypes.data), what)
        return a, ids, fl

    def ktrim_batch(self, bases, offsets, paired):
        return self._host_op(lib().bbduk_ktrim_batch, "ktrim_batch", bases, offsets, paired)

    def kfilter_batch(self, bases, offsets, paired):
        return self._host_op(lib().bbduk_kfilter_batch, "kfilter_batch", bases, offsets, paired)

    def kmask_batch(self, bases, offsets, paired):
        """ktrim=n: (masked bases per read, ids, flags, uint32 bit mask over the concatenated bases)."""
        bases = np.ascontiguousarray(bases, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        mask = np.zeros((int(offsets[-1]) + 31) // 32 + 1, np.uint32)
        self._check(lib().bbduk_kmask_batch(self.h, bases.ctypes.data if len(bases) else None, offsets.ctypes.data, n, int(paired),
                                            a.ctypes.data, ids.ctypes.data, fl.ctypes.data, mask.ctypes.data), "kmask_batch")
        return a, ids, fl, mask

    def ktrimtips_batch(self, bases, offsets, paired):
        """ktrim=rl: (right amounts, left amounts, ids, flags)."""
        bases = np.ascontiguousarray(bases, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        xr = np.empty(n, np.int32); xl = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        self._check(lib().bbduk_ktrimtips_batch(self.h, bases.ctypes.data if len(bases) else None, offsets.ctypes.data, n, int(paired),
                                                xr.ctypes.data, xl.ctypes.data, ids.ctypes.data, fl.ctypes.data), "ktrimtips_batch")
        return xr, xl, ids, fl

    def ksplit_batch(self, bases, offsets):
        """ksplit (unpaired): (bases removed, ids, flags, leftmost, rightmost)."""
        bases = np.ascontiguousarray(bases, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        x = np.empty(n, np.int32); lm = np.empty(n, np.int32); rm = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        self._check(lib().bbduk_ksplit_batch(self.h, bases.ctypes.data if len(bases) else None, offsets.ctypes.data, n,
                                             x.ctypes.data, lm.ctypes.data, rm.ctypes.data, ids.ctypes.data, fl.ctypes.data), "ksplit_batch")
        return x, ids, fl, lm, rm

    def kfilter_batch_matches(self, bases, offsets, paired, max_ids):
        """findBestMatch + the per-read lists rename=t prints (BBDukProcessorS.java:1702, 2508-2522):
        (found, ids, flags, nids, match_ids[n,max_ids], match_counts[n,max_ids]); entries past a read's list are 0."""
        bases = np.ascontiguousarray(bases, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        nids = np.zeros(n, np.int32); mi = np.zeros((n, max_ids), np.int32); mc = np.zeros((n, max_ids), np.int32)
        self._check(lib().bbduk_kfilter_batch_matches(self.h, bases.ctypes.data if len(bases) else None, offsets.ctypes.data, n, int(paired),
                                                      a.ctypes.data, ids.ctypes.data, fl.ctypes.data, max_ids, nids.ctypes.data,
                                                      mi.ctypes.data, mc.ctypes.data), "kfilter_batch_matches")
        return a, ids, fl, nids, mi, mc

    def process_batch(self, bases, offsets, paired):
        if self.params.mode == MODE_KSPLIT:
            if paired:
                raise BBDukError("ksplit works on unpaired reads (BBDukProcessorS.java:2334)")
            return self.ksplit_batch(bases, offsets)[:3]
        if self.params.mode == MODE_KTRIM_TIPS:
            xr, xl, ids, fl = self.ktrimtips_batch(bases, offsets, paired)
            return xr + xl, ids, fl
        if self.params.mode == MODE_KMASK:
            return self.kmask_batch(bases, offsets, paired)[:3]
        if self.params.mode == MODE_KFILTER:
            return self.kfilter_batch(bases, offsets, paired)
        return self.ktrim_batch(bases, offsets, paired)

    # ---- packed boundary format (2-bit codes + undefined bits; offsets still count bases)
    def process_batch_packed(self, codes, undef, offsets, paired):
        codes = np.ascontiguousarray(codes, np.uint32); undef = np.ascontiguousarray(undef, np.uint32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.empty(n, np.uint8)
        fn = lib().bbduk_kfilter_batch_packed if self.params.mode == MODE_KFILTER else lib().bbduk_ktrim_batch_packed
        self._check(fn(self.h, codes.ctypes.data if len(codes) else None, undef.ctypes.data if len(undef) else None,
                       offsets.ctypes.data, n, int(paired), a.ctypes.data, ids.ctypes.data, fl.ctypes.data), "batch_packed")
        return a, ids, fl

    def process_batch_packed_device(self, d_codes, d_undef, d_offsets, total_bases, paired, d_a, d_id, d_fl, d_counters, stream_ptr=0):
        n = d_offsets.numel() - 1
        fn = lib().bbduk_kfilter_batch_packed_device if self.params.mode == MODE_KFILTER else lib().bbduk_ktrim_batch_packed_device
        self._check(fn(self.h, d_codes.data_ptr(), d_undef.data_ptr(), d_offsets.data_ptr(), n, int(total_bases), int(paired),
                       d_a.data_ptr(), d_id.data_ptr(), d_fl.data_ptr(), d_counters.data_ptr(), stream_ptr),
                    "batch_packed_device")

    # ---- device-buffer operators (torch tensors are only carriers of HBM pointers)
    def process_batch_device(self, d_bases, d_offsets, paired, d_a, d_id, d_fl, d_counters, stream_ptr=0):
        n = d_offsets.numel() - 1
        fn = lib().bbduk_kfilter_batch_device if self.params.mode == MODE_KFILTER else lib().bbduk_ktrim_batch_device
        self._check(fn(self.h, d_bases.data_ptr(), d_offsets.data_ptr(), n, d_bases.numel(), int(paired),
                       d_a.data_ptr(), d_id.data_ptr(), d_fl.data_ptr(), d_counters.data_ptr(), stream_ptr),
                    "batch_device")

    def kernel_time_ms(self, last_k: int) -> float:
        ms = C.c_float()
        self._check(lib().bbduk_kernel_time_ms(self.h, last_k, C.byref(ms)), "kernel_time_ms")
        return ms.value

    # ---- counters
    @property
    def counters_len(self) -> int:
        return lib().bbduk_counters_len(self.h)

    def counters(self) -> np.ndarray:
        out = np.zeros(self.counters_len, np.int64)
        self._check(lib().bbduk_get_counters(self.h, out.ctypes.data, len(out)), "get_counters")
        return out

    def reset_counters(self):
        self._check(lib().bbduk_reset_counters(self.h), "reset_counters")

    # ---- multi-GPU: the counter all-reduce (RCCL) behind the C ABI
    def comm_create(self, nranks: int, rank: int, unique_id: bytes):
        """One process per GPU: join the communicator rank 0 opened with comm_unique_id()."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(lib().bbduk_comm_create(self.h, nranks, rank, buf), "comm_create")

    def allreduce_counters(self):
        """Sum of the handle's own counter vector over the communicator (blocking)."""
        self._check(lib().bbduk_allreduce_counters(self.h), "allreduce_counters")

    def allreduce_counters_device(self, d_counters, stream_ptr=0):
        self._check(lib().bbduk_allreduce_counters_device(self.h, d_counters.data_ptr(), stream_ptr), "allreduce_counters_device")

    @property
    def comm_size(self) -> int:
        return lib().bbduk_comm_size(self.h)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI: 128 bytes rank 0 hands to every rank."""
    buf = (C.c_uint8 * 128)()
    if lib().bbduk_comm_unique_id(buf) != OK:
        raise BBDukError("bbduk_comm_unique_id failed (librccl not loadable?)")
    return bytes(buf)


def comm_create_local(gpus):
    """One process, several handles (on one or several devices): form their local group."""
    arr = (C.c_void_p * len(gpus))(*[g.h for g in gpus])
    rc = lib().bbduk_comm_create_local(arr, len(gpus))
    if rc != OK:
        raise BBDukError("comm_create_local rc=%d: %s" % (rc, lib().bbduk_last_error(gpus[0].h).decode()))


def allreduce_counters_local(gpus):
    arr = (C.c_void_p * len(gpus))(*[g.h for g in gpus])
    rc = lib().bbduk_allreduce_counters_local(arr, len(gpus))
    if rc != OK:
        raise BBDukError("allreduce_counters_local rc=%d: %s" % (rc, lib().bbduk_last_error(gpus[0].h).decode()))


class BBDuk:
    """`bbduk.sh <args>` minus the file streaming: parse, load refs, build the index, hold it on the GPU."""

    def __init__(self, args: str, device: int = 0, resource_dir: str = DATA_DIR, refs=None, build: str = "host", hooks=None):
        """build="host": the C++ mirror of the Java index build + upload (what a JNI caller does with its own tables);
        build="device": the reference sequences go to the GPU and the map is built there (bbduk_build_table_device)."""
        self.host = HostIndex(args)
        if refs is not None:
            for r in refs:
                self.host.add_ref(r)
        else:
            self.host.load_refs(resource_dir)
        if build == "device":
            self.gpu = BBDukGpu(self.host.params(device))
            for k_, v_ in (hooks or {}).items():
                self.gpu.test_hook(k_, v_)
            self.gpu._check(lib().bbduk_host_build_on_device(self.host.h, self.gpu.h), "build_on_device")
            self.stored_kmers = self.gpu.table_size
        else:
            self.stored_kmers = self.host.build_index()
            self.gpu = BBDukGpu(self.host.params(device))
            for k_, v_ in (hooks or {}).items():
                self.gpu.test_hook(k_, v_)
            rc = lib().bbduk_host_upload_index(self.host.h, self.gpu.h)
            self.gpu._check(rc, "upload_index")

    def process_reads(self, reads, paired: bool):
        bases, offsets = pack_reads(reads)
        return self.gpu.process_batch(bases, offsets, paired)

    def counters(self):
        return self.gpu.counters()

    def close(self):
        self.gpu.close(); self.host.close()


# ---- synthetic workload
TRUSEQ_R1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTG"      # data/adapters.fa >Reverse_adapter
TRUSEQ_R2 = b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTCGCCGTATCATT"            # revcomp of >TruSeq_Universal_Adapter


def synth_params(seed: int, read_len=150, ins_min=50, ins_max=350, adapter1=TRUSEQ_R1, adapter2=TRUSEQ_R2,
                 sub_rate=0.005, n_rate=0.001, contam: bytes = b"", contam_frac=0.0) -> SynthParams:
    sp = SynthParams()
    sp.seed = seed; sp.read_len = read_len; sp.ins_min = ins_min; sp.ins_max = ins_max
    sp.adapter1 = adapter1; sp.adapter1_len = len(adapter1)
    sp.adapter2 = adapter2; sp.adapter2_len = len(adapter2)
    q = lambda x: min(0xFFFFFFFF, int(round(x * 4294967296.0)))
    sp.sub_rate_q32 = q(sub_rate); sp.n_rate_q32 = q(n_rate); sp.contam_frac_q32 = q(contam_frac)
    sp.contam = contam if contam else None; sp.contam_len = len(contam)
    sp._keep = (adapter1, adapter2, contam)
    return sp


def synth_generate_host(sp: SynthParams, first_pair: int, n_pairs: int):
    bases = np.empty(n_pairs * 2 * sp.read_len, np.uint8)
    offsets = np.empty(2 * n_pairs + 1, np.int64)
    rc = lib().bbduk_synth_generate_host(C.byref(sp), first_pair, n_pairs, bases.ctypes.data, offsets.ctypes.data)
    if rc != OK:
        raise BBDukError("synth_generate_host rc=%d" % rc)
    return bases, offsets


def synth_pair_inserts(sp: SynthParams, first_pair: int, n_pairs: int) -> np.ndarray:
    """Insert size of every pair: the generator's truth (a read keeps min(read_len, insert) genome bases)."""
    out = np.empty(n_pairs, np.int32)
    if lib().bbduk_synth_pair_inserts(C.byref(sp), first_pair, n_pairs, out.ctypes.data) != OK:
        raise BBDukError("synth_pair_inserts failed")
    return out


def synth_generate_device(sp: SynthParams, first_pair: int, n_pairs: int, d_bases, d_offsets, device: int, stream_ptr=0):
    rc = lib().bbduk_synth_generate_device(C.byref(sp), first_pair, n_pairs, d_bases.data_ptr(), d_offsets.data_ptr(),
                                           device, stream_ptr)
    if rc != OK:
        raise BBDukError("synth_generate_device rc=%d" % rc)


def read_fasta(path: str):
    """[(name, seq bytes)] -- small helper for contaminant sequences (phiX) used by the generator."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    out, name, seq = [], None, bytearray()
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, bytes(seq)))
                name, seq = line[1:].decode(), bytearray()
            else:
                seq += line
    if name is not None:
        out.append((name, bytes(seq)))
    return out
