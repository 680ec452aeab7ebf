"""ctypes mirror of include/seal_gpu.h: jgi/Seal.java's k-mer path on the MI355X (the next tool on bbduk's encode / lookup core).
The operator interface follows the reference's command line: Seal("k=31 ambig=toss mkh=2 ...").  No CPU fallback: every call goes to
libbbduk_hip.so."""
import ctypes as C
import numpy as np

from .bbduk import BBDukError, lib as _bbduk_lib, pack_reads

OK = 0
MATCH_ALL, MATCH_FIRST, MATCH_UNIQUE = 0, 1, 2
AMBIG_FIRST, AMBIG_ALL, AMBIG_RANDOM, AMBIG_TOSS = 0, 1, 2, 3
FLAG_REMOVED, FLAG_MATCHED = 1, 2
NCOUNTERS = 16
COUNTER_NAMES = ["readsIn", "basesIn", "fragsIn", "readsMatched", "basesMatched", "readsUnmatched", "basesUnmatched",
                 "readsQFiltered", "basesQFiltered", "readsQTrimmed"]
SYMBOLS = ["seal_default_params", "seal_params_from_args", "seal_create", "seal_destroy", "seal_last_error", "seal_add_ref_sequence",
           "seal_upload_pairs", "seal_finalize", "seal_num_scaffolds", "seal_table_keys", "seal_table_pairs", "seal_batch_device",
           "seal_batch", "seal_counters_len", "seal_read_counters", "seal_reset_counters", "seal_last_kernel_ms", "seal_comm_create", "seal_allreduce_counters"]


class SealParams(C.Structure):     # struct seal_params
    _fields_ = [(n, C.c_int32) for n in ("k", "maskMiddle", "midMaskLen", "rcomp", "forbidNs", "hdist", "refSkip", "restrictLeft",
                                         "restrictRight", "qSkip", "speed", "matchMode", "ambigMode", "keepPairsTogether", "minKmerHits")] + \
               [("minKmerFraction", C.c_float), ("clearzone", C.c_int32), ("clearzoneFraction", C.c_float), ("minReadLength", C.c_int32), ("maxReadLength", C.c_int32),
                ("minLenFraction", C.c_float), ("requireBothBad", C.c_int32), ("maxScaffolds", C.c_int32), ("device", C.c_int32)]


_bound = False


def lib():
    global _bound
    L = _bbduk_lib()
    if not _bound:
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.seal_default_params.argtypes = [C.POINTER(SealParams)]
        L.seal_params_from_args.argtypes = [C.c_char_p, C.POINTER(SealParams), C.c_char_p, C.c_int]
        L.seal_create.argtypes = [C.POINTER(SealParams), C.POINTER(vp)]
        L.seal_destroy.argtypes = [vp]
        L.seal_last_error.restype = C.c_char_p
        L.seal_last_error.argtypes = [vp]
        L.seal_add_ref_sequence.argtypes = [vp, vp, i64, C.POINTER(i32)]
        L.seal_upload_pairs.argtypes = [vp, vp, vp, i64]
        L.seal_finalize.argtypes = [vp]
        L.seal_num_scaffolds.argtypes = [vp]
        L.seal_table_keys.restype = i64; L.seal_table_keys.argtypes = [vp]
        L.seal_table_pairs.restype = i64; L.seal_table_pairs.argtypes = [vp]
        L.seal_batch_device.argtypes = [vp, vp, vp, i64, i64, i32, i64, i32, vp, vp, vp, vp, vp, vp, vp]
        L.seal_batch.argtypes = [vp, vp, vp, i64, i32, i64, i32, vp, vp, vp, vp, vp]
        L.seal_counters_len.restype = i64; L.seal_counters_len.argtypes = [vp]
        L.seal_read_counters.argtypes = [vp, vp]
        L.seal_reset_counters.argtypes = [vp]
        L.seal_last_kernel_ms.restype = C.c_double; L.seal_last_kernel_ms.argtypes = [vp]
        L.seal_comm_create.argtypes = [vp, i32, i32, C.c_char_p]
        L.seal_allreduce_counters.argtypes = [vp]
        L.seal_test_hook.argtypes = [vp, i32, i64]
        _bound = True
    return L


def parse_args(args: str, device: int = 0, max_scaffolds: int = 1 << 16) -> SealParams:
    p = SealParams()
    err = C.create_string_buffer(256)
    if lib().seal_params_from_args(args.encode(), C.byref(p), err, 256) != OK:
        raise BBDukError("seal: " + err.value.decode())
    p.device = device; p.maxScaffolds = max_scaffolds
    return p


class Seal:
    """One table + the batch operator.  refs: list[bytes] (scaffold i gets id i+1), or pairs=(keys, ids) from a table built elsewhere."""

    def __init__(self, args: str, refs=None, pairs=None, device: int = 0, max_scaffolds: int = 1 << 16, hooks=None):
        self.p = parse_args(args, device, max_scaffolds)
        self.h = C.c_void_p()
        if lib().seal_create(C.byref(self.p), C.byref(self.h)) != OK:
            raise BBDukError("seal_create failed (parameter out of range, or no device)")
        for r in (refs or []):
            buf = np.frombuffer(r, dtype=np.uint8)
            sid = C.c_int32()
            self._check(lib().seal_add_ref_sequence(self.h, buf.ctypes.data, len(r), C.byref(sid)), "seal_add_ref_sequence")
        if pairs is not None:
            keys = np.ascontiguousarray(pairs[0], np.int64); ids = np.ascontiguousarray(pairs[1], np.int32)
            self._check(lib().seal_upload_pairs(self.h, keys.ctypes.data, ids.ctypes.data, len(keys)), "seal_upload_pairs")
        for k_, v_ in (hooks or {}).items():         # include/bbduk_test_hooks.h (tests / experiments): e.g. {HOOK_BUCKET_BITS: 21}
            self._check(lib().seal_test_hook(self.h, int(k_), int(v_)), "seal_test_hook")
        self._check(lib().seal_finalize(self.h), "seal_finalize")

    def _check(self, rc, what):
        if rc != OK:
            raise BBDukError("%s: rc=%d %s" % (what, rc, lib().seal_last_error(self.h).decode()))

    def close(self):
        if self.h:
            lib().seal_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_scaffolds(self):
        return lib().seal_num_scaffolds(self.h)

    @property
    def table_keys(self):
        return lib().seal_table_keys(self.h)

    @property
    def table_pairs(self):
        return lib().seal_table_pairs(self.h)

    def process_reads(self, reads, paired: bool, first_numeric_id: int = 0, max_ids: int = 8):
        """list[bytes] (paired: mates interleaved) -> sites, assigned, max, ids[n, max_ids], flags (all per read)."""
        bases, offsets = pack_reads(reads)
        n = len(reads)
        sites = np.zeros(n, np.int32); assigned = np.zeros(n, np.int32); mx = np.zeros(n, np.int32)
        ids = np.zeros((n, max_ids), np.int32); flags = np.zeros(n, np.uint8)
        self._check(lib().seal_batch(self.h, bases.ctypes.data, offsets.ctypes.data, n, 1 if paired else 0, first_numeric_id, max_ids,
                                     sites.ctypes.data, assigned.ctypes.data, mx.ctypes.data, ids.ctypes.data, flags.ctypes.data), "seal_batch")
        return sites, assigned, mx, ids, flags

    def process_device(self, d_bases, d_offsets, n, total_bases, paired, first_numeric_id, max_ids, d_sites, d_assigned, d_max, d_ids, d_flags,
                       d_counters, stream_ptr=0):
        self._check(lib().seal_batch_device(self.h, d_bases.data_ptr(), d_offsets.data_ptr(), n, total_bases, 1 if paired else 0, first_numeric_id,
                                            max_ids, d_sites.data_ptr(), d_assigned.data_ptr(), d_max.data_ptr(), d_ids.data_ptr() if max_ids else 0,
                                            d_flags.data_ptr(), d_counters.data_ptr(), stream_ptr), "seal_batch_device")

    def counters_len(self):
        return lib().seal_counters_len(self.h)

    def counters(self):
        out = np.zeros(self.counters_len(), np.int64)
        self._check(lib().seal_read_counters(self.h, out.ctypes.data), "seal_read_counters")
        S = self.p.maxScaffolds
        c = {n: int(out[i]) for i, n in enumerate(COUNTER_NAMES)}
        return c, out[NCOUNTERS:NCOUNTERS + S], out[NCOUNTERS + S:NCOUNTERS + 2 * S], out[NCOUNTERS + 2 * S:NCOUNTERS + 3 * S], out[NCOUNTERS + 3 * S:NCOUNTERS + 4 * S]

    def reset_counters(self):
        self._check(lib().seal_reset_counters(self.h), "seal_reset_counters")

    def kernel_ms(self):
        return lib().seal_last_kernel_ms(self.h)

    def comm_create(self, nranks: int, rank: int, unique_id: bytes):
        self._check(lib().seal_comm_create(self.h, nranks, rank, unique_id), "seal_comm_create")

    def allreduce_counters(self):
        self._check(lib().seal_allreduce_counters(self.h), "seal_allreduce_counters")
