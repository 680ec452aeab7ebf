"""
ctypes binding for oracle/libbbduk_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from bbtools_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbbduk_oracle.so")


class BboArgs(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("k", "mink", "hdist", "hdist2", "edist", "edist2", "qhdist", "qhdist2", "maskMiddle", "midMaskLen",
                 "rcomp", "forbidN", "ktrimRight", "ktrimLeft", "maxBadKmers0", "minReadLength")] + \
               [("minLenFraction", C.c_float)] + \
               [(n, C.c_int) for n in ("requireBothBad", "trimPad", "ktrimExclusive", "restrictLeft", "restrictRight",
                                       "skipR1", "skipR2", "minSkip", "maxSkip", "trimPairsEvenly", "qSkip", "speed")] + \
               [("minKmerFraction", C.c_float), ("minCoveredFraction", C.c_float), ("ktrimN", C.c_int),
                ("kbig", C.c_int), ("findBestMatch", C.c_int), ("ksplit", C.c_int), ("kmaskFullyCovered", C.c_int), ("trimFailuresTo1bp", C.c_int)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bbduk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.bbo_create.restype = C.c_void_p
        L.bbo_create.argtypes = [C.POINTER(BboArgs)]
        L.bbo_destroy.argtypes = [C.c_void_p]
        L.bbo_constant.restype = C.c_int64
        L.bbo_constant.argtypes = [C.c_void_p, C.c_char_p]
        L.bbo_rcomp.restype = C.c_int64
        L.bbo_rcomp.argtypes = [C.c_int64, C.c_int]
        L.bbo_add_ref_sequence.restype = C.c_int64
        L.bbo_add_ref_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.bbo_load_fasta.argtypes = [C.c_void_p, C.c_char_p]
        L.bbo_num_scaffolds.argtypes = [C.c_void_p]
        L.bbo_stored_kmers.restype = C.c_int64
        L.bbo_stored_kmers.argtypes = [C.c_void_p]
        L.bbo_table_get.argtypes = [C.c_void_p, C.c_int64]
        L.bbo_test_set_probe_window.argtypes = [C.c_int]
        L.bbo_test_set_probe_window.restype = None
        L.bbo_num_ways.argtypes = [C.c_void_p]
        L.bbo_way_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bbo_dump_pairs.restype = C.c_int64
        L.bbo_dump_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.bbo_get_value.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]
        L.bbo_ktrim_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.bbo_count_set_kmers.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.bbo_process_batch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.bbo_process_batch_split.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.bbo_process_batch_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.bbo_process_batch_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.bbo_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.bbo_counters_len.argtypes = [C.c_void_p]
        L.bbo_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.bbo_reset_counters.argtypes = [C.c_void_p]
        L.bbo_default_args.argtypes = [C.POINTER(BboArgs)]
        _lib = L
    return _lib


def make_args(**kw) -> BboArgs:
    a = BboArgs()
    lib().bbo_default_args(C.byref(a))
    for k, v in kw.items():
        if not hasattr(a, k):
            raise KeyError(k)
        setattr(a, k, v)
    return a


def set_probe_window(n: int = 60):
    """Tests only: the HashArray1D probe window of oracles created from now on (reference: 60, kmer/HashArray.java:687)."""
    lib().bbo_test_set_probe_window(n)


class Oracle:
    """Thin object wrapper; method names follow the reference's (ktrim, countSetKmers, processList stage)."""

    def __init__(self, **kw):
        self.args = make_args(**kw)
        self.h = lib().bbo_create(C.byref(self.args))
        if not self.h:
            raise ValueError("bbo_create rejected the arguments")

    def close(self):
        if self.h:
            lib().bbo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def constant(self, name: str) -> int:
        return lib().bbo_constant(self.h, name.encode())

    def add_ref(self, seq: bytes) -> int:
        return lib().bbo_add_ref_sequence(self.h, seq, len(seq))

    def load_fasta(self, path: str) -> int:
        if path.endswith(".gz"):
            import gzip, tempfile
            with gzip.open(path, "rb") as f, tempfile.NamedTemporaryFile(suffix=".fa", delete=False) as t:
                t.write(f.read()); tmp = t.name
            try:
                return lib().bbo_load_fasta(self.h, tmp.encode())
            finally:
                os.unlink(tmp)
        return lib().bbo_load_fasta(self.h, path.encode())

    @property
    def num_scaffolds(self) -> int:
        return lib().bbo_num_scaffolds(self.h)

    @property
    def stored_kmers(self) -> int:
        return lib().bbo_stored_kmers(self.h)

    def table_get(self, key: int) -> int:
        return lib().bbo_table_get(self.h, key)

    def dump_pairs(self):
        n = lib().bbo_dump_pairs(self.h, None, None, 0)
        keys = np.empty(n, dtype=np.int64); vals = np.empty(n, dtype=np.int32)
        lib().bbo_dump_pairs(self.h, keys.ctypes.data, vals.ctypes.data, n)
        return keys, vals

    def way_images(self):
        """[(prime, keys[int64 ncells], values[int32 ncells], vkeys, vvals)] -- HashArray1D images, one per way."""
        out = []
        for w in range(lib().bbo_num_ways(self.h)):
            prime = C.c_int(); ncells = C.c_int64(); nv = C.c_int64()
            pk = C.c_void_p(); pv = C.c_void_p(); pvk = C.c_void_p(); pvv = C.c_void_p()
            lib().bbo_way_image(self.h, w, C.byref(prime), C.byref(ncells), C.byref(pk), C.byref(pv),
                                C.byref(nv), C.byref(pvk), C.byref(pvv))
            keys = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_int64)), (ncells.value,)).copy()
            vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_int32)), (ncells.value,)).copy()
            if nv.value:
                vk = np.ctypeslib.as_array(C.cast(pvk, C.POINTER(C.c_int64)), (nv.value,)).copy()
                vv = np.ctypeslib.as_array(C.cast(pvv, C.POINTER(C.c_int32)), (nv.value,)).copy()
            else:
                vk = np.empty(0, np.int64); vv = np.empty(0, np.int32)
            out.append((prime.value, keys, vals, vk, vv))
        return out

    def get_value(self, kmer, rkmer, length_mask, qpos, length, qhdist) -> int:
        return lib().bbo_get_value(self.h, kmer, rkmer, length_mask, qpos, length, qhdist)

    def ktrim(self, read: bytes, pairnum: int = 0):
        id0 = C.c_int(-1)
        x = lib().bbo_ktrim_read(self.h, read, len(read), pairnum, C.byref(id0))
        return x, id0.value

    def count_set_kmers(self, read: bytes, pairnum: int = 0, max_bad: int = 0):
        i = C.c_int(-1)
        f = lib().bbo_count_set_kmers(self.h, read, len(read), pairnum, max_bad, C.byref(i))
        return f, i.value

    def process_batch(self, bases: np.ndarray, offsets: np.ndarray, paired: bool, nthreads: int = 1):
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.zeros(n, np.uint8)
        rc = lib().bbo_process_batch(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(paired),
                                     a.ctypes.data, ids.ctypes.data, fl.ctypes.data, nthreads)
        if rc != 0:
            raise ValueError("bbo_process_batch rc=%d" % rc)
        return a, ids, fl

    def process_batch_mask(self, bases: np.ndarray, offsets: np.ndarray, paired: bool, nthreads: int = 1):
        """ktrim=n: (masked-count per read, ids, flags, uint32 bit mask over the concatenated bases)."""
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.zeros(n, np.uint8)
        mask = np.zeros((int(offsets[-1]) + 31) // 32 + 1, np.uint32)
        rc = lib().bbo_process_batch_mask(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(paired),
                                          a.ctypes.data, ids.ctypes.data, fl.ctypes.data, mask.ctypes.data, nthreads)
        if rc != 0:
            raise ValueError("bbo_process_batch_mask rc=%d" % rc)
        return a, ids, fl, mask

    def process_batch_tips(self, bases: np.ndarray, offsets: np.ndarray, paired: bool, nthreads: int = 1):
        """ktrim=rl: (right amounts, left amounts, ids, flags)."""
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.zeros(n, np.uint8); left = np.zeros(n, np.int32)
        rc = lib().bbo_process_batch_ex(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(paired),
                                        a.ctypes.data, ids.ctypes.data, fl.ctypes.data, None, left.ctypes.data, nthreads)
        if rc != 0:
            raise ValueError("bbo_process_batch_ex rc=%d" % rc)
        return a - left, left, ids, fl

    def process_batch_split(self, bases: np.ndarray, offsets: np.ndarray, nthreads: int = 1):
        """ksplit (unpaired): (bases removed, ids, flags, leftmost, rightmost)."""
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        a = np.empty(n, np.int32); ids = np.empty(n, np.int32); fl = np.zeros(n, np.uint8)
        lm = np.empty(n, np.int32); rm = np.empty(n, np.int32)
        rc = lib().bbo_process_batch_split(self.h, bases.ctypes.data, offsets.ctypes.data, n,
                                           a.ctypes.data, ids.ctypes.data, fl.ctypes.data, lm.ctypes.data, rm.ctypes.data, nthreads)
        if rc != 0:
            raise ValueError("bbo_process_batch_split rc=%d" % rc)
        return a, ids, fl, lm, rm

    def process_batch_matches(self, bases: np.ndarray, offsets: np.ndarray, paired: bool, max_ids: int, nthreads: int = 1):
        """findBestMatch + rename's lists: (found, ids, flags, nids, match_ids[n,max_ids], match_counts[n,max_ids]); unused entries 0."""
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        a = np.zeros(n, np.int32); ids = np.zeros(n, np.int32); fl = np.zeros(n, np.uint8)
        nids = np.zeros(n, np.int32); mi = np.zeros((n, max_ids), np.int32); mc = np.zeros((n, max_ids), np.int32)
        rc = lib().bbo_process_batch_matches(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(paired), a.ctypes.data, ids.ctypes.data,
                                             fl.ctypes.data, max_ids, nids.ctypes.data, mi.ctypes.data, mc.ctypes.data, nthreads)
        if rc != 0:
            raise ValueError("bbo_process_batch_matches rc=%d" % rc)
        return a, ids, fl, nids, mi, mc

    def counters(self) -> np.ndarray:
        out = np.zeros(lib().bbo_counters_len(self.h), np.int64)
        lib().bbo_get_counters(self.h, out.ctypes.data)
        return out

    def reset_counters(self):
        lib().bbo_reset_counters(self.h)


def pack_reads(reads):
    """list[bytes] -> (bases uint8[], offsets int64[n+1]) : the boundary's batch layout."""
    offsets = np.zeros(len(reads) + 1, np.int64)
    if reads:
        offsets[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offsets
