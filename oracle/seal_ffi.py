"""
ctypes binding for oracle/libseal_oracle.so (the C restatement of jgi/Seal.java's k-mer path).  TEST INFRASTRUCTURE ONLY:
importable from tests/, never from bbtools_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libseal_oracle.so")
COUNTER_NAMES = ["readsIn", "basesIn", "fragsIn", "readsMatched", "basesMatched", "readsUnmatched", "basesUnmatched",
                 "readsQFiltered", "basesQFiltered", "readsQTrimmed"]


class SoArgs(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("k", "maskMiddle", "midMaskLen", "rcomp", "forbidN", "hdist", "refSkip", "restrictLeft", "restrictRight",
                                       "qSkip", "speed", "matchMode", "ambigMode", "keepPairsTogether", "minKmerHits")] + \
               [("minKmerFraction", C.c_float), ("clearzone", C.c_int), ("clearzoneFraction", C.c_float), ("minReadLength", C.c_int),
                ("maxReadLength", C.c_int), ("minLenFraction", C.c_float), ("requireBothBad", C.c_int)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "seal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libseal_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.so_default_args.argtypes = [C.POINTER(SoArgs)]
        L.so_create.restype = C.c_void_p; L.so_create.argtypes = [C.POINTER(SoArgs)]
        L.so_destroy.argtypes = [C.c_void_p]
        L.so_add_ref_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.so_finalize.argtypes = [C.c_void_p]
        L.so_num_scaffolds.argtypes = [C.c_void_p]
        L.so_num_pairs.restype = C.c_int64; L.so_num_pairs.argtypes = [C.c_void_p]
        L.so_dump_pairs.restype = C.c_int64; L.so_dump_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.so_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.so_scaffold_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.so_reset_counters.argtypes = [C.c_void_p]
        L.so_process.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class SealOracle:
    """args: dict of so_args fields over the reference's defaults."""

    def __init__(self, refs, **kw):
        a = SoArgs(); lib().so_default_args(C.byref(a))
        for k, v in kw.items():
            if not hasattr(a, k):
                raise KeyError(k)
            setattr(a, k, v)
        self.a = a
        self.c = lib().so_create(C.byref(a))
        if not self.c:
            raise ValueError("so_create refused the arguments")
        for r in refs:
            lib().so_add_ref_sequence(self.c, r, len(r))
        lib().so_finalize(self.c)

    def __del__(self):
        try:
            if self.c:
                lib().so_destroy(self.c); self.c = None
        except Exception:
            pass

    @property
    def num_scaffolds(self):
        return lib().so_num_scaffolds(self.c)

    def pairs(self):
        n = lib().so_num_pairs(self.c)
        keys = np.zeros(n, np.int64); ids = np.zeros(n, np.int32)
        lib().so_dump_pairs(self.c, keys.ctypes.data, ids.ctypes.data, n)
        return keys, ids

    def process(self, r1, r2, numeric_id, cap=64):
        """-> (ids list, info = sites, assigned, max1, max2, removed, assigned1, assigned2, sites1, sites2)"""
        out = np.zeros(cap, np.int32); info = np.zeros(9, np.int32)
        n = lib().so_process(self.c, r1, len(r1), r2, len(r2) if r2 is not None else 0, numeric_id, out.ctypes.data, cap, info.ctypes.data)
        return [int(x) for x in out[:min(n, cap)]], info

    def process_reads(self, reads, paired, first_numeric_id=0, max_ids=8):
        """The same per-read layout as bbtools_amd.seal.Seal.process_reads."""
        n = len(reads)
        sites = np.zeros(n, np.int32); assigned = np.zeros(n, np.int32); mx = np.zeros(n, np.int32)
        ids = np.zeros((n, max_ids), np.int32); flags = np.zeros(n, np.uint8)
        kpt = bool(self.a.keepPairsTogether)
        step = 2 if paired else 1
        for u in range(0, n, step):
            r1 = reads[u]; r2 = reads[u + 1] if paired else None
            got, info = self.process(r1, r2, first_numeric_id + u // step)
            fl = (1 if info[4] else 0) | (2 if (not info[4] and info[1] >= 1) else 0)
            if not paired or kpt:
                for q in range(step):
                    sites[u + q] = info[0]; assigned[u + q] = info[1]; mx[u + q] = info[2]; flags[u + q] = fl
                    ids[u + q, :min(len(got), max_ids)] = got[:max_ids]
            else:
                # keepPairsTogether=f: the restatement reports read 1's scaffolds, then read 2's
                a1, a2 = int(info[5]), int(info[6])
                for q, g in enumerate((got[:a1], got[a1:a1 + a2])):
                    sites[u + q] = info[7 + q]; assigned[u + q] = len(g); mx[u + q] = info[2 + q]; flags[u + q] = fl
                    ids[u + q, :min(len(g), max_ids)] = g[:max_ids]
        return sites, assigned, mx, ids, flags

    def counters(self):
        out = np.zeros(len(COUNTER_NAMES), np.int64)
        lib().so_counters(self.c, out.ctypes.data)
        S = self.num_scaffolds
        arrs = []
        for w in range(4):
            a = np.zeros(S, np.int64); lib().so_scaffold_counts(self.c, w, a.ctypes.data); arrs.append(a)
        return {n: int(out[i]) for i, n in enumerate(COUNTER_NAMES)}, arrs[0], arrs[1], arrs[2], arrs[3]

    def reset_counters(self):
        lib().so_reset_counters(self.c)
