"""Builds libbbduk_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "csrc", "bbduk_hip.hip"), os.path.join(_HERE, "csrc", "bbduk_host.cpp")]
_DEPS = _SRC + [os.path.join(_HERE, "csrc", "synth.h"),
                os.path.join(_HERE, "..", "include", "bbduk_gpu.h"), os.path.join(_HERE, "..", "include", "bbduk_host.h")]
_SO = os.path.join(_HERE, "libbbduk_hip.so")


def lib_path() -> str:
    return _SO


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.getmtime(d) > t for d in _DEPS)


def _compile(out: str, extra=(), verbose: bool = False) -> str:
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result", *extra, *_SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return _SO
    return _compile(_SO, verbose=verbose)


def build_timing_variant(verbose: bool = False) -> str:
    """Experiment build with the BBDUK_DBG stage-deletion switches compiled in (profiles/ab.sh loads it through
    BBDUK_LIB_PATH); never the product library."""
    return _compile(os.path.join(_HERE, "ab_tsw.so"), ["-DBBDUK_TIMING_SWITCHES"], verbose)
