"""Builds libbbduk_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# bbduk_hip.hip = host code + the secondary kernels; bbduk_k_*.hip / bbduk_stream.hip = one kernel family each (they compile in parallel)
_SRC = [os.path.join(_HERE, "csrc", f) for f in ("bbduk_hip.hip", "bbduk_k_ktrimr.hip", "bbduk_k_ktriml.hip", "bbduk_k_kfilter.hip",
                                                  "bbduk_k_modes_a.hip", "bbduk_k_modes_b.hip", "bbduk_stream.hip", "bbduk_bigs.hip", "bbduk_bigs_every.hip", "bbduk_bigs_every_b.hip", "bbduk_big_tiles.hip", "bbduk_bigs_general.hip", "bbduk_bigs_general_b.hip", "bbduk_bigs_kbig.hip", "bbduk_stream_every.hip", "bbduk_stream_every_b.hip",
                                                  "bbduk_ingest.hip", "bbduk_comm.hip", "bbduk_host.cpp")]
_DEPS = _SRC + [os.path.join(_HERE, "csrc", "synth.h"), os.path.join(_HERE, "csrc", "bbduk_internal.h"),
                os.path.join(_HERE, "csrc", "bbduk_device.inc"), os.path.join(_HERE, "csrc", "bbduk_kernels.h"), os.path.join(_HERE, "csrc", "bbduk_stream_scan.inc"), os.path.join(_HERE, "csrc", "bbduk_seed.inc"), os.path.join(_HERE, "csrc", "bbduk_bigs.inc"),
                os.path.join(_HERE, "csrc", "bbduk_seal.inc"), os.path.join(_HERE, "..", "include", "seal_gpu.h"),
                os.path.join(_HERE, "..", "include", "bbduk_gpu.h"), os.path.join(_HERE, "..", "include", "bbduk_host.h")]
_SO = os.path.join(_HERE, "libbbduk_hip.so")
_CLI_SRC = os.path.join(_HERE, "csrc", "bbduk_cli.cpp")
_CLI = os.path.join(_HERE, "bbduk_cli")


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


def _obj_dir(tag: str) -> str:
    d = os.path.join(_HERE, "build", tag)
    os.makedirs(d, exist_ok=True)
    return d


def _compile(out: str, extra=(), verbose: bool = False, tag: str = "product") -> str:
    """One object per translation unit (rebuilt only when it or a header changed, all stale ones in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [d for d in _DEPS if d not in _SRC]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", *extra]
    objs, jobs = [], []
    for src in _SRC:
        obj = os.path.join(_obj_dir(tag), os.path.basename(src) + ".o")
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append([_hipcc(), *flags, "-x", "hip", "-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(run, jobs))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        _compile(_SO, verbose=verbose)
    build_cli(force=force, verbose=verbose)
    return _SO


def cli_path() -> str:
    return _CLI


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """The non-JVM caller: plain C++ above the two C ABIs, linked against the library beside it."""
    if not force and os.path.exists(_CLI) and os.path.getmtime(_CLI) > max(os.path.getmtime(_CLI_SRC), os.path.getmtime(_SO)):
        return _CLI
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
    cmd = ["g++", "-O2", "-g", "-rdynamic", "-pthread", "-std=c++17", _CLI_SRC, "-o", _CLI, "-L" + _HERE, "-lbbduk_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _CLI


def build_timing_variant(verbose: bool = False) -> str:
    """Experiment build with the BBDUK_DBG stage-deletion switches compiled in (profiles/ab.sh loads it through
    BBDUK_LIB_PATH); never the product library."""
    return _compile(os.path.join(_HERE, "ab_tsw.so"), ["-DBBDUK_TIMING_SWITCHES"], verbose, tag="tsw")


def build_variant(name: str, flags, verbose: bool = False) -> str:
    """Experiment build of the same sources with extra compiler flags (e.g. -DBBDUK_AB_ASCII_ONLY) as bbtools_amd/<name>.so,
    for same-box A/B runs through BBDUK_LIB_PATH; never the product library."""
    return _compile(os.path.join(_HERE, name + ".so"), list(flags), verbose, tag=name)
