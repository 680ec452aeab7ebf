"""
bbtools_amd -- MI355X-native BBDuk k-mer matching path (ktrim / kfilter).

Only what the hot path needs lives here:
  csrc/bbduk_hip.hip   gfx950 kernels + the C ABI declared in include/bbduk_gpu.h
  csrc/bbduk_host.cpp  C++ host mirror of BBDukParser / BBDukLoader+BBDukIndexMod (include/bbduk_host.h)
  bbduk.py             Python binding over the C ABI (ctypes), same names as the reference's operators
  dist.py              read-sharding across GPUs + the single counter all-reduce (RCCL via torch.distributed)
The extension is mandatory: importing bbtools_amd.bbduk without libbbduk_hip.so raises.
"""
from ._build import build, lib_path, stale  # noqa: F401
