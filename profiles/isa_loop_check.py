#!/usr/bin/env python3
"""Where do a kernel's spills execute?  Compiles one translation unit to gfx950 assembly (device only, no GPU needed) and lists, for every loop the
compiler annotates in the chosen kernel, its depth and how many scratch_* (VGPR spill traffic) and v_readlane / v_writelane (SGPR spill traffic)
instructions sit inside it -- the resource report's static counts say nothing about that.
usage: python profiles/isa_loop_check.py <csrc file> <mangled-name substring> [more substrings ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
src = os.path.join(ROOT, "bbtools_amd", "csrc", sys.argv[1])
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
                           "-x", "hip", "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
for pat in sys.argv[2:]:
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and pat in l and l.rstrip().endswith(pat) is False and ":" in l)
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
    L = text[start:end]
    name = text[start].split(":")[0]
    loops = {}
    for i, l in enumerate(L):
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if m:
            key = (m.group(1), int(m.group(2))); loops.setdefault(key, [i, i]); loops[key][1] = i
    tot_s = sum("scratch_" in x for x in L); tot_l = sum(("v_readlane" in x or "v_writelane" in x) for x in L)
    print("%s\n  whole kernel: %d scratch_*, %d v_readlane/v_writelane, %d VALU lines" % (name, tot_s, tot_l, sum(x.strip().startswith("v_") for x in L)))
    for (h, d), (a, b) in sorted(loops.items(), key=lambda kv: kv[1][0]):
        seg = L[a:b + 1]
        print("  loop %-10s depth %d  %5d lines  VALU %4d  global_load %2d  scratch_* %2d  lane moves %3d" %
              (h, d, b - a, sum(x.strip().startswith("v_") for x in seg), sum(x.strip().startswith("global_load") for x in seg),
               sum("scratch_" in x for x in seg), sum(("v_readlane" in x or "v_writelane" in x) for x in seg)))
