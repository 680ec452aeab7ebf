#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage output -> one line per kernel.  usage: python profiles/kernel_resources.py res.txt [filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read(); flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt and flt not in name: continue
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        dem = name
    print("%-64s VGPR %3d  AGPR %3d  SGPR %3d  sgprSpill %3d  vgprSpill %3d  scratch %4d  occ %d  LDS %d" % (
        dem[-64:], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", r.get("SGPRs", -1)), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
        r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
