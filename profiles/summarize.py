"""Condenses a rocprofv3 output tree (kernel stats + counter CSVs) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:8]:
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"),
                                                            r.get("AverageNs"), r.get("Percentage")))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r.get("Kernel_Name", "")][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
    print("== %s (per-dispatch mean)" % sub)
    for k, cs in agg.items():
        if "bbduk_wave" not in k and "bbduk_stream" not in k and "bbduk_span" not in k:
            continue
        for c, vals in cs.items():
            print("  %-50s %-24s n=%d mean=%.6g" % (k[:50], c, len(vals), sum(vals) / len(vals)))
import json
print("== bench line of each profiled run (value, kernel_ms from the library's HIP events)")
for f in sorted(glob.glob(os.path.join(out, "*_bench.json"))):
    txt = open(f).read().strip().splitlines()
    if txt:
        try:
            d = json.loads(txt[-1])
            print("  %-22s value=%s %s ms_per_step=%s kernel_ms=%s reads=%s" % (os.path.basename(f), d["value"], d["unit"], d["ms_per_step"],
                  d["roofline"]["kernel_ms"], d["config"].get("reads_per_gpu")))
        except Exception:
            print("  %-22s %s" % (os.path.basename(f), txt[-1][:300]))
