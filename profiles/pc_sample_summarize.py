"""Aggregates rocprofv3 PC-sampling output (csv and/or json) per instruction: samples, share, stall reasons when the
stochastic method provides them.  usage: pc_sample_summarize.py <rocprof out dir> <output prefix>"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
found = False
for f in glob.glob(os.path.join(src, "**", "*pc_sampling*.csv"), recursive=True):
    found = True
    rows = csv.DictReader(open(f))
    cols = rows.fieldnames
    print("csv", f, cols)
    by = collections.Counter()
    reasons = collections.defaultdict(collections.Counter)
    n = 0
    rcols = [c for c in cols if any(t in c.lower() for t in ("stall", "reason", "issued", "inst_type", "wave_issued", "snapshot"))]
    for r in rows:
        n += 1
        key = (r.get("Instruction", ""), r.get("Instruction_Comment", ""))
        by[key] += 1
        for c in rcols:
            reasons[key][c + "=" + r[c]] += 1
    with open(dst + "." + os.path.basename(f) + ".top.txt", "w") as o:
        o.write("# %d samples, %d distinct instructions; columns: %s\n" % (n, len(by), cols))
        for (ins, com), c in by.most_common(400):
            o.write("%8d %6.3f%%  %s  ; %s  %s\n" % (c, 100.0 * c / max(n, 1), ins, com,
                    " ".join("%s:%d" % kv for kv in reasons[(ins, com)].most_common(6))))
    print("samples", n)
for f in glob.glob(os.path.join(src, "**", "*results.json"), recursive=True):
    try:
        d = json.load(open(f))
    except Exception as e:
        print("json", f, "unreadable", e); continue
    found = True
    tool = d["rocprofiler-sdk-tool"][0]
    print("json", f, "keys", list(tool.keys()))
    bufs = tool.get("buffer_records", {})
    for k in bufs:
        if "pc_sampl" in k:
            recs = bufs[k]
            print(k, len(recs), recs[0] if recs else None)
    # keep the raw per-record json for the two pc-sampling buffers only when small
    for k in [k for k in bufs if "pc_sampl" in k]:
        if bufs[k]:
            with open(dst + "." + k + ".head.json", "w") as o:
                json.dump(bufs[k][:50], o, indent=1)
    strs = tool.get("strings", {})
    if "pc_sample_instructions" in strs:
        with open(dst + ".instructions.json", "w") as o:
            json.dump({"instructions": strs["pc_sample_instructions"], "comments": strs.get("pc_sample_comments")}, o)
if not found:
    print("no pc sampling output under", src)
    sys.exit(3)
