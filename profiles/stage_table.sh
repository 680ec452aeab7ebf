#!/bin/bash
# Round 5: the per-stage instruction budget of a kernel by stage deletion -- one rocprofv3 --pmc pass of profiles/stage_times.py per
# timing mask on the -DBBDUK_TIMING_SWITCHES build (bbtools_amd/ab_tsw.so), wave-instructions per read and kernel time per mask, then stage = mask 0
# minus the mask.  usage: bash profiles/stage_table.sh <tag> <kernel-name substring> "<stage_times.py options>" mask [mask ...]
set -u
TAG=$1; KPAT=$2; OPTS=$3; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stage_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in "$@"; do
  eval rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/m$m -- python $ROOT/profiles/stage_times.py $OPTS $m > $OUT/m$m.json 2> $OUT/m$m.err
done
cd $ROOT
python - "$OUT" "$KPAT" "$@" <<'PY'
import csv, glob, collections, json, sys
out, kpat, masks = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = {}
for m in masks:
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/m%s/**/*counter_collection.csv" % (out, m), recursive=True):
        for r in csv.DictReader(open(f)):
            if kpat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    try:
        j = json.loads([l for l in open("%s/m%s.json" % (out, m)) if l.startswith("{")][-1])
    except Exception:
        j = {"reads": 1, "kernel_ms": float("nan")}
    n = j["reads"]
    rows[m] = {k: sum(v) / len(v) / n for k, v in agg.items()}
    rows[m]["ms"] = j["kernel_ms"]
base = rows[masks[0]]
print("%-14s %8s %7s %7s %6s %7s | %8s %7s %7s" % ("mask", "ms", "VALU", "SALU", "LDS", "VMEM", "d ms", "d VALU", "d SALU"))
for m in masks:
    r = rows[m]
    g = lambda k: r.get(k, float("nan"))
    print("%-14s %8.3f %7.1f %7.1f %6.1f %7.2f | %8.3f %7.1f %7.1f" % (m, r["ms"], g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"),
          base["ms"] - r["ms"], base.get("SQ_INSTS_VALU", 0) - g("SQ_INSTS_VALU"), base.get("SQ_INSTS_SALU", 0) - g("SQ_INSTS_SALU")))
PY
