#!/bin/bash
# HBM traffic of the big-layout wave kernel on BASELINE configs[3] (profiles/bench_c4_full.py): FETCH_SIZE / WRITE_SIZE in their own passes.
# usage: bash profiles/c4_fetch.sh <tag> <ref-bases>
set -u
TAG=${1:-r02}; REF=${2:-1e10}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c4fetch_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/profiles/bench_c4_full.py --ref-bases $REF --check-reads 0 --steps 2"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- $CMD > $OUT/$C.json 2> $OUT/$C.err
done
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections,json
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# bbduk_wave_kernel<KFILTER, .., BIG>, per launch; reference $REF bases, 20 M reads per launch")
for k in sorted(agg): print("%-12s n=%d mean=%.6g KiB = %.2f GB" % (k, len(agg[k]), sum(agg[k])/len(agg[k]), sum(agg[k])/len(agg[k])*1024/1e9))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk" in r["Name"]: print("trace", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "pct", r["Percentage"])
try: print(open("$OUT/FETCH_SIZE.json").read()[:1500])
except Exception as e: print(e)
PY
cat $OUT/summary.txt
