#!/bin/bash
# usage: bash profiles/run_pmc.sh <tag> <reads> "<counters pass1>" "<counters pass2>" ...
set -u
TAG=$1; READS=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-28s n=%d mean=%.6g" % (k, len(agg[k]), sum(agg[k])/len(agg[k])))
PY
