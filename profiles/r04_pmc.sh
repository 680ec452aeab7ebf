#!/bin/bash
# Runs on the GPU box: kernel trace + SQ issue counters + FETCH/WRITE of one command's scan kernel (separate rocprofv3 passes, as the guide asks).
# usage: profiles/r04_pmc.sh <tag> <kernel-name-substring> -- <command ...>
set -u
TAG=$1; PAT=$2; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="$*"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-62s %-20s n=%d mean=%.6g" % (k[0], k[1], len(agg[k]), sum(agg[k])/len(agg[k])))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print("stats:", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat $OUT/summary.txt; tail -3 $OUT/trace.json | cut -c1-400
