#!/bin/bash
# Runs on the GPU box: kernel stats and FETCH_SIZE / SQ counters of the genome-size case (k=31 vs a 4.6 Mbase genome, big layout + twin), round 3's new default.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_mid; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/profiles/exp_layout_mid.py --layouts default --genomes 4600000"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave_kernel" in r["Kernel_Name"] or "bbduk_stream_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-62s %-18s n=%d mean=%.6g" % (k[0], k[1], len(agg[k]), sum(agg[k])/len(agg[k])))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print("stats:", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat $OUT/summary.txt; cat $OUT/trace.json | cut -c1-300
