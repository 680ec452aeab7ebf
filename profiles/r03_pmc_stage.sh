#!/bin/bash
# Instruction counts per stage of the stream kernel: one rocprofv3 --pmc pass per timing mask (bbtools_amd/ab_tsw.so).
# usage: bash profiles/r03_pmc_stage.sh <tag> mask [mask ...]
set -u
TAG=$1; shift; MASKS="$*"; READS=20000000
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcstage_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in $MASKS; do
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/m$M -- python $ROOT/profiles/stage_times.py --reads $READS $M > $OUT/m$M.json 2> $OUT/m$M.err
done
cd $ROOT
python - <<PY
import csv,glob,collections
n=$READS
for M in "$MASKS".split():
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/m%s/**/*counter_collection.csv" % M, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_stream_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v)/n for k,v in agg.items()}
    print("mask %-10s per read: VALU %6.1f SALU %5.1f LDS %4.1f VMEM_RD %4.2f BRANCH %4.1f  wave-cycles %5.0f wait %5.0f" % (M, m.get("SQ_INSTS_VALU",0), m.get("SQ_INSTS_SALU",0), m.get("SQ_INSTS_LDS",0), m.get("SQ_INSTS_VMEM_RD",0), m.get("SQ_INSTS_BRANCH",0), m.get("SQ_WAVE_CYCLES",0)*4, m.get("SQ_WAIT_ANY",0)*4))
PY
