#!/bin/bash
# Instruction counts per read of two builds (ab_old.so vs current): one rocprofv3 --pmc pass each.
set -u
READS=${1:-20000000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ab_insts; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
BBDUK_LIB_PATH=$ROOT/bbtools_amd/ab_old.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/old -- $CMD > $OUT/old.json 2> $OUT/old.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/new -- $CMD > $OUT/new.json 2> $OUT/new.err
cd $ROOT
python - <<PY
import csv,glob,collections
for d in ["old","new"]:
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_wave" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    n=$READS
    print("%s  per read: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f BRANCH %.1f  wave-cycles(x4)/read %.0f" % (d, m["SQ_INSTS_VALU"]/n, m["SQ_INSTS_SALU"]/n, m["SQ_INSTS_LDS"]/n, m["SQ_INSTS_VMEM_RD"]/n, m["SQ_INSTS_BRANCH"]/n, m["SQ_WAVE_CYCLES"]*4/n))
PY
