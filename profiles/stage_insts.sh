#!/bin/bash
# Instruction counts per deleted stage (needs bbtools_amd/ab_tsw.so): one rocprofv3 --pmc pass per BBDUK_DBG value.
set -u
READS=${1:-20000000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stage_insts; mkdir -p $OUT
export TMPDIR=/tmp BBDUK_LIB_PATH=$ROOT/bbtools_amd/ab_tsw.so
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
for d in 0 3 4 7 2 1 6 5; do
  BBDUK_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/d$d -- $CMD > $OUT/d$d.json 2> $OUT/d$d.err
done
cd $ROOT
python - <<PY
import csv,glob,collections
for d in [0,3,4,7,2,1,6,5]:
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/d%d/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_wave" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    n=$READS
    print("dbg%d  per read: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f BRANCH %.1f  wave-cycles(x4)/read %.0f" % (d, m["SQ_INSTS_VALU"]/n, m["SQ_INSTS_SALU"]/n, m["SQ_INSTS_LDS"]/n, m["SQ_INSTS_VMEM_RD"]/n, m["SQ_INSTS_BRANCH"]/n, m["SQ_WAVE_CYCLES"]*4/n))
PY
