#!/bin/bash
# Instruction counts per read of the secondary operators (ktrim=n, ktrim=rl, ksplit kernels): one rocprofv3 --pmc pass over
# profiles/bench_modes.py.  usage: bash profiles/modes_insts.sh [reads]
set -u
READS=${1:-4000000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/modes_insts; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python $ROOT/profiles/bench_modes.py --reads $READS > $OUT/run.json 2> $OUT/run.err
cd $ROOT
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_k" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n=$READS
for k,cs in agg.items():
    m={c:sum(v)/len(v) for c,v in cs.items()}
    print("%-40s per read: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f  wave-cycles(x4)/read %.0f" % (k, m["SQ_INSTS_VALU"]/n, m["SQ_INSTS_SALU"]/n, m["SQ_INSTS_LDS"]/n, m["SQ_INSTS_VMEM_RD"]/n, m["SQ_WAVE_CYCLES"]*4/n))
PY
cat $OUT/run.json
