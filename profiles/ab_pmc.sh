#!/bin/bash
# A/B of two builds of the library (BBDUK_LIB_PATH) with SQ counters; usage: bash profiles/ab_pmc.sh <tag> <lib.so> [reads]
set -u
TAG=$1; LIB=$2; READS=${3:-20000000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abpmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp BBDUK_LIB_PATH=$LIB
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
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
for k in sorted(agg): print("$TAG %-24s n=%d mean=%.6g" % (k, len(agg[k]), sum(agg[k])/len(agg[k])))
PY
