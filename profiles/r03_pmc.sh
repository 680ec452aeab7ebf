#!/bin/bash
# Instruction and stall counters of the first-hit kernel on the readlens-style 2xL workload, stream kernel vs pair-scan kernel, product build.
# usage: bash profiles/r03_pmc.sh <tag> [read_len] [reads]
set -u
TAG=${1:-a}; RL=${2:-150}; READS=${3:-20000000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for V in stream pair; do
  FLAG=""; [ $V = pair ] && FLAG="--pair-scan"
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$V$i -- python $ROOT/profiles/stage_times.py --lib libbbduk_hip --reads $READS --read-len $RL $FLAG 0 > $OUT/$V$i.json 2> $OUT/$V$i.err
  done
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
n=$READS
for V in ("stream","pair"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s*/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_stream_kernel" in r["Kernel_Name"] or "bbduk_wave_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# %s kernel, 2x$RL, per launch of %d reads (mean over launches); per read in brackets" % (V, n))
    for k in sorted(agg): print("%-26s n=%d mean=%.5g  [%.2f]" % (k, len(agg[k]), sum(agg[k])/len(agg[k]), sum(agg[k])/len(agg[k])/n))
PY
cat $OUT/summary.txt; tail -n 3 $OUT/stream1.err
