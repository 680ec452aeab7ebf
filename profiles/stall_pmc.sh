#!/bin/bash
# Stall-reason counters of the dominant kernel (bbduk_wave_kernel) on the configs[1] launch: what paces it -- VALU pipe cycles,
# scalar pipe, instruction-issue waits, LDS, memory?  Runs on the GPU box (gpurun); counters in their own passes, no tracing.
# usage: bash profiles/stall_pmc.sh <tag> [reads] [extra bench args...]
set -u
TAG=${1:-r02}; READS=${2:-100000000}; shift 2 || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stall_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
         "SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_SENDMSG SQ_LEVEL_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# bbduk_wave_kernel, per launch (mean over the launches of a pass); reads per launch = $READS")
for k in sorted(agg): print("%-28s n=%d mean=%.6g" % (k, len(agg[k]), sum(agg[k])/len(agg[k])))
PY
cat $OUT/summary.txt
