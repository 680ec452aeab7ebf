#!/bin/bash
# PMC passes over the HBM-resident configuration (profiles/bench_c4_full.py): HBM traffic and stall counters of the big-layout wave kernel.
# usage: bash profiles/c4_pmc.sh <tag> <ref-bases> ; counters in their own passes, no tracing (gpurun refuses the combination)
set -u
TAG=${1:-r02}; REF=${2:-1e9}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c4pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/profiles/bench_c4_full.py --ref-bases $REF --check-reads 0 --steps 2"
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# bbduk_wave_kernel<.., BIG>, per launch (mean over the launches of a pass); reference $REF bases, 20 M reads per launch")
for k in sorted(agg): print("%-34s n=%d mean=%.6g" % (k, len(agg[k]), sum(agg[k])/len(agg[k])))
PY
cat $OUT/summary.txt; tail -2 $OUT/p1.json | cut -c1-600
