#!/bin/bash
# Runs on the GPU box: memory-side counters of one command's scan kernel, one rocprofv3 --pmc pass per counter group, every pass under a
# hard timeout (a pass that the profiler aborts must not hang the lease).  usage: profiles/r04_mem_pmc.sh <tag> <kernel-regex> -- <command ...>
set -u
TAG=$1; PAT=$2; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/mem_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="$*"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-include-regex "$PAT" --pmc $C --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("# per launch (mean over the launches of a pass); command: $CMD")
for k in sorted(agg): print("%-42s %-44s n=%d mean=%.6g" % (k[0], k[1], len(agg[k]), sum(agg[k])/len(agg[k])))
PY
cat $OUT/summary.txt; tail -1 $OUT/p1.json | cut -c1-400
