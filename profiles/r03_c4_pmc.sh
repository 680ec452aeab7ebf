#!/bin/bash
# BASELINE configs[3] (10^10 keys, 239 GB map): what bounds bbduk_wave_kernel<KFILTER, .., BIG>?  TLB, L2->fabric read requests by size,
# DRAM credit stalls, latency, wave stall shares.  One rocprofv3 --pmc pass per counter group (each pass rebuilds the map: ~12 s).
# usage: bash profiles/r03_c4_pmc.sh <tag> [ref-bases]
set -u
TAG=${1:-a}; REF=${2:-1e10}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c4pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/profiles/bench_c4_full.py --ref-bases $REF --check-reads 0 --steps 2 ${3:-}"
cd /tmp
i=0
# (at most ~3 counters of one block per pass: five TCP counters at once is "Request exceeds the capabilities of the hardware to collect")
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_BUBBLE_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-include-regex "bbduk_(wave|stream)_kernel" --pmc $C --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $ROOT
python - <<PY > $OUT/summary.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave_kernel" in r["Kernel_Name"] or "bbduk_stream_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# bbduk_wave_kernel<KFILTER, .., BIG>, per launch (mean over the launches of a pass); reference $REF bases, 20 M reads per launch")
for k in sorted(agg): print("%-46s n=%d mean=%.6g" % (k, len(agg[k]), sum(agg[k])/len(agg[k])))
PY
cat $OUT/summary.txt; tail -1 $OUT/p1.json | cut -c1-700
