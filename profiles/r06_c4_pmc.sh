#!/bin/bash
# Round 6: fabric requests and issue counters of the 10^10-key scan per (m, load).  usage: bash profiles/r06_c4_pmc.sh "<m:load> ..." [extra bench_c4_full.py options]   (GPU box)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06/c4pmc; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ML in ${1:-10:55 9:50}; do
  M=${ML%%:*}; L=${ML##*:}
  CMD="python $ROOT/profiles/bench_c4_full.py --ref-bases 1e10 --check-reads 0 --steps 2 $( [ "$M" != 0 ] && echo --gm $M --load $L ) ${2:-}"
  i=0
  for C in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/m${M}_l${L}_p$i -- $CMD > $OUT/m${M}_l${L}_p$i.json 2> $OUT/m${M}_l${L}_p$i.err
  done
done
cd $ROOT
python - <<PY > $ROOT/gpurun_out/r06/c4_counters.txt
import csv,glob,collections,json,os,re
for tag in sorted(set(re.sub(r"_p\d+$", "", os.path.basename(d)) for d in glob.glob("$OUT/m*_p*") if os.path.isdir(d))):
    agg=collections.defaultdict(list); j={}
    for f in glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_bigs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("$OUT/%s_p*.json" % tag):
        for l in open(f):
            if l.startswith("{"): j=json.loads(l)
    reads=j.get("reads",1)
    print("# %s: per launch of bbduk_bigs_kernel (mean over the launches of a pass) and per read; %s reads per launch, kernel %.3f ms = %.1f Gbases/s, spilled keys %s of %s" % (tag, reads, j.get("kernel_ms",0), j.get("Gbases_per_s",0), j.get("spilled_keys"), j.get("keys")))
    for k in sorted(agg): print("  %-34s n=%d mean=%.6g per_read=%.3f" % (k, len(agg[k]), sum(agg[k])/len(agg[k]), sum(agg[k])/len(agg[k])/reads))
PY
cat $ROOT/gpurun_out/r06/c4_counters.txt
