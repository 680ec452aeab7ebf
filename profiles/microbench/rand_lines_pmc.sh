#!/bin/bash
# (the binary: mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines)
[ -x profiles/microbench/_bin/rand_lines ] || { mkdir -p profiles/microbench/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_lines.hip -o profiles/microbench/_bin/rand_lines; }
# TLB / fabric counters of the random-line gathers: whole 222 GiB span against 1 GiB slices (sums over all launches of a run).
# usage: bash profiles/microbench/rand_lines_pmc.sh   (on the GPU box; writes gpurun_out/r04/rand_lines_pmc.txt)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04/rlpmc; mkdir -p $OUT
B=$ROOT/profiles/microbench/_bin/rand_lines
export TMPDIR=/tmp; cd /tmp
for SL in 0 1024; do
  i=0
  for C in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $OUT/s${SL}_p$i -- $B 222 2000 $SL 4 4 16 > $OUT/s${SL}_p$i.json 2> $OUT/s${SL}_p$i.err
  done
done
cd $ROOT
python - <<PY > gpurun_out/r04/rand_lines_pmc.txt
import csv,glob,collections
for sl in (0,1024):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for f in glob.glob("$OUT/s%d_p*/**/*counter_collection.csv"%sl, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("# rand_lines 222 GiB, 2000 M gathers x 3 repetitions, slice MiB = %d (0 = whole span): sums over all gather launches"%sl)
    for k in sorted(agg): print("%-40s launches=%d sum=%.6g"%(k,n[k],agg[k]))
PY
cat gpurun_out/r04/rand_lines_pmc.txt
