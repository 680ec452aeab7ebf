#!/bin/bash
# Request granularity sweep (round 6): rate per allocation kind x load flavour, then the fabric request-size counters for the interesting ones.
# usage: bash profiles/microbench/rand_gran_sweep.sh   (on the GPU box; writes gpurun_out/r06/rand_gran.jsonl and rand_gran_pmc.txt)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT/rg
B=$ROOT/profiles/microbench/_bin/rand_gran
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rand_gran.hip -o $B
SPAN=${SPAN:-200}
: > $OUT/rand_gran.jsonl
for A in 0 1 2; do for F in 0 1 2 3 4 5; do
  timeout -s KILL 120 $B $SPAN 2000 $A $F 64 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
done; done
timeout -s KILL 120 $B $SPAN 2000 0 0 128 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
timeout -s KILL 120 $B $SPAN 2000 0 0 32 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
timeout -s KILL 120 $B $SPAN 2000 0 0 128 4 1 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
timeout -s KILL 120 $B $SPAN 2000 1 0 128 4 1 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
timeout -s KILL 120 $B $SPAN 2000 0 0 64 8 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
timeout -s KILL 120 $B $SPAN 2000 1 0 64 8 >> $OUT/rand_gran.jsonl 2>> $OUT/rand_gran.err
cat $OUT/rand_gran.jsonl
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD_[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_EA0_RDRET[A-Za-z0-9_]*" | sort -u > $OUT/rdreq_counters.txt
cat $OUT/rdreq_counters.txt
for AF in "0 0" "0 1" "0 4" "1 0" "1 4" "2 0"; do
  set -- $AF
  timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $OUT/rg/a$1_f$2 -- $B $SPAN 500 $1 $2 64 > $OUT/rg/a$1_f$2.json 2> $OUT/rg/a$1_f$2.err
done
cd $ROOT
python - <<PY > $OUT/rand_gran_pmc.txt
import csv,glob,collections,os
for d in sorted(glob.glob("$OUT/rg/a*_f*")):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("#", os.path.basename(d), open(d+".json").read().strip())
    for k in sorted(agg): print("  %-32s launches=%d sum=%.6g"%(k,n[k],agg[k]))
PY
cat $OUT/rand_gran_pmc.txt
