#!/bin/bash
# Round 6: fabric requests per read of the 10^10-key scan at the library's own geometry (one short pass; the six-counter TCC pass of r06_c4_pmc.sh hung).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06/c4rd; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p1 -- python $ROOT/profiles/bench_c4_full.py --ref-bases 1e10 --check-reads 0 --steps 2 ${1:-} > $OUT/p1.json 2> $OUT/p1.err
cd $ROOT
python - <<PY | tee $ROOT/gpurun_out/r06/c4_rdreq.txt
import csv,glob,collections,json
agg=collections.defaultdict(list); j={}
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_bigs" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for l in open("$OUT/p1.json"):
    if l.startswith("{"): j=json.loads(l)
reads=j.get("reads",1)
print("# bbduk_bigs_kernel at the library's own geometry for 10^10 keys (m = 9, %s of %s keys spilled, table %s bytes); %s reads per launch, kernel %.3f ms = %.1f Gbases/s under the counters" % (j.get("spilled_keys"), j.get("keys"), j.get("table_bytes"), reads, j.get("kernel_ms",0), j.get("Gbases_per_s",0)))
for k in sorted(agg): print("  %-24s n=%d mean=%.6g per_read=%.3f" % (k, len(agg[k]), sum(agg[k])/len(agg[k]), sum(agg[k])/len(agg[k])/reads))
PY
