#!/bin/bash
# Round 6: the read-aligned form of the big layout's scan against its stream form (timing switch 22 turns the aligned form off), same box: time, instructions, L1 accesses, L2 requests
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06/alpmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
OPTS=${1:?usage: ab_aligned_pmc.sh "<stage_times.py options>"}
for M in 0 22; do
  eval rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/a$M -- python $ROOT/profiles/stage_times.py $OPTS $M > $OUT/a$M.json 2> $OUT/a$M.err
  eval rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/b$M -- python $ROOT/profiles/stage_times.py $OPTS $M > $OUT/b$M.json 2> $OUT/b$M.err
done
cd $ROOT
python - <<PY
import csv,glob,collections,json
for m in ("0","22"):
    agg=collections.defaultdict(list)
    for d in ("a"+m, "b"+m):
        for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
            for r in csv.DictReader(open(f)):
                if "bbduk_bigs" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    j=json.loads([l for l in open("$OUT/a%s.json"%m) if l.startswith("{")][-1])
    print("mask", m, "(aligned)" if m=="0" else "(stream form)", j["kernel_ms"], {k: round(sum(v)/len(v)/j["reads"],2) for k,v in sorted(agg.items())})
PY
