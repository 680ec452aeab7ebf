#!/bin/bash
# Instruction counts per read of the main kernel on reads of one length, for two builds of the library (A/B):
# usage: bash profiles/readlens_insts.sh <read_len> [other .so]
set -u
L=${1:-151}; ALT=${2:-}
ROOT=$(pwd); export TMPDIR=/tmp
for which in product variant; do
  [ $which = variant ] && [ -z "$ALT" ] && continue
  OUT=$ROOT/gpurun_out/readlens_insts_$which; rm -rf $OUT; mkdir -p $OUT
  LIBARG=""; [ $which = variant ] && LIBARG="--lib $ALT"
  ( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/p -- python $ROOT/profiles/bench_readlens.py $LIBARG --bases 6e8 $L > $OUT/run.json 2> $OUT/run.err )
  python - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bbduk_wave" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n=json.loads(open("$OUT/run.json").read().strip().splitlines()[-1])["reads"]
for k,cs in agg.items():
    m={c:sum(v)/len(v) for c,v in cs.items()}
    print("$which L=$L %-48s per read: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f  wave-cycles(x4)/read %.0f wait %.0f" % (k, m["SQ_INSTS_VALU"]/n, m["SQ_INSTS_SALU"]/n, m["SQ_INSTS_LDS"]/n, m["SQ_INSTS_VMEM_RD"]/n, m["SQ_WAVE_CYCLES"]*4/n, m["SQ_WAIT_ANY"]*4/n))
PY
done
