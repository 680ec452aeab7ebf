#!/bin/bash
# Instruction counts with stages deleted (needs bbtools_amd/ab_tsw.so = _build.build_timing_variant()): one rocprofv3 --pmc
# pass per BBDUK_DBG mask.  Bits: 1 filter-no-gather, 2 keys+hashes only, 6 gathers-no-match, 3 no main scan, 4 no short
# k-mers, 5 no staging loads, 7 no decide.  usage: bash profiles/stage_insts.sh [reads] [mask ...]
set -u
READS=${1:-20000000}; shift || true
MASKS=${@:-"0 8 16 128 4 2 64 32 184"}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stage_insts; mkdir -p $OUT
export TMPDIR=/tmp BBDUK_LIB_PATH=$ROOT/bbtools_amd/ab_tsw.so
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
for d in $MASKS; do
  BBDUK_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/d$d -- $CMD > $OUT/d$d.json 2> $OUT/d$d.err
done
cd $ROOT
python - <<PY
import csv,glob,collections
for d in "$MASKS".split():
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/d%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_wave" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    n=$READS
    print("mask %4s  per read: VALU %6.1f SALU %5.1f LDS %4.1f VMEM_RD %4.2f BRANCH %4.1f  wave-cycles(x4)/read %5.0f" % (d, m["SQ_INSTS_VALU"]/n, m["SQ_INSTS_SALU"]/n, m["SQ_INSTS_LDS"]/n, m["SQ_INSTS_VMEM_RD"]/n, m["SQ_INSTS_BRANCH"]/n, m["SQ_WAVE_CYCLES"]*4/n))
PY
