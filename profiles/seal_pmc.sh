#!/bin/bash
# Runs on the GPU box: kernel-trace stats and one FETCH_SIZE pass (its own run, no trace flags) over the Seal benchmark.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_seal; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/profiles/bench_seal.py --pairs 1000000 --check 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
cd $ROOT
python profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
grep -i "seal" $OUT/summary.txt
