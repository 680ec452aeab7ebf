#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench command, summaries into gpurun_out/prof_*.
# usage: bash profiles/run_profile.sh <tag> [reads] [steps]
set -u
TAG=${1:-r01}; READS=${2:-100000000}; STEPS=${3:-3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --reads $READS --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc_bench.json 2> $OUT/pmc_tcc.err
cd $ROOT
python profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
