#!/bin/bash
# PC sampling of the dominant kernel on the configs[1] launch.  Runs on the GPU box (gpurun).
# Asks rocprofv3-avail what the agent supports, then tries stochastic (hardware) and host-trap sampling over a few
# intervals; keeps the exact error text of whatever is unavailable.
# usage: bash profiles/pc_sample.sh <tag> [reads] [extra bench args...]
set -u
TAG=${1:-r03}; READS=${2:-20000000}; shift 2 || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pcs_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
CMD="python $ROOT/bench.py --reads $READS --steps 6 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp
{ echo "## rocprofv3-avail list --pc-sampling"; rocprofv3-avail list --pc-sampling; echo "## rocprofv3-avail info --pc-sampling"; rocprofv3-avail info --pc-sampling; } > $OUT/avail_pcs.txt 2>&1
for M in "stochastic cycles 1048576" "stochastic cycles 65536" "stochastic cycles 4194304" "host_trap time 1000" "host_trap time 100" "host_trap time 10000" "host_trap time 1"; do
  set -- $M
  [ -e $OUT/$1.ok ] && continue
  D=/tmp/pcs_$1_$3; rm -rf $D
  timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 \
      --pc-sampling-interval $3 --kernel-trace --output-format csv json -d $D -- $CMD > $OUT/$1_$3.bench.json 2> $OUT/$1_$3.err
  rc=$?
  echo "rc=$rc method=$1 unit=$2 interval=$3" >> $OUT/status.txt
  ls -laR $D >> $OUT/$1_$3.files.txt 2>&1
  python $ROOT/profiles/pc_sample_summarize.py $D $OUT/$1_$3 >> $OUT/status.txt 2>&1 && [ $rc -eq 0 ] && touch $OUT/$1.ok
  tail -c 3000 $OUT/$1_$3.err > $OUT/$1_$3.err.tail; rm -f $OUT/$1_$3.err
done
cat $OUT/avail_pcs.txt $OUT/status.txt
