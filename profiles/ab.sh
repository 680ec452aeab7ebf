#!/bin/bash
# A/B timing of two library builds on the same box: ab_old.so (any earlier build copied there) vs the current one.
# With a -DBBDUK_TIMING_SWITCHES build (bbtools_amd/ab_tsw.so, see _build.build_timing_variant) BBDUK_DBG=<n> deletes a stage.
CFG=${1:-C2}
P="python profiles/bench_configs.py --only $CFG"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["kernel_ms"])'
for rep in 1 2; do
  [ -f bbtools_amd/ab_old.so ] && { echo -n "old "; BBDUK_LIB_PATH=$PWD/bbtools_amd/ab_old.so $P 2>/dev/null | python -c "$J"; }
  echo -n "new "; $P 2>/dev/null | python -c "$J"
done
if [ -f bbtools_amd/ab_tsw.so ]; then
  for d in 0 4 2 64 8 16 128 184; do echo -n "tsw mask$d "; BBDUK_LIB_PATH=$PWD/bbtools_amd/ab_tsw.so BBDUK_DBG=$d $P 2>/dev/null | python -c "$J"; done
fi
