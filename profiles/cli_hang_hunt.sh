#!/bin/bash
# Round 5: hunt the intermittent stall of `bbduk_cli devices=0,0,0` (one process, three handles on one device, three host threads,
# a local communicator at the end).  PAR copies of the run at a time (contention is what a production host has), RUNS in all; every run carries
# watchdog=SECS, which dumps every thread's stack and exits 97 if the process has not ended in time.  Output: gpurun_out/hang_hunt/{summary.txt,hang_*.log}
# usage: bash profiles/cli_hang_hunt.sh [RUNS] [PAR] [SECS] [extra cli args...]
RUNS=${1:-300}; PAR=${2:-6}; SECS=${3:-60}; shift 3 2>/dev/null
ROOT=$(pwd); OUT=$ROOT/gpurun_out/hang_hunt; mkdir -p $OUT
G=$ROOT/tests/golden
one() {
  i=$1; t0=$(date +%s.%N)
  $ROOT/bbtools_amd/bbduk_cli in=$G/cli_c2_r1.fq in2=$G/cli_c2_r2.fq tsv=/tmp/hh_$i.tsv batch=14 devices=0,0,0 watchdog=$SECS ktrim=r k=23 mink=11 hdist=1 ref=adapters "${@:2}" > /tmp/hh_$i.out 2> /tmp/hh_$i.err
  rc=$?; t1=$(date +%s.%N)
  if [ $rc -ne 0 ]; then cp /tmp/hh_$i.err $OUT/hang_$i.log; fi
  if ! cmp -s /tmp/hh_$i.tsv $G/cli_c2.tsv; then echo "run $i: tsv differs (rc $rc)" >> $OUT/summary.txt; fi
  echo "$i $rc $(awk "BEGIN{print $t1 - $t0}")" >> $OUT/runs.txt
  rm -f /tmp/hh_$i.tsv /tmp/hh_$i.out /tmp/hh_$i.err
}
: > $OUT/runs.txt; : > $OUT/summary.txt
i=0
while [ $i -lt $RUNS ]; do
  for p in $(seq 1 $PAR); do i=$((i+1)); [ $i -le $RUNS ] && one $i "$@" & done
  wait
done
awk '{n++; if ($2!=0) bad++; s+=$3; if ($3>m) m=$3} END {printf "runs %d  non-zero exits %d  mean %.2f s  max %.2f s\n", n, bad+0, s/n, m}' $OUT/runs.txt >> $OUT/summary.txt
cat $OUT/summary.txt; ls $OUT | head
