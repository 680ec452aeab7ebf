#!/bin/bash
# Same-box A/B of two builds of the library over big-layout launches (4.6 Mbase genome, 20 M reads of 2x150), alternating A B A B.
# usage: bash profiles/ab_two_libs_genome.sh <libA> <libB>
A=${1:-ab_head}; B=${2:-libbbduk_hip}
for CFG in "k=31 hdist=0|--big-layout 3 --lds-bits 10" "k=31 hdist=0|" "ktrim=r k=31 hdist=0|" "ktrim=l k=31 hdist=0|" "ktrim=rl k=31 hdist=0|"; do
  ARGS=${CFG%%|*}; EXTRA=${CFG##*|}
  echo "## $ARGS $EXTRA"
  for L in $A $B $A $B; do
    python profiles/stage_times.py --lib $L --genome 4600000 --args "$ARGS" $EXTRA 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  %-14s %.3f ms  %.1f Gbases/s' % ('$L', j['kernel_ms'], j['Gbases_per_s']))"
  done
done
