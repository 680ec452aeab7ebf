#!/bin/bash
# Same-box A/B of two builds of the library over the operator launches (20 M reads of 2x150 per launch, kernel ms by HIP events, alternating A B A B).
# usage: bash profiles/ab_two_libs.sh <libA> <libB>      (names under bbtools_amd/ without .so)
A=${1:-ab_head}; B=${2:-libbbduk_hip}
for ARGS in "ktrim=r k=23 mink=11 hdist=1 ref=adapters" "ktrim=rl k=23 mink=11 hdist=1 ref=adapters" "ktrim=n k=23 mink=11 hdist=1 ref=adapters" "ksplit=t k=23 mink=11 hdist=1 ref=adapters" "ktrim=l k=23 mink=11 hdist=1 ref=adapters"; do
  echo "## $ARGS"
  for L in $A $B $A $B; do
    python profiles/stage_times.py --lib $L --args "$ARGS" 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  %-14s %.3f ms  %.1f Gbases/s' % ('$L', j['kernel_ms'], j['Gbases_per_s']))"
  done
done
