#!/bin/bash
# A/B of two builds of the library on one box: the product build against bbtools_amd/<name>.so (e.g. a -D variant made with
# bbtools_amd._build._compile(out, extra=[...], tag=...)).  Prints the headline's ms per step and Gbases/s, alternating, REPS times.
ALT=${1:?usage: ab_lib.sh <path to the other .so> [reads] [reps]}
READS=${2:-100000000}
REPS=${3:-2}
J='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], d.get("parity_sample_ok"), d["roofline"]["kernel_ms"])'
for rep in $(seq $REPS); do
  echo -n "product "; python bench.py --no-extras --no-cpu-baseline --reads $READS --steps 5 2>/dev/null | python -c "$J"
  echo -n "variant "; python bench.py --no-extras --no-cpu-baseline --reads $READS --steps 5 --lib $ALT 2>/dev/null | python -c "$J"
done
