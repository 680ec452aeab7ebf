#!/bin/bash
# Round 6: the 10^10-key map on 128-byte two-half lines -- m (minimizer length) x load sweep.  usage: bash profiles/r06_c4_sweep.sh "<m:load> ..."  (GPU box)
OUT=gpurun_out/r06; mkdir -p $OUT
for ML in ${1:-10:60 10:50 9:50 9:55}; do
  M=${ML%%:*}; L=${ML##*:}
  timeout -s KILL 400 python profiles/bench_c4_full.py --ref-bases 1e10 --reads 20000000 --steps 3 --check-reads 50000 --gm $M --load $L 2>$OUT/c4_m${M}_l${L}.err | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l); continue
    d.pop('line_load_histogram', None); print(json.dumps(d))" >> $OUT/c4_sweep.jsonl
  tail -2 $OUT/c4_m${M}_l${L}.err
done
cat $OUT/c4_sweep.jsonl
