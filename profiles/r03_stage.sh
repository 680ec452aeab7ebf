#!/bin/bash
# usage: bash profiles/r03_stage.sh <tag>   -- headline A/B (stream vs pair scan) + stage deletion of both kernels on one box
TAG=${1:-a}; OUT=gpurun_out/stage_$TAG; mkdir -p $OUT
J='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], d.get("parity_sample_ok"), d["roofline"]["kernel_ms"], d["device_status"])'
for rep in 1 2; do
  echo -n "stream "; python bench.py --no-extras --cpu-seconds 3 --steps 5 2>$OUT/err_s$rep.txt | python -c "$J"
  echo -n "pair   "; python bench.py --no-extras --no-cpu-baseline --steps 5 --pair-scan 2>$OUT/err_p$rep.txt | python -c "$J"
done 2>&1 | tee $OUT/headline.txt
python profiles/stage_times.py 0 10 11 10+11 3 4 5 7 13 14 13+14 3+4 3+4+7 3+4+5+7 0 > $OUT/stream.jsonl 2> $OUT/err_st.txt
python profiles/stage_times.py --pair-scan 0 3 4 5 7 3+4 3+4+7 3+4+5+7 0 > $OUT/pair.jsonl 2>> $OUT/err_st.txt
cat $OUT/stream.jsonl $OUT/pair.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('pair  ' if d['pair_scan'] else 'stream', '%-10s' % d['bits'], d['kernel_ms'], d['Gbases_per_s'])"
tail -n 5 $OUT/err_st.txt
