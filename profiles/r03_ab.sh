#!/bin/bash
# Stream kernel against the pair-scan kernel on one box: GPU parity tests first, then the headline, the read lengths and the mixed batches.
# usage: bash profiles/r03_ab.sh <tag> [skip-tests]
TAG=${1:-a}; OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
J='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], d.get("parity_sample_ok"), d["roofline"]["kernel_ms"], d["device_status"])'
if [ -z "${2:-}" ]; then timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt; fi
for rep in 1 2; do
  echo -n "stream "; python bench.py --no-extras --cpu-seconds 4 --steps 5 2>$OUT/err_s$rep.txt | python -c "$J"
  echo -n "pair   "; python bench.py --no-extras --no-cpu-baseline --steps 5 --pair-scan 2>$OUT/err_p$rep.txt | python -c "$J"
done 2>&1 | tee $OUT/headline.txt
python profiles/bench_readlens.py 75 100 150 151 250 300 > $OUT/readlens_stream.jsonl 2>$OUT/err_rl.txt
python profiles/bench_readlens.py --pair-scan 75 100 150 151 250 300 > $OUT/readlens_pair.jsonl 2>>$OUT/err_rl.txt
python profiles/bench_mixed_lengths.py > $OUT/mixed_stream.jsonl 2>$OUT/err_mx.txt
python profiles/bench_mixed_lengths.py --pair-scan > $OUT/mixed_pair.jsonl 2>>$OUT/err_mx.txt
python - <<PY
import json
for n in ("readlens", "mixed"):
    a = [json.loads(l) for l in open("$OUT/%s_stream.jsonl" % n)]; b = [json.loads(l) for l in open("$OUT/%s_pair.jsonl" % n)]
    for x, y in zip(a, b):
        print(n, x.get("read_len", x.get("share_cut_to_2x100")), "stream", x["Gbases_per_s"], "pair", y["Gbases_per_s"])
PY
tail -3 $OUT/err_*.txt | tail -30
