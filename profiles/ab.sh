P="python profiles/bench_configs.py --only C2"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["kernel_ms"])'
for rep in 1 2; do
for d in 0 2 1 6 3; do
  echo -n "dbg$d old "; BBDUK_LIB_PATH=$PWD/bbtools_amd/ab_old.so BBDUK_DBG=$d $P | python -c "$J"
  echo -n "dbg$d new "; BBDUK_DBG=$d $P | python -c "$J"
done; done
