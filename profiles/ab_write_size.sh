# Round 6: HBM write bytes of the headline launch with the read-aligned blocks (mask 0) and without (22), timing build.  ONE derived counter per pass: WRITE_SIZE and FETCH_SIZE in one pass hang the run.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06/wrab; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for M in 0 22; do
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w$M -- python $ROOT/profiles/stage_times.py $M > $OUT/w$M.json 2> $OUT/w$M.err
done
cd $ROOT
python - <<PY
import csv,glob,collections
for m in ("0","22"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/w%s/**/*counter_collection.csv"%m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbduk_stream_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mask",m,{k:(len(v),sum(v)/len(v)) for k,v in agg.items()})
PY
