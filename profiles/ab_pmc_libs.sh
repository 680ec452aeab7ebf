ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06/abpmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for LIB in libbbduk_hip ab_r5; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$LIB -- python $ROOT/profiles/stage_times.py --genome 4600000 --args "k=31 hdist=0" --lib $LIB 0 > $OUT/$LIB.json 2> $OUT/$LIB.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/${LIB}_m -- python $ROOT/profiles/stage_times.py --genome 4600000 --args "k=31 hdist=0" --lib $LIB 0 > $OUT/${LIB}_m.json 2> $OUT/${LIB}_m.err
done
cd $ROOT
python - <<PY
import csv,glob,collections,json
for lib in ("libbbduk_hip","ab_r5"):
    agg=collections.defaultdict(list)
    for d in (lib, lib+"_m"):
        for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
            for r in csv.DictReader(open(f)):
                if "bbduk_bigs" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    j=json.loads([l for l in open("$OUT/%s.json"%lib) if l.startswith("{")][-1])
    print(lib, j["kernel_ms"], {k: round(sum(v)/len(v)/j["reads"],2) for k,v in sorted(agg.items())})
PY
