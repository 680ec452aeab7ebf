set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02_pytest_gpu.txt
bash profiles/run_profile.sh r02 > gpurun_out/r02_profile.log 2>&1
python profiles/bench_modes.py > gpurun_out/r02_modes.json 2> gpurun_out/r02_modes.err
( export TMPDIR=/tmp; R=$(pwd); cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_modes -- python $R/profiles/bench_modes.py > /dev/null 2> $R/gpurun_out/prof_r02_modes.err )
python bench.py > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err
python profiles/stress_parity.py --seconds 240 --seed 202 > gpurun_out/r02_soak.json 2> gpurun_out/r02_soak.err
tail -3 gpurun_out/r02_pytest_gpu.txt; tail -2 gpurun_out/r02_soak.json | cut -c1-300; cat gpurun_out/r02_modes.json | cut -c1-150; tail -1 gpurun_out/r02_bench_full.json | cut -c1-600
