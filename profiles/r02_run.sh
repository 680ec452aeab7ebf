# the one gpurun command behind the r02_final / r02_modes / r02_soak / r02_seal files (run from the repository root on the GPU box)
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r02_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.txt 2>&1
bash profiles/run_profile.sh r02 > gpurun_out/r02_profile.log 2>&1
python profiles/bench_modes.py > gpurun_out/r02_modes.json 2> gpurun_out/r02_modes.err
python profiles/bench_seal.py > gpurun_out/r02_seal.json 2> gpurun_out/r02_seal.err
python profiles/bench_general.py 2> /dev/null | grep args > gpurun_out/r02_general.json
( export TMPDIR=/tmp; R=$(pwd); cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_modes -- python $R/profiles/bench_modes.py > /dev/null 2> $R/gpurun_out/prof_r02_modes.err )
python bench.py > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err
python profiles/stress_parity.py --seconds ${SOAK_SECONDS:-240} --seed 203 > gpurun_out/r02_soak.json 2> gpurun_out/r02_soak.err
cat gpurun_out/r02_pytest_gpu.txt; tail -1 gpurun_out/r02_smoke.txt; tail -2 gpurun_out/r02_soak.json | cut -c1-400; cat gpurun_out/r02_modes.json | cut -c1-150; cat gpurun_out/r02_seal.json | cut -c1-400; tail -1 gpurun_out/r02_bench_full.json | cut -c1-300
