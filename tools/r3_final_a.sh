# Round-3 evidence run, part A: GPU tests, smoke, bench (B = 128 default and B = 64).
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r3_pytest_gpu.log 2>&1; tail -3 gpurun_out/r3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r3_bench_B128.json 2> gpurun_out/r3_bench_B128.err; tail -2 gpurun_out/r3_bench_B128.err
python bench.py --batch 64 --no-extras --no-cpu-baseline > gpurun_out/r3_bench_B64.json 2> gpurun_out/r3_bench_B64.err
cut -c1-600 gpurun_out/r3_bench_B128.json
