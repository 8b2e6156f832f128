set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_dit.py -x -q > gpurun_out/r3_last_tests.log 2>&1 || { tail -30 gpurun_out/r3_last_tests.log; exit 1; }
tail -2 gpurun_out/r3_last_tests.log
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r3_bench_last.json 2> gpurun_out/r3_bench_last.err
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_last.json')); print(d['value'], d['ms_per_step'], d['config']['nfe_schedule'], d['roofline']['frac'], d['roofline']['launches_timed']); [print(r['kernel'][:30], r['launches_timed'], round(r['avg_launch_us'],1), round(r['frac'],3)) for r in d['roofline_resid_ln']]"
timeout -k 10 200 python tools/ab_two_stream.py 128
