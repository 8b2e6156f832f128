set -e
python -m pytest tests -q -m gpu -s > gpurun_out/r3_gpu_tests_verbose.txt 2>&1 || (grep -E "FAILED|Error|assert" gpurun_out/r3_gpu_tests_verbose.txt | head -30; tail -5 gpurun_out/r3_gpu_tests_verbose.txt; exit 1)
tail -3 gpurun_out/r3_gpu_tests_verbose.txt
python bench.py --no-extras > gpurun_out/r3_bench_b128.json 2> gpurun_out/r3_bench_b128.err || tail -5 gpurun_out/r3_bench_b128.err
cat gpurun_out/r3_bench_b128.json
