set -e
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3_gpu_tests.txt 2>&1 || (tail -40 gpurun_out/r3_gpu_tests.txt; exit 1)
tail -3 gpurun_out/r3_gpu_tests.txt
python tools/ab_two_stream.py 128,64,256 2>&1 | tee gpurun_out/r3_two_stream.txt
