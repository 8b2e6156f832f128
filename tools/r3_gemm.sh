set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/r3_gemm_tests.txt 2>&1 || (tail -30 gpurun_out/r3_gemm_tests.txt; exit 1)
tail -3 gpurun_out/r3_gemm_tests.txt
./tools/probe/gemm_timeline 3 4096 > gpurun_out/r3_timeline_ff1.txt 2>&1
./tools/probe/gemm_timeline 0 1536 > gpurun_out/r3_timeline_qkv.txt 2>&1
head -5 gpurun_out/r3_timeline_ff1.txt; head -5 gpurun_out/r3_timeline_qkv.txt
python tools/bench_gemm.py -1 64 2>&1 | tee gpurun_out/r3_bench_gemm.txt
python tools/sweep_nfe.py 2>&1 | tee gpurun_out/r3_sweep.txt
