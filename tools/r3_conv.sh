set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sampler.py tests/test_train_encoder.py -x -q > gpurun_out/r3_conv_tests.log 2>&1 || { tail -30 gpurun_out/r3_conv_tests.log; exit 1; }
tail -2 gpurun_out/r3_conv_tests.log
timeout -k 10 200 python tools/time_enc.py 2>&1 | grep "cond encode"
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -1
