set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_encoder.py tests/test_train_block.py -x -q > gpurun_out/r3_train_tests3.log 2>&1 || { tail -30 gpurun_out/r3_train_tests3.log; exit 1; }
tail -3 gpurun_out/r3_train_tests3.log
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -1
