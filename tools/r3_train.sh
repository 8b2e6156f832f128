set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_train_block.py -x -q -s > gpurun_out/r3_train_tests.log 2>&1 || { tail -30 gpurun_out/r3_train_tests.log; exit 1; }
tail -15 gpurun_out/r3_train_tests.log
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -2
