set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_train_encoder.py -x -q -s > gpurun_out/r3_train_enc_tests.log 2>&1 || { tail -30 gpurun_out/r3_train_enc_tests.log; exit 1; }
tail -4 gpurun_out/r3_train_enc_tests.log
RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so timeout -k 10 300 python tools/bench_wgrad_levels.py > gpurun_out/r3_wgrad_levels.log 2>&1
grep -v "splits=[136]" gpurun_out/r3_wgrad_levels.log | tail -40
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -1
