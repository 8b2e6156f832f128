set -e
mkdir -p gpurun_out
RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so timeout -k 10 300 python tools/bench_tn_target.py 2>&1 | grep gemm_tn
timeout -k 10 900 python -m pytest tests/test_train_encoder.py -x -q > gpurun_out/r3_train_tests4.log 2>&1 || { tail -30 gpurun_out/r3_train_tests4.log; exit 1; }
tail -2 gpurun_out/r3_train_tests4.log
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -1
