set -e
export RALD_LIB_OVERRIDE=rald_amd/librald_hip_probe.so
for v in 192 128 96 64; do echo "RALD_GEMM_SMALL_MAX=$v"; RALD_GEMM_SMALL_MAX=$v timeout -k 10 200 python tools/sweep_nfe.py 4,8,12,16 2>&1 | grep "B="; done
unset RALD_LIB_OVERRIDE
timeout -k 10 300 python tools/bench_train_full.py 8 2>&1 | tail -1
