timeout -k 10 500 python tools/ab_libs_nfe.py 4,6,8,12 2 rald_amd/librald_hip.so rald_amd/librald_hip_sm128.so rald_amd/librald_hip_sm96.so
for v in "" _sm128; do RALD_LIB_OVERRIDE=rald_amd/librald_hip$v.so timeout -k 10 200 python tools/bench_train_full.py 8 2>&1 | tail -1; done
