set -e
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "resid or gemm or ln" 2>&1 | tail -3
python tools/ab_libs_nfe.py 64,128 2 rald_amd/librald_hip_old.so rald_amd/librald_hip.so 2>&1 | tee gpurun_out/r3_ab_loop2.txt
