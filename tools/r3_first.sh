set -e
mkdir -p gpurun_out
./tools/probe/gemm_timeline 3 4096 > gpurun_out/r3_timeline_ff1.txt 2>&1
./tools/probe/gemm_timeline 0 1536 > gpurun_out/r3_timeline_qkv.txt 2>&1
head -8 gpurun_out/r3_timeline_ff1.txt; head -8 gpurun_out/r3_timeline_qkv.txt
RALD_LIB_OVERRIDE=$PWD/rald_amd/librald_hip_probe.so python - > gpurun_out/r3_gn.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import importlib.util, os
spec = importlib.util.spec_from_file_location("ab", "tools/ablate_gemm.py"); ab = importlib.util.module_from_spec(spec); spec.loader.exec_module(ab)
for M in (32768,):
    for name, N, K, epi in [("ff1", 4096, 512, 3), ("qkv", 1536, 512, 0)]:
        print(M, name, " ".join(f"abl{a}:{ab.run(M, N, K, epi, 5, a):7.1f}us" for a in (64, 192, 64, 192, 64, 192)), flush=True)
PY
cat gpurun_out/r3_gn.txt
rocprofv3 -L > gpurun_out/r3_counters.txt 2>&1 || true
bash tools/pmc_mfma.sh 64 r03base
bash tools/pmc_mfma.sh 1 r03base
python tools/pmc_summarize.py gpurun_out/pmc_mfma_r03base_B64 > gpurun_out/r3_pmc_mfma_B64.csv
python tools/pmc_summarize.py gpurun_out/pmc_mfma_r03base_B1 > gpurun_out/r3_pmc_mfma_B1.csv
head -20 gpurun_out/r3_pmc_mfma_B64.csv
