set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_pytest_dpp.log 2>&1 || { tail -30 gpurun_out/r3_pytest_dpp.log; exit 1; }
tail -2 gpurun_out/r3_pytest_dpp.log
timeout -k 10 200 python tools/sweep_nfe.py 1,2,8,64,128 2>&1 | grep "B="
timeout -k 10 200 python - <<'PY'
import torch
from rald_amd import bench_ae
r = bench_ae.run((1,))
print({k: round(v, 4) for k, v in r.items() if isinstance(v, float) and k.startswith("ae_")})
PY
