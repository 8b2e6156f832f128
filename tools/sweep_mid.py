"""Scratch: batch sweep of the NFE for the RALD_GEMM_MID engine choices (one process each)."""
import os, subprocess, sys
for v in ("0", "1", "2", "0"):
    print("== RALD_GEMM_MID =", v, flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_nfe.py"), "2,4,8,12"], env=dict(os.environ, RALD_GEMM_MID=v))
