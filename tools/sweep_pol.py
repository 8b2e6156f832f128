"""Scratch: NFE time at B=64 for the bf16-output store cache policies (RALD_GEMM_ABLATE 0 / 128 / 256 / 384)."""
import os, subprocess, sys
for v in ("0", "128", "256", "384", "0", "128"):
    print("== RALD_GEMM_ABLATE =", v, flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_nfe.py"), "64"], env=dict(os.environ, RALD_GEMM_ABLATE=v))
