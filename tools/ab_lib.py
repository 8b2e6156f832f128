"""Scratch A/B of two builds of the library on the GEMM shapes (separate processes per lib)."""
import os, subprocess, sys
for lib in sys.argv[1:]:
    env = dict(os.environ, RALD_LIB_OVERRIDE=lib)
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "tools/bench_gemm.py", "-1", "64"], env=env)
