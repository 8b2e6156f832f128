"""Scratch: epilogue ablations of gemm_resid_ln (RALD_NT_STORE bits: 1 nt, 2 skip x store, 4 skip h store), one process per setting."""
import os, subprocess, sys
for v in ("1", "3", "5", "7", "0", "1"):
    print("== RALD_NT_STORE =", v, flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_resid_ln.py"), "64"], env=dict(os.environ, RALD_NT_STORE=v))
