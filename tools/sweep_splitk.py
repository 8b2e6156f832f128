"""Scratch: batch sweep of the NFE for several RALD_SPLITK_MAXM settings (one process each)."""
import os, subprocess, sys
for v in ("2048", "4096", "8192", "16384", "2048"):
    print("== RALD_SPLITK_MAXM =", v, flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_nfe.py"), "4,8,16,32"], env=dict(os.environ, RALD_SPLITK_MAXM=v))
