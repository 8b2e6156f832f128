"""Scratch A/B of probe-build environment switches on whole NFEs: alternating subprocesses, same box.
usage: python tools/ab_env_nfe.py <batches> <rounds> "VAR=1 VAR2=0" "..." ...   (an empty string = defaults); uses the PROBE library"""
import os, subprocess, sys
B, rounds, cfgs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
child = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_libs_nfe.py")).read().split("child = r'''")[1].split("'''")[0]
lib = os.path.abspath("rald_amd/librald_hip_probe.so")
for r in range(rounds):
    for c in cfgs:
        env = dict(os.environ, RALD_LIB_OVERRIDE=lib)
        for kv in c.split():
            k, v = kv.split("="); env[k] = v
        out = subprocess.run([sys.executable, "-c", child, B], env=env, capture_output=True, text=True)
        print(f"{c or '(defaults)':40s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
