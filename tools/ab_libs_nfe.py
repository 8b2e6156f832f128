"""Scratch A/B of library builds on whole NFEs: alternating subprocesses (one library each), ABAB..., same box.
usage: python tools/ab_libs_nfe.py <batch> <rounds> libA.so libB.so ..."""
import os, subprocess, sys
B, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
child = r'''
import sys, time, torch, os
sys.path.insert(0, os.getcwd())
from rald_amd import models_radar_generation as G, weights, synth
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0))
m = m.cuda(); h = m._handle(512, 64); h.set_sigmas([1.0])
res = []
for B in [int(b) for b in sys.argv[1].split(",")]:
    x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
    for _ in range(5): h.denoise(x, cache, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): h.denoise(x, cache, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    res.append(f"B={B}: {dt*1e3:.3f} ms {B/dt:.0f}/s")
print("  ".join(res), flush=True)
'''
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, RALD_LIB_OVERRIDE=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child, B], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
