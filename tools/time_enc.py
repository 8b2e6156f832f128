"""Scratch: radar-condition encode time at B = 1, 4, 8 (radar encoder + condition cache of the 24-block denoiser)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import config, models_radar_generation as G, weights, synth
m = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
m = m.cuda(); h = m._handle()
for B in (1, 4, 8):
    cube = synth.radar_cube(B).cuda()
    for _ in range(3): h.encode_cond(cube, want_tokens=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): h.encode_cond(cube, want_tokens=False)
    torch.cuda.synchronize()
    print(f"cond encode B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
