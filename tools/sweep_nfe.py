"""Scratch perf probe: ms per NFE of the 24-block denoiser vs batch (cond tokens given)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, weights, synth

depth = 24
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix=""), 0))
m = m.cuda()
h = m._handle(512, 64)
h.set_sigmas([1.0])
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8, 16, 32, 64]
GF = 132.18
for B in batches:
    x = synth.latents(range(B)).cuda()
    cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
    for _ in range(3):
        h.denoise(x, cache, 0)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        h.denoise(x, cache, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}  {dt*1e3:8.3f} ms/NFE  {B/dt:9.1f} sample-NFE/s  {B*GF/dt/1e3:7.1f} TFLOP/s", flush=True)
