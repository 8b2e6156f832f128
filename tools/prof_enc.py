"""Scratch: radar-condition encode (radar encoder + cond cache) at B=8 for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import config, models_radar_generation as G, weights, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
m = m.cuda(); h = m._handle()
cube = synth.radar_cube(B).cuda()
for _ in range(4): h.encode_cond(cube, want_tokens=False)
torch.cuda.synchronize()
