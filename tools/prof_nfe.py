"""Scratch: a few NFEs at B=64 in the mode given by RALD_QKV_DTYPE, for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, weights, synth
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0))
m = m.cuda(); h = m._handle(512, 64); h.set_sigmas([1.0])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
for _ in range(8): h.denoise(x, cache, 0)
torch.cuda.synchronize()
