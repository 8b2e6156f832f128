"""How much of a batch-1 block's time is the first touch of its weights?  NFE time per block for shallow models (weights stay in
L2 / the 256-MB memory-side cache between NFEs) against the 24-block model (300 MB of bf16 weights stream from HBM every NFE)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, weights, synth
def timed(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
res = {}
for depth in (1, 2, 4, 8, 16, 24):
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix=""), 0)); m = m.cuda()
    h = m._handle(512, 64); h.set_sigmas([1.0])
    for B in (1,):
        x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
        res[depth] = timed(lambda: h.denoise(x, cache, 0))
    del m, h
ds = sorted(res)
for a, b in zip(ds, ds[1:]):
    print(f"depth {a:2d} -> {b:2d}: {res[a]:7.1f} -> {res[b]:7.1f} us per NFE, marginal {(res[b]-res[a])/(b-a):6.2f} us per block ({12.5*b:.0f} MB of weights)")
