"""Scratch: time of one training step of the 24-block denoiser (forward + backward + clip/AdamW/EMA), B per GPU."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, synth, train_dit as TD, weights
from rald_amd.train_utils import FlatAdamW

depth = 24
Bs = [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8"])]
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix=""), 0))
m = m.cuda()
named = dict(m.named_parameters())
opt = FlatAdamW(list(named.values()), lr=1e-4, ema=True)
tr = TD.DitTrainer(named, depth)
for B in Bs:
    y, cond = synth.normal([B, 512, 32], 1).cuda(), synth.cond_tokens(B).cuda()
    rnd, noise = synth.normal([B], 2), synth.normal([B, 512, 32], 3).cuda()
    def step():
        opt.zero_grad()
        loss, _ = tr.forward_backward(y, cond, rnd, noise)
        opt.clip_grad_norm_(10.0)
        opt.step(ema_rate=0.999)
        tr.refresh_weights()
        return loss
    for _ in range(2): l = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    for _ in range(n): l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t1 = time.perf_counter()
    tr.forward_backward(y, cond, rnd, noise); torch.cuda.synchronize()
    fb = time.perf_counter() - t1
    gstep = TD.GraphedTrainStep(tr, opt, B, 512, 32, 64, 512)
    for _ in range(2): gstep(y, cond, rnd, noise)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(n): lg, _ = gstep(y, cond, rnd, noise)
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t2) / n
    print(f"B={B:3d}: graphed {dg*1e3:8.1f} ms/step ({B/dg:7.1f} samples/s, {3*132.18*B/dg/1e3:6.1f} TFLOP/s at 3x fwd FLOPs); loss {float(lg):.4f}", flush=True)
    print(f"B={B:3d}: {dt*1e3:8.1f} ms/step ({B/dt:7.1f} samples/s, {3*132.18*B/dt/1e3:6.1f} TFLOP/s at 3x fwd FLOPs); fwd+bwd alone {fb*1e3:7.1f} ms; loss {float(l):.4f}", flush=True)
