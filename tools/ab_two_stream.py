"""Scratch: one NFE of the 24-block denoiser with and without the in-handle two-stream schedule, interleaved in one process."""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, weights, synth
m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0))
m = m.cuda(); h = m._handle(512, 64); h.set_sigmas([1.0])
for B in [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["128"])]:
    x = synth.latents(range(B)).cuda(); cache = h.encode_cond_tokens(synth.cond_tokens(B).cuda())
    line = f"B={B}:"
    for rnd in range(3):
        for mb in (0, 64 if B < 128 else 128):
            h.set_two_stream_min_batch(mb)
            for _ in range(3): h.denoise(x, cache, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): h.denoise(x, cache, 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            line += f"  min{mb}: {dt*1e3:.2f} ms {B/dt:.0f}/s |"
    print(line, flush=True)
