"""Scratch: eval-style batch-1 latencies, eager launches vs hipGraph replay."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, models_ae as A, weights, synth, config
m = G.kl_d512_m512_l32_d24_edm(configs=config.shipped_generation_config())
m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0)); m = m.cuda()
h = m._handle()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for B in (1, 2, 4, 8):
    cube = synth.radar_cube(B).cuda()
    print(f"B={B} encode_cond (radar enc + cond cache): {t(lambda: h.encode_cond(cube)):.2f} ms", flush=True)
    _, cache = h.encode_cond(cube)
    lat = synth.latents(range(B)).cuda()
    e = t(lambda: h.sample(lat, cache, 18, use_graph=False)); g = t(lambda: h.sample(lat, cache, 18, use_graph=True))
    print(f"B={B} 18-step sampler: eager {e:.1f} ms  graph {g:.1f} ms  -> {B/g*1e3:.1f} samples/s, {35*B/g*1e3:.0f} NFE/s", flush=True)
ae = A.kl_d512_m512_l32_mix(N=10000); ae.load_state_dict(weights.make_state_dict(weights.ae_spec(), 0)); ha = ae.cuda()._handle()
z = synth.normal([1, 512, 32], 1).cuda()
print(f"AE decode_latents B=1: eager {t(lambda: ha.decode_latents(z, use_graph=False), 5):.2f} ms  graph {t(lambda: ha.decode_latents(z, use_graph=True), 5):.2f} ms")
