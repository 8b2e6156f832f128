"""Scratch: does running two half-batches on two HIP streams (two handles) beat one full batch?  The kernels of one
launch run the same phase on every CU at once (MFMA loop, then HBM-heavy epilogue); two streams desynchronise them."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rald_amd import models_radar_generation as G, weights, synth

def make():
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=24)
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24, with_radar=False, prefix=""), 0))
    m = m.cuda(); h = m._handle(512, 64); h.set_sigmas([1.0])
    return m, h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
models = [make() for _ in range(NS)]
x = synth.latents(range(B)).cuda(); tok = synth.cond_tokens(B).cuda()
def bench(parts):
    hs = [models[i][1] for i in range(parts)]
    per = B // parts
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(parts)]
    caches = [hs[i].encode_cond_tokens(tok[i * per:(i + 1) * per].contiguous()) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    hs[i].denoise(xs[i], caches[i], 0)
    run(3); torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B} as {parts} stream(s) x {per}: {dt*1e3:7.3f} ms per {B}-sample NFE  {B/dt:8.1f} sample-NFE/s", flush=True)
for parts in (1, NS, 1, NS):
    bench(parts)
