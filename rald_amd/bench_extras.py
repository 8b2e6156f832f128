"""Secondary measurements bench.py appends to its JSON line at N=1: the 18-step Heun sampler
(35 NFE, condition cached) and - once built - AE encode/decode latency (BASELINE configs[1])."""
from __future__ import annotations

import time

import torch

from . import synth


def _time(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(dit_handle) -> dict:
    out = {}
    B = 8
    lat = synth.latents(range(B)).cuda()
    cache = dit_handle.encode_cond_tokens(synth.cond_tokens(B).cuda())
    dt = _time(lambda: dit_handle.sample(lat, cache, 18), reps=2)
    out["sampler18_B8_s"] = dt
    out["sampler18_B8_samples_per_s"] = B / dt
    try:
        from . import bench_ae
        out.update(bench_ae.run())
    except ImportError:
        pass
    return out
