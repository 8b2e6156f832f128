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
    try:
        out.update(_two_streams(dit_handle))
    except Exception as e:
        out["two_streams_error"] = repr(e)
    for name, fn in (("streams", _sampler_streams), ("fp8", _fp8_mode)):
        try:
            out.update(fn())
        except Exception as e:          # secondary numbers never invalidate the headline line
            out[f"{name}_error"] = repr(e)
    return out


def _denoiser(depth=24):
    from . import models_radar_generation as G, weights
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix=""), 0))
    return m.cuda()


def _two_streams(h0, B=64) -> dict:
    """Two independent batches of B on two HIP streams (two handles: one stream at a time per handle).  Every kernel of one
    launch runs the same phase on all CUs at once (MFMA loop, then the HBM-heavy epilogue); a second stream desynchronises
    them.  Reported beside the headline, which stays one batch of 64 on one stream."""
    m1 = _denoiser()
    h1 = m1._handle(512, 64)
    h1.set_sigmas([1.0])
    hs = (h0, h1)
    h0.set_sigmas([1.0])
    x = synth.latents(range(B)).cuda()
    cond = synth.cond_tokens(B).cuda()
    caches = [h.encode_cond_tokens(cond) for h in hs]
    streams = [torch.cuda.Stream() for _ in hs]
    torch.cuda.synchronize()

    def both():
        for h, c, st in zip(hs, caches, streams):
            with torch.cuda.stream(st):
                h.denoise(x, c, 0)
    dt = _time(both, reps=10, warm=3)
    return {"two_streams_B64x2_nfe_ms": dt * 1e3, "two_streams_B64x2_sample_nfe_per_s": 2 * B / dt}


def _sampler_streams() -> dict:
    """18-step sampler, radar cube -> latents (condition encode included): B = 1 frames one after the other (the reference's
    eval loop) against the same frames on concurrent streams (EDMPrecond.sample_concurrent), and two batches of 8."""
    from . import config, models_radar_generation as G, weights
    m = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
    m = m.cuda()
    out = {}
    for B, n in ((1, 4), (8, 2)):
        cubes = [synth.radar_cube(B).cuda() for _ in range(n)]
        seq = _time(lambda: [m.sample(cond=c, cond_type="radar") for c in cubes], reps=2, warm=1)
        con = _time(lambda: m.sample_concurrent(cubes, None, cond_type="radar"), reps=2, warm=1)
        out[f"sample18_B{B}x{n}_sequential_ms"] = seq * 1e3
        out[f"sample18_B{B}x{n}_concurrent_ms"] = con * 1e3
    return out


def _fp8_mode(B=64) -> dict:
    """BASELINE config #5: one NFE with MXFP8 q/k/v projections (qkv_dtype='fp8') and with the GEGLU projection in MXFP8
    as well ('fp8_ff1'), same batch as the headline (which stays bf16)."""
    out = {}
    x = synth.latents(range(B)).cuda()
    cond = synth.cond_tokens(B).cuda()
    for mode, key in (("fp8", "fp8_qkv"), ("fp8_ff1", "fp8_qkv_ff1")):
        m = _denoiser()
        m.qkv_dtype = mode
        h = m._handle(512, 64)
        h.set_sigmas([1.0])
        cache = h.encode_cond_tokens(cond)
        dt = _time(lambda: h.denoise(x, cache, 0), reps=10, warm=3)
        out[f"{key}_nfe_ms_B64"] = dt * 1e3
        out[f"{key}_sample_nfe_per_s"] = B / dt
        del m, h
    return out


def _train_step(B=8) -> dict:
    """one training iteration of EDMPrecond as the reference runs it (engine_generation.py:74-110, radar
    encoder trained jointly): cube -> encoder -> 24-block denoiser -> EDMLoss, backward through all of it, clip +
    fused AdamW/EMA, weight refresh; B = the reference's per-GPU training batch."""
    from . import config, models_radar_generation as G, train_dit as TD, weights
    from .train_utils import FlatAdamW
    m = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
    m = m.cuda()
    opt = FlatAdamW(list(m.parameters()), lr=1e-4, ema=True)
    tr = TD.EdmTrainer(m, opt)
    y, cube = synth.normal([B, 512, 32], 1).cuda(), synth.radar_cube(B).cuda()
    rnd, noise = synth.normal([B], 2), synth.normal([B, 512, 32], 3).cuda()
    dt = _time(lambda: tr.step(y, cube, rnd, noise), reps=3, warm=2)
    return {"train_step_ms_B8": dt * 1e3, "train_samples_per_s_B8": B / dt}


def _edm24():
    from . import config, models_radar_generation as G, weights
    m = G.EDMPrecond(n_latents=512, channels=32, depth=24, configs=config.shipped_generation_config())
    m.load_state_dict(weights.make_state_dict(weights.dit_spec(depth=24), 0), strict=True)
    return m.cuda()


def config4_leg(rank: int, world: int, frames: int = 4) -> dict:
    """BASELINE config #4, the reference's evaluate chain (engine_generation.py:173-300): per frame radar cube ->
    EDMPrecond.sample (18 Heun steps, condition encoded once) -> vae.decode on 1.2 M query points -> refine pass on 500 k
    queries (the latent stack runs once per frame).  Headline of the leg: the reference's own eval_batch_size = 1, `frames`
    frames per rank, no collective.  The same chain at eval batches of 8 and 64 frames (the loop of `evaluate_sharded`; the
    query sets are decoded per frame against the batch's contexts) is reported beside it: the reference's batch size is a
    YAML value, and batching is where the chip fills up."""
    from . import bench_ae, engine_generation as E
    m, vae = _edm24(), bench_ae.build_ae()
    q1, q2 = synth.queries(1, 1200000, seed=11).cuda(), synth.queries(1, 500000, seed=12).cuda()
    cubes = [synth.radar_cube(1, seed=4000 + rank * frames + i).cuda() for i in range(frames)]
    E.sample_and_decode(m, vae, cubes[0], [q1, q2], batch_seeds=torch.tensor([0]))          # warm-up: graph capture, workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, c in enumerate(cubes):
        out = E.sample_and_decode(m, vae, c, [q1, q2], batch_seeds=torch.tensor([rank * frames + i]))
    n_occ = int(out["occupied"][0].sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"seconds": dt, "units_all_ranks": frames * world, "unit": "frames/s", "frames_per_rank": frames, "eval_batch_size": 1,
           "queries_per_frame": 1700000, "ms_per_frame_this_rank": dt / frames * 1e3, "occupied_last_frame": n_occ,
           "workload": "configs[3]: radar cube -> 18-step sample -> decode 1.2 M + 500 k queries, batch-sharded, no collective"}
    # batched evaluation: B frames sampled together, both query sets decoded for the whole batch (latent stack at batch B)
    for B in (8, 64):
        try:
            cube = synth.radar_cube(B, seed=5000 + rank).cuda()
            seeds = torch.arange(rank * B, rank * B + B)

            qb1, qb2 = q1.expand(B, -1, -1), q2.expand(B, -1, -1)

            def batch():
                return E.sample_and_decode(m, vae, cube, [qb1, qb2], batch_seeds=seeds)
            batch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch()
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            res[f"eval_batch_{B}"] = {"ms_per_frame_this_rank": dtb / B * 1e3, "frames_per_s_this_rank": B / dtb}
        except Exception as e:                    # secondary numbers never invalidate the leg
            res[f"eval_batch_{B}"] = {"error": repr(e)}
    return res


def ddp_step_leg(rank: int, world: int, B: int = 8, steps: int = 3) -> dict:
    """One data-parallel training iteration as main_generation.py / engine_generation.py:74-110 run it (DDP over the batch,
    radar encoder trained jointly): forward + backward of EDMLoss through encoder and 24 blocks, bucketed gradient SUM over
    RCCL (train_utils.GradReducer, 1/world folded into the clip), clip_grad_norm_(10), fused AdamW + EMA.  B samples per GPU."""
    from . import train_dit as TD
    from .train_utils import FlatAdamW, GradReducer
    m = _edm24()
    opt = FlatAdamW(list(m.parameters()), lr=1e-4, ema=True)
    red = GradReducer(opt.flat_g) if world > 1 else None
    tr = TD.EdmTrainer(m, opt, reducer=red)
    y, cube = synth.normal([B, 512, 32], 100 + rank).cuda(), synth.radar_cube(B, seed=200 + rank).cuda()
    rnd, noise = synth.normal([B], 300 + rank), synth.normal([B, 512, 32], 400 + rank).cuda()
    tr.step(y, cube, rnd, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, norm = tr.step(y, cube, rnd, noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"seconds": dt, "units_all_ranks": steps * B * world, "unit": "training samples/s", "batch_per_gpu": B, "steps": steps,
            "ms_per_step_this_rank": dt / steps * 1e3, "loss": float(loss), "grad_norm": float(norm),
            "gradient_exchange": "RCCL all-reduce, 64 MiB buckets of the flat fp32 gradient (735 MB)" if world > 1 else "none (1 GPU)"}
