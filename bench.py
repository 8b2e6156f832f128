#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on MI355X.

metric   : denoising steps/sec (whole node) = sample*NFE per second: one forward of the
           24-block radar-conditioned denoiser for one sample, condition tokens cached
           (SURVEY.md §8d).  One bench "step" = one NFE over a batch of B samples per GPU.
workload : BASELINE.json configs[2] - models_radar_generation DiT denoiser
           (kl_d512_m512_l32_d24_edm) on random 512x32 latents + synthetic radar-spectrum
           condition tokens [B,64,512]; seeded random weights (no checkpoints exist offline).
extras   : the AE half of the metric (configs[1]: encode / decode ms + `roofline_ae`, one object per leg), the config-#4 chain
           (radar cube -> 18-step sample -> decode of 1.2 M + 500 k queries, frames/s) and, at N > 1, one data-parallel
           training step with the RCCL gradient exchange are appended to the same JSON line.
guard    : every RALD_* environment variable is recorded in config.env; the run exits non-zero if the loaded library is a
           PROBE build (the only build with work-skipping switches) or if a probe-only / library-override variable is set.

Multi-GPU (driver launches torchrun, one rank per GPU): samples are independent, so the batch
is sharded across ranks with NO data-path collective ("weak" scaling: B per GPU is fixed);
the only communication is the barrier + MAX-reduce of the elapsed time (RCCL).

  python bench.py --gpus N --steps K --warmup W [--batch B]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_NFE = 132.18          # SURVEY.md §8d / BASELINE.md §3 (multiply-add = 2 FLOP)
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip table)
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E (same table; ~6.3 TB/s is what a streaming copy reaches)


def source_fingerprint():
    """sha256 over the kernel / host sources of the library (rald_amd/csrc/*.hip, *.h, include/*.h): identifies the code a number was
    measured on where no git metadata travels (the GPU box gets a snapshot without .git).  profiles/traffic.json carries the same
    fingerprint of the tree its counters were collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "rald_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "rald_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def git_commit():
    """HEAD of the tree this line was measured on (the GPU box receives a snapshot without .git: the builder writes BUILD_COMMIT
    beside bench.py before a run it wants stamped; otherwise 'unknown')."""
    try:
        import subprocess
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5)
        if out.returncode == 0 and out.stdout.strip():
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=5)
            return out.stdout.strip() + ("+dirty" if dirty.stdout.strip() else "")
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "BUILD_COMMIT")).read().strip()
    except Exception:
        return "unknown"


# variables that only a PROBE build of the library reads (rald_amd/csrc/common.h), plus the library override itself
# RALD_* variables the PRODUCT reads (Python layer); every other RALD_* name is a probe-build switch of the library (RALD_PROBE_ENV in
# rald_amd/csrc, inert in the shipped build) or an override, and a bench line measured with one set is refused
PRODUCT_ENV = ("RALD_DIST_BACKEND", "RALD_BENCH_LEG_TIMEOUT", "RALD_GRAPH", "RALD_GRAPH_MAX_BATCH", "RALD_QKV_DTYPE")


def env_guard():
    """config.env for the JSON line; refuses to measure anything but the shipped library at its shipped settings."""
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("RALD_")}
    bad = [k for k in env if k not in PRODUCT_ENV]
    if env.get("RALD_QKV_DTYPE", "bf16") != "bf16":
        bad.append("RALD_QKV_DTYPE (the headline is the bf16 path; the MXFP8 modes are measured in the extras)")
    if bad:
        print(f"[bench] refusing to run: {bad} are probe-build / override switches; a number measured with them is not the product's",
              file=sys.stderr, flush=True)
        sys.exit(3)
    from rald_amd import _lib
    flags = _lib.lib().rald_build_flags()
    if flags != 0:
        print(f"[bench] refusing to run: {_lib.LIB_PATH} is a PROBE build (rald_build_flags() = {flags})", file=sys.stderr, flush=True)
        sys.exit(3)
    return env


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    dev = local % ndev                      # one rank per GPU; the modulo only matters for 1-GPU rehearsals
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("RALD_DIST_BACKEND", "nccl")      # nccl = RCCL over xGMI; gloo for rehearsals
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(val, world):
    if world == 1:
        return val
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([val], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_denoiser(depth=24):
    from rald_amd import models_radar_generation as G, weights
    m = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=depth)
    sd = weights.make_state_dict(weights.dit_spec(depth=depth, with_radar=False, prefix="model."), seed=0)
    m.load_state_dict({k[len("model."):]: v for k, v in sd.items()})
    return m.cuda(), sd


def cpu_baseline_nfe(sd, depth, seconds_budget=20.0):
    """The oracle (fp32 PyTorch restatement of the reference path, oracle/rald_oracle.py) timed on
    this box's host cores: a bounded sample of the same workload (B=2 NFEs)."""
    from oracle import rald_oracle as O
    from rald_amd import synth
    # the GPU box gives a 1-GPU job a 16-core share; os.cpu_count() reports the whole host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    B = 2
    x = synth.latents(range(B))
    cond = synth.cond_tokens(B)
    t = torch.tensor([0.0])
    with torch.no_grad():
        tw = time.perf_counter()
        O.latent_transformer(sd, x, t, cond, depth=depth)            # warm-up
        tw = time.perf_counter() - tw
        log(f"cpu baseline: warm-up NFE pair took {tw:.2f}s on {cores} threads")
        reps, t0 = 0, time.perf_counter()
        while reps < 2 or (time.perf_counter() - t0 < seconds_budget and reps < 12):
            O.latent_transformer(sd, x, t, cond, depth=depth)
            reps += 1
        dt = time.perf_counter() - t0
    res = {"value": B * reps / dt, "unit": "sample*NFE/s", "cores": cores, "kind": "port",
           "sample": f"{reps} NFEs of the fp32 oracle at B={B} (same weights/shapes), torch CPU {cores} threads"}
    # AE encode / decode on the same host cores (one warm-up + 2 reps each, P = Q = 10 000, B = 1)
    try:
        from rald_amd import weights
        sd_ae = weights.make_state_dict(weights.ae_spec(), 0)
        pc, q, eps = synth.point_cloud(1, 10000), synth.queries(1, 10000), synth.normal([1, 512, 32], 3)
        with torch.no_grad():
            _, z, _, _ = O.ae_encode(sd_ae, pc, eps)
            t0 = time.perf_counter()
            for _ in range(2):
                O.ae_encode(sd_ae, pc, eps)
            res["ae_encode_ms"] = (time.perf_counter() - t0) / 2 * 1e3
            O.ae_decode(sd_ae, z, q, depth=24)
            t0 = time.perf_counter()
            for _ in range(2):
                O.ae_decode(sd_ae, z, q, depth=24)
            res["ae_decode_10k_ms"] = (time.perf_counter() - t0) / 2 * 1e3
    except Exception as e:
        res["ae_error"] = repr(e)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="samples per GPU per NFE")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    env = env_guard()
    world, rank, local = dist_setup(args.gpus)
    from rald_amd import synth
    depth = 24
    log("building denoiser + packing weights")
    m, sd = build_denoiser(depth)
    h = m._handle(512, 64)
    B = args.batch
    # each rank owns its own shard of the global batch: samples rank*B .. rank*B+B-1
    seeds = range(rank * B, rank * B + B)
    x = synth.latents(seeds).cuda()
    cond = synth.cond_tokens(B, seed=777 + rank).cuda()
    cache = h.encode_cond_tokens(cond)
    h.set_sigmas([1.0])
    torch.cuda.synchronize()

    log(f"warm-up {args.warmup} + timed {args.steps} NFEs at B={B}/GPU on {world} GPU(s)")
    for _ in range(args.warmup):
        h.denoise(x, cache, 0)
    barrier(world)
    h.profile_begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # the dominant kernel (FF1) is bracketed by HIP events on every launch of the timed region; the three residual + LayerNorm GEMMs on its
        # first three NFEs only: an event pair costs ~2.5 us of stream time and 96 pairs per NFE took 0.5 ms (2.5 %) off the rate being measured
        if i == 3:
            h.profile_set_kinds(0x1)
        h.denoise(x, cache, 0)
    barrier(world)
    elapsed = time.perf_counter() - t0
    kinds = h.profile_end_kinds()
    ff1_ms, ff1_launches = kinds[0]
    elapsed = max_over_ranks(elapsed, world)

    total_units = world * B * args.steps
    value = total_units / elapsed
    # which of the two bit-identical schedules of an NFE this box runs (rald_amd/_handles.py: timed once per handle and batch size between 128
    # and 255 samples): the whole batch, or two half-batches on two streams - then the timed launches are the first half's, Bt samples each
    tuned = h._two_stream_tuned.get(B)
    split = bool(tuned and tuned[0])
    Bt = ((B // 2 + 32) // 64) * 64 if split else B
    schedule = ("two half-batches (%d + %d samples) on two streams" % (Bt, B - Bt)) if split else "whole batch on one stream"
    if tuned:
        schedule += "; this box: whole %.2f ms, split %.2f ms per NFE" % (tuned[1], tuned[2])
    # dominant kernel: the FF1 GEGLU GEMM [Bt*512, 512] x [512, 4096] (40% of an NFE's FLOPs)
    ff1_flops = 2.0 * (Bt * 512) * 4096 * 512
    avg_s = (ff1_ms / max(ff1_launches, 1)) * 1e-3
    achieved = ff1_flops / avg_s / 1e12 if avg_s > 0 else 0.0
    traffic, traffic_commit, tj = None, None, {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"ff1_geglu_gemm_B{Bt}")
            traffic_commit = tj.get("_source_sha")
        except Exception:
            traffic = None
    M = Bt * 512
    # the three fused residual + LayerNorm GEMMs (x += A.W^T + b; h = AdaLN(x)): largest time share of an NFE.  Algorithmic HBM bytes per
    # launch: A bf16 [M,K] + x fp32 read and written [M,512] + h bf16 [M,512] (+ W, once); FLOPs 2.M.512.K
    def ln_roofline(kind, K, name):
        ms, n = kinds[kind]
        if n == 0:
            return None
        t = ms / n * 1e-3
        flops = 2.0 * M * 512 * K
        byt = M * K * 2 + M * 512 * (4 + 4 + 2) + 512 * K * 2 * (Bt if kind == 2 else 1)     # (kind 2: one folded weight matrix per sample)
        hbm, mf = byt / t / 1e9, flops / t / 1e12
        bound = "hbm" if hbm / PEAK_HBM_GBS > mf / PEAK_BF16_TFLOPS else "mfma"
        return {"kernel": name, "bound": bound, "achieved": hbm if bound == "hbm" else mf, "peak": PEAK_HBM_GBS if bound == "hbm" else PEAK_BF16_TFLOPS,
                "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": (hbm / PEAK_HBM_GBS) if bound == "hbm" else (mf / PEAK_BF16_TFLOPS),
                "tflops": mf, "hbm_gbs": hbm, "avg_launch_us": t * 1e6, "launches_timed": n,
                "algorithmic_bytes_per_launch": byt, "algorithmic_flop_per_launch": flops,
                "traffic": (tj.get("in_situ_bytes_per_launch", {}) or {}).get(f"{name.split(' ')[0]}_B{Bt}")}
    roofline_ln = [r for r in (ln_roofline(1, 512, "gemm_resid_ln_K512_attn1 (to_out + residual + AdaLN)"),
                               ln_roofline(2, 512, "gemm_resid_ln_K512_attn2 (folded cross-attention output + residual + AdaLN, per-sample weights)"),
                               ln_roofline(3, 2048, "gemm_resid_ln_K2048_ff2 (ff.net.2 + residual + AdaLN)")) if r]

    out = {
        "metric": "denoising steps/sec (whole node)", "value": value, "unit": "sample*NFE/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[2]: kl_d512_m512_l32_d24_edm denoiser NFE, 512x32 latents, 64x512 radar condition tokens (cached)",
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "nfe_schedule": schedule, "samples_per_timed_launch": Bt,
                   "weights": "seeded random (rald_amd.weights, seed 0)", "env": env, "commit": git_commit(), "source_sha": source_fingerprint()},
        "whole_path_tflops": value * GFLOP_PER_NFE / 1e3,
        "heun_steps_per_s": value * 18.0 / 35.0, "samples_per_s_18step": value / 35.0,
        "roofline": {"bound": "mfma", "kernel": "rald::gemm_nt_glds_kernel<256,256,4,2,2,EPI_GEGLU> (FF1: [B*512,512]x[512,4096]^T, GEGLU epilogue)",
                     "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_measured_on_source_sha": traffic_commit,
                     "launches_timed": ff1_launches, "avg_launch_us": avg_s * 1e6,
                     "algorithmic_flop_per_launch": ff1_flops},
        "roofline_resid_ln": roofline_ln,
    }
    log(f"GPU: {value:.1f} sample*NFE/s, FF1 kernel {achieved:.0f} TFLOP/s over {ff1_launches} launches")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_nfe(sd, depth)
    elif rank == 0:
        out["cpu_baseline"] = None
    watchdog = None
    if not args.no_extras:
        # every rank runs these two legs on its own shard (frames / training samples are independent; the training step
        # exchanges gradients over RCCL); rank 0 reports the whole-job rates
        from rald_amd import bench_extras
        if world > 1:
            # The headline above is measured; the legs below exchange gradients over RCCL, which no run before the driver's own has
            # exercised on more than one GPU.  If a collective ever hangs there, the headline line must still be printed: a watchdog
            # prints it (legs marked as timed out) and ends the process.
            import threading
            limit = float(os.environ.get("RALD_BENCH_LEG_TIMEOUT", "600"))

            leg_state = {"leg": None, "started": None}

            def _bail():
                # a hung collective: print what was measured, say WHICH leg hung, and fail the run (a hang must not read as a pass)
                if rank == 0:
                    snap = dict(out)
                    snap["secondary_legs"] = {"status": "TIMEOUT", "leg": leg_state["leg"], "limit_s": limit,
                                              "note": "the headline above was measured before the legs started; exit code 4"}
                    print(json.dumps(snap), flush=True)
                print(f"[bench] rank {rank}: leg {leg_state['leg']!r} did not finish within {limit:.0f} s - exiting 4", file=sys.stderr, flush=True)
                os._exit(4)
            watchdog = threading.Timer(limit, _bail)
            watchdog.daemon = True
            watchdog.start()
        for name, fn in (("config4", bench_extras.config4_leg), ("ddp_step", bench_extras.ddp_step_leg)):
            if world > 1:
                leg_state["leg"] = name
            try:
                barrier(world)
                res = fn(rank, world)
                res["seconds"] = max_over_ranks(res["seconds"], world)
                res["value"] = res.pop("units_all_ranks") / res["seconds"]
                out[name] = res
            except Exception as e:      # secondary legs never invalidate the headline line
                out[name] = {"error": repr(e)}
            barrier(world)
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0 and world == 1 and not args.no_extras:
        log("extras: sampler / AE timings")
        try:
            out.update(bench_extras.run(h))
        except Exception as e:  # extras never invalidate the headline line
            out["extras_error"] = repr(e)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
