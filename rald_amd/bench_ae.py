"""AE encode / decode latency (BASELINE.json configs[1]: full VecSet AE, kl_d512_m512_l32_mix,
P = 10 000 points; decode at Q = 10 000 and 1 200 000 queries) on synthetic view-cone clouds."""
from __future__ import annotations

import time

import torch

from . import models_ae as A
from . import synth, weights


def _time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def build_ae():
    m = A.kl_d512_m512_l32_mix(N=10000)
    m.load_state_dict(weights.make_state_dict(weights.ae_spec(), 0))
    return m.cuda()


def run(batches=(1, 8)) -> dict:
    out = {}
    m = build_ae()
    h = m._handle()
    for B in batches:
        pc = synth.point_cloud(B, 10000).cuda()
        eps = synth.normal([B, 512, 32], 3).cuda()
        out[f"ae_encode_ms_B{B}"] = _time(lambda: h.encode(pc, eps))
        z = h.encode(pc, eps)[1]
        out[f"ae_decode_latents_ms_B{B}"] = _time(lambda: h.decode_latents(z))
        ctx = h.decode_latents(z)
        q10k = synth.queries(B, 10000).cuda()
        out[f"ae_decode_queries_10k_ms_B{B}"] = _time(lambda: h.decode_queries(ctx, q10k))
        # what the reference's decode(z, queries) costs as one call (stack + queries), Q = 10 000
        out[f"ae_decode_10k_ms_B{B}"] = out[f"ae_decode_latents_ms_B{B}"] + out[f"ae_decode_queries_10k_ms_B{B}"]
    B = 1
    z = h.encode(synth.point_cloud(1, 10000).cuda(), synth.normal([1, 512, 32], 3).cuda())[1]
    ctx = h.decode_latents(z)
    q = synth.queries(1, 1200000).cuda()
    ms = _time(lambda: h.decode_queries(ctx, q), reps=3, warm=1)
    out["ae_decode_queries_1200k_ms_B1"] = ms
    out["ae_decode_queries_Mq_per_s"] = 1.2e6 / ms / 1e3
    # decode post-processing on the device: compaction + inverse-norm + polar->cartesian of the
    # 1.2 M logits, then Chamfer of the positives against a 10 000-point surface
    from . import postprocess as PP
    pc_range = [0, -90, -20, 15.8, 90, 20]
    logits = h.decode_queries(ctx, q)[0]
    qq = q[0]
    out["post_occupied_points_1200k_ms"] = _time(lambda: PP.occupied_points(logits, qq, pc_range, True, False, True), reps=5)
    pts = PP.occupied_points(logits, qq, pc_range, True, False, True)
    if pts.shape[0] == 0:                      # random weights may put every logit on one side: use a fixed 5 % subset
        pts = PP.occupied_points(synth.normal([1200000], 5).cuda() - 1.645, qq, pc_range, True, False, True)
    gt = PP.polar2cartesian(PP.inverse_norm_points(synth.point_cloud(1, 10000, seed=9)[0].cuda(), pc_range, True, False))
    out["post_chamfer_n_pred"] = int(pts.shape[0])
    out["post_chamfer_ms"] = _time(lambda: PP.cal_metrics(pts, gt), reps=3)
    out["roofline_ae"] = rooflines(out)
    return out


PEAK_BF16_TFLOPS = 2500.0                   # dense bf16 MFMA (MI355X_MICROARCH.md, chip table)
PEAK_HBM_GBPS = 8000.0
# v_exp_f32 issues in 8 cycles per wave64 instruction (MI355X_MICROARCH.md, per-instruction cycle constants): 8 lanes/clk/SIMD
PEAK_EXP_PER_S = 256 * 4 * 8 * 2.4e9


def rooflines(t: dict) -> dict:
    """One object per AE leg (the 'AE enc/dec ms' half of BASELINE.json's metric).  Algorithmic work per unit is SURVEY.md
    section 8d's: encode 47.06 GFLOP per cloud of 10 000 points, latent stack 116.52 GFLOP per latent set - both dense
    contractions, priced against the bf16 MFMA peak - and the query decoder, whose folded form (rald_amd/csrc/ae_decode.hip)
    executes 512 exponentials + 0.07 MFLOP per query and moves 16 B: priced against the v_exp_f32 issue rate, with its HBM
    fraction and the reference-equivalent rate (2.15 MFLOP per query as the reference computes it) beside it."""
    r = {}
    # executed work beside the reference-form work: the folded encoder executes 21.0 of the reference's 47.06
    # GFLOP per cloud (tools: DESIGN.md section 4, folded encoder) - `frac` prices the reference's FLOPs over the measured time (what a
    # user of the reference gets), `executed_frac` the FLOPs the kernels really perform; the latent stack executes what the reference does
    for B in (1, 8):
        for leg, key, gflop, executed in (("encode", f"ae_encode_ms_B{B}", 47.06, 21.0), ("decode_latents", f"ae_decode_latents_ms_B{B}", 116.52, 116.52)):
            ach = gflop * B / t[key]                                    # GFLOP / ms = TFLOP/s
            ex = executed * B / t[key]
            r[f"{leg}_B{B}"] = {"bound": "mfma", "ms": t[key], "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / PEAK_BF16_TFLOPS, "algorithmic_gflop_per_launch": gflop * B,
                                "executed_gflop_per_launch": executed * B, "executed_tflops": ex, "executed_frac": ex / PEAK_BF16_TFLOPS}
    ms = t["ae_decode_queries_1200k_ms_B1"]
    exps = 512 * 1.2e6 / (ms * 1e-3)
    r["decode_queries_1200k_B1"] = {"bound": "valu", "kernel": "rald::ae_decode_stream_kernel (one launch)", "ms": ms, "achieved": exps / 1e12,
                                    "peak": PEAK_EXP_PER_S / 1e12, "unit": "T exp/s", "frac": exps / PEAK_EXP_PER_S,
                                    "algorithmic_bytes_per_launch": 16 * 1.2e6, "hbm_gbps": 16 * 1.2e6 / (ms * 1e-3) / 1e9,
                                    "hbm_frac": 16 * 1.2e6 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                                    "reference_equivalent_tflops": 2.150e6 * 1.2e6 / (ms * 1e-3) / 1e12,
                                    "executed_tflops": 0.59e6 * 1.2e6 / (ms * 1e-3) / 1e12,      # 0.59 MFLOP per query after folding
                                    "traffic": _traffic("ae_decode_queries_1200k")}
    return r


def _traffic(key):
    import json, os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        return json.load(open(p)).get(key)
    except Exception:
        return None
