"""The ~15 lines of the reference's generation engine that call the hot path
(engine_generation.py:183-232, :274-300), restated as a batch-sharded driver: radar cube ->
EDMPrecond.sample -> vae.decode on query sets -> occupancy = logits > 0, plus the per-frame
inference tail (:250-322: query generation, helper points, refine pass, Chamfer) kept on the device.
Data loading and PLY writing are out of scope."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch

from . import distributed as D
from . import postprocess as PP
from . import query_points as QP


@torch.no_grad()
def sample_and_decode(model, vae, radar_cube: torch.Tensor, query_sets: Sequence[torch.Tensor],
                      batch_seeds: Optional[torch.Tensor] = None) -> Dict[str, object]:
    """One evaluation batch on ONE rank: `model.sample` (18 Heun steps, condition encoded once),
    then every query set decoded against the same latents (the 24-layer latent stack runs once:
    rald_amd.models_ae memoises the decoder context per latent tensor).
    Returns {'latents': [B,512,C], 'logits': [ [B,Q_i] ... ], 'occupied': [bool masks]}."""
    sampled = model.sample(cond=radar_cube, batch_seeds=batch_seeds, cond_type='radar')        # :195
    logits = [vae.decode(sampled, q).squeeze(-1) for q in query_sets]                          # :204, :275, :300
    return {"latents": sampled, "logits": logits, "occupied": [l > 0 for l in logits]}        # :229-232


@torch.no_grad()
def chamfer_of_decode(logits: torch.Tensor, queries: torch.Tensor, surface: torch.Tensor, lidar_pc_range,
                      norm_anisotropy: bool = True, norm_isotropy: bool = False, view_cone_mode: bool = True) -> float:
    """engine_generation.py:283-322 for one sample, on the device: positives of `logits` [Q] ->
    metric (cartesian) coordinates, ground-truth `surface` [P,3] likewise, Chamfer distance."""
    pred = PP.occupied_points(logits, queries, lidar_pc_range, norm_anisotropy, norm_isotropy, view_cone_mode)
    gt = PP.inverse_norm_points(surface, lidar_pc_range, norm_anisotropy, norm_isotropy)
    if view_cone_mode:
        gt = PP.polar2cartesian(gt)
    return PP.cal_metrics(pred, gt)


def _get(ns, name, default=None):
    return ns.get(name, default) if hasattr(ns, "get") else getattr(ns, name, default)


@torch.no_grad()
def infer_point_cloud(vae, sampled_tokens: torch.Tensor, args, helper_points: Optional[torch.Tensor] = None,
                      surface: Optional[torch.Tensor] = None, rng: Optional[torch.Generator] = None) -> Dict[str, object]:
    """engine_generation.py:250-322 for ONE sample (the reference asserts batch 1 when helper points are
    used), everything between the sampler and the metric on the device:
    uniform (or cartesian-box) queries [+ helper points] -> vae.decode -> positives -> un-normalised polar
    points -> [refine: jittered copies -> normalise -> decode -> positives] -> cartesian if view_cone_mode
    -> Chamfer distance against `surface` (normalised ground truth [P,3]) unless skip_eval_metric.
    `rng=None` consumes numpy's global RNG in the reference's order; a device generator avoids host draws.
    Returns {'pred': [n,3] metric coordinates, 'cd': float or None, 'n_queries': int}."""
    if sampled_tokens.shape[0] != 1:
        raise AssertionError("Batch size should be 1 when using query helper points")          # :265
    lidar, inf = args.dataset.lidar, args.eval.inference
    aniso, iso = lidar.norm_anisotropy, lidar.norm_isotropy
    dev = sampled_tokens.device
    if _get(args.eval, "use_cart_query", False):
        grid = QP.generate_cart_query_points(args, device=dev, rng=rng)                        # :251-256
    else:
        grid = QP.generate_query_points(args, device=dev, rng=rng)                             # :258
    if _get(inf, "query_helper", False) and helper_points is not None:
        grid = torch.cat((grid, helper_points.to(dev, torch.float32).reshape(-1, 3)), dim=0)    # :264-271
    output = vae.decode(sampled_tokens, grid[None]).squeeze(-1)[0]                              # :275
    pred = PP.occupied_points(output, grid, lidar.pc_range, aniso, iso, view_cone_mode=False)   # :283-289
    n_queries = grid.shape[0]
    if _get(inf, "refine_query", False):
        refined = QP.refine_queries(pred, args, rng=rng)                                        # :292-297
        out_r = vae.decode(sampled_tokens, refined[None]).squeeze(-1)[0]                        # :300
        pred = PP.occupied_points(out_r, refined, lidar.pc_range, aniso, iso, view_cone_mode=False)   # :304-310
        n_queries += refined.shape[0]
    view_cone = bool(_get(lidar, "view_cone_mode", False))
    if view_cone:
        pred = PP.polar2cartesian(pred)                                                         # :313-315
    cd = None
    if surface is not None and not _get(args.eval, "skip_eval_metric", False):
        gt = PP.inverse_norm_points(surface.to(dev), lidar.pc_range, aniso, iso)                 # :290
        cd = PP.cal_metrics(pred, PP.polar2cartesian(gt) if view_cone else gt)                   # :320
    return {"pred": pred, "cd": cd, "n_queries": n_queries}


@torch.no_grad()
def evaluate_sharded(model, vae, cubes: torch.Tensor, queries: torch.Tensor, eval_batch_size: int = 1,
                     metric_fn: Optional[Callable[[torch.Tensor, int], float]] = None) -> Dict[str, float]:
    """Batch-sharded evaluation: every rank owns the samples DistributedSampler would give it,
    runs them in eval batches with seeds = global sample index, and the only collective is the
    final metric reduction (utils/misc.py:45-47).  `cubes` [N,R,A,E,2] and `queries` [N,Q,3] are the
    full (host) arrays; each rank moves only its shard to the device."""
    rank = torch.distributed.get_rank() if D.is_dist() else 0
    world = D.world_size()
    mine = D.shard_sample_indices(cubes.shape[0], rank, world)
    total, count = 0.0, 0.0
    occupied_fraction = []
    dev = next(model.parameters()).device
    for i in range(0, len(mine), eval_batch_size):
        idx = mine[i:i + eval_batch_size]
        out = sample_and_decode(model, vae, cubes[idx].to(dev), [queries[idx].to(dev)],
                                batch_seeds=torch.tensor(idx))
        occ = out["occupied"][0].float().mean(dim=1)
        for j, gi in enumerate(idx):
            val = metric_fn(out["logits"][0][j], gi) if metric_fn else float(occ[j])
            total += val
            count += 1
            occupied_fraction.append(float(occ[j]))
    total, count = D.reduce_sum_count(total, count)
    return {"metric_mean": total / max(count, 1.0), "n_samples": count}
