"""Device counterparts of the host post-processing the reference runs after ``vae.decode``
(engine_generation.py:229-243, :283-322): same function names and argument meaning as
``utils/utils.py`` (``inverse_norm_points``, ``cal_metrics``) and
``dataset_preprocessor/lidar.py`` (``polar2cartesian``), operating on CUDA tensors through
``rald_post_*`` (include/rald_hip.h).  ``occupied_points`` is the fused form of
``np.where(output > 0)`` -> ``grid[ind]`` -> ``inverse_norm_points`` -> ``polar2cartesian``.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch

from ._handles import _f32c, _need_cuda, _ptr, _stream
from ._lib import check, lib


def _range(pc_range: Sequence[float]):
    if len(pc_range) != 6:
        raise ValueError("pc_range must have 6 elements [min0,min1,min2,max0,max1,max2]")
    return (C.c_double * 6)(*[float(v) for v in pc_range])      # Python floats in the reference's YAML -> doubles


def occupied_points(logits: torch.Tensor, queries: torch.Tensor, lidar_pc_range, norm_anisotropy: bool, norm_isotropy: bool,
                    view_cone_mode: bool = True, threshold: float = 0.0, return_index: bool = False):
    """logits [Q], queries [Q,3] (normalised) -> positive queries in metric (cartesian if view_cone_mode)
    coordinates, in ascending query order: [n_pos, 3] (and their indices)."""
    _need_cuda(logits, "logits")
    logits, queries = _f32c(logits).reshape(-1), _f32c(queries).reshape(-1, 3)
    Q = logits.numel()
    if queries.shape[0] != Q:
        raise ValueError("one logit per query expected")
    pts = torch.empty(Q, 3, device=logits.device, dtype=torch.float32)
    idx = torch.empty(Q, device=logits.device, dtype=torch.int64) if return_index else None
    cnt = torch.zeros(1, device=logits.device, dtype=torch.int64)
    scratch = torch.empty(lib().rald_post_scratch_bytes(Q), device=logits.device, dtype=torch.uint8)
    check(lib().rald_post_occupied_points(C.c_void_p(_ptr(logits)), C.c_void_p(_ptr(queries)), Q, _range(lidar_pc_range),
                                          int(norm_anisotropy), int(norm_isotropy), int(view_cone_mode), float(threshold),
                                          C.c_void_p(_ptr(pts)), C.c_void_p(_ptr(idx) if return_index else 0), C.c_void_p(_ptr(cnt)),
                                          C.c_void_p(_ptr(scratch)), C.c_void_p(_stream())))
    n = int(cnt.item())                                   # the only host sync: the reference syncs on the full D2H here
    return (pts[:n], idx[:n]) if return_index else pts[:n]


def _transform(points: torch.Tensor, lidar_pc_range, aniso: bool, iso: bool, view_cone: bool) -> torch.Tensor:
    _need_cuda(points, "points")
    points = _f32c(points).reshape(-1, 3)
    out = torch.empty_like(points)
    if points.shape[0]:
        check(lib().rald_post_transform_points(C.c_void_p(_ptr(points)), points.shape[0], _range(lidar_pc_range), int(aniso), int(iso),
                                               int(view_cone), C.c_void_p(_ptr(out)), C.c_void_p(_stream())))
    return out


def inverse_norm_points(points, lidar_pc_range, norm_anisotropy, norm_isotropy):
    """utils/utils.py:50-75."""
    return _transform(points, lidar_pc_range, norm_anisotropy, norm_isotropy, False)


def polar2cartesian(points):
    """dataset_preprocessor/lidar.py:57-63 ((r, az deg, el deg) -> (x, y, z))."""
    # identity normalisation: scale 1, offset 0 on every axis
    return _transform(points, [-1, -1, -1, 1, 1, 1], True, False, True)


def cal_metrics(y_pred: torch.Tensor, y_gt: torch.Tensor) -> float:
    """utils/utils.py:116-142 - Chamfer distance; inf for an empty prediction, like the reference."""
    if y_pred.shape[0] == 0:
        return float("inf")
    _need_cuda(y_pred, "y_pred")
    y_pred, y_gt = _f32c(y_pred).reshape(-1, 3), _f32c(y_gt).reshape(-1, 3).to(y_pred.device)
    sums = torch.zeros(2, device=y_pred.device, dtype=torch.float64)
    check(lib().rald_post_chamfer_sums(C.c_void_p(_ptr(y_pred)), y_pred.shape[0], C.c_void_p(_ptr(y_gt)), y_gt.shape[0],
                                       C.c_void_p(_ptr(sums)), C.c_void_p(_stream())))
    s = sums.cpu()
    return float(0.5 * s[1] / y_gt.shape[0] + 0.5 * s[0] / y_pred.shape[0])


def accuracy_iou(outputs: torch.Tensor, labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """engine_generation.py:229-243 per sample: (accuracy [B], iou [B]); the caller takes .mean()."""
    _need_cuda(outputs, "outputs")
    outputs, labels = _f32c(outputs), _f32c(labels).to(outputs.device)
    B, Q = outputs.shape
    acc = torch.empty(B, device=outputs.device, dtype=torch.float32)
    iou = torch.empty(B, device=outputs.device, dtype=torch.float32)
    check(lib().rald_post_iou(C.c_void_p(_ptr(outputs)), C.c_void_p(_ptr(labels)), B, Q, C.c_void_p(_ptr(acc)), C.c_void_p(_ptr(iou)),
                              C.c_void_p(_stream())))
    return acc, iou
