"""Query generation + refine on the device: the numpy code the reference runs on
the host between ``model.sample`` and ``vae.decode`` (engine_generation.py:250-300), with the same
function names and argument meaning as ``utils/utils.py`` (``generate_query_points``, ``norm_points``,
``remove_points_outside_fov``) and ``datasets/utils/query_helper.py`` (``aug_query_helper``), returning
CUDA tensors through ``rald_query_*`` (include/rald_hip.h).

Random numbers: ``rng=None`` replays the reference's draws from numpy's GLOBAL RNG in the reference's
order, so ``np.random.seed(s)`` gives bit-identical queries (the draws are uploaded, the arithmetic runs
on the device); ``rng=torch.Generator(device)`` draws on the device instead (same distribution,
different stream, no host work).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Union

import numpy as np
import torch

from ._handles import _f32c, _need_cuda, _ptr, _stream
from ._lib import check, lib

Rng = Optional[torch.Generator]


def _d(vals: Sequence[float], n: int, what: str):
    if len(vals) != n:
        raise ValueError(f"{what} must have {n} elements")
    return (C.c_double * n)(*[float(v) for v in vals])


def _uniform(shape, device, rng: Rng) -> torch.Tensor:
    """float64 U[0,1): numpy's global stream (reference order) or a device generator."""
    if rng is None:
        return torch.from_numpy(np.random.random_sample(shape)).to(device)
    return torch.rand(shape, dtype=torch.float64, device=device, generator=rng)


def _device(device, rng: Rng) -> torch.device:
    if device is None:
        device = rng.device if rng is not None else "cuda"
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("rald_amd.query_points runs on the HIP device only (no CPU fallback)")
    return device


def generate_query_points(args, coordinate_type: str = "polar", device=None, rng: Rng = None) -> torch.Tensor:
    """utils/utils.py:147-175 -> float32 [num_query_points, 3] on the device (the reference's
    ``.astype('float32')`` of engine_generation.py:259 included)."""
    n = int(args.eval.inference.num_query_points)
    lidar = args.dataset.lidar
    if coordinate_type == "polar":
        pc_range = lidar.pc_range
    elif coordinate_type == "cart":
        pc_range = lidar.pc_range_cart
    else:
        raise ValueError("coordinate_type must be 'polar' or 'cart'")
    return uniform_queries(n, pc_range, lidar.norm_anisotropy, lidar.norm_isotropy, device, rng)


def uniform_queries(n: int, pc_range, norm_anisotropy: bool, norm_isotropy: bool, device=None, rng: Rng = None) -> torch.Tensor:
    device = _device(device, rng)
    if not (norm_anisotropy or norm_isotropy):
        raise ValueError("one of norm_anisotropy / norm_isotropy is required")
    u = _uniform((3, n), device, rng)                          # x draws, then y, then z - numpy's order
    out = torch.empty(n, 3, device=device, dtype=torch.float32)
    check(lib().rald_query_uniform(C.c_void_p(_ptr(u)), n, _d(pc_range, 6, "pc_range"), int(norm_anisotropy), int(norm_isotropy),
                                   C.c_void_p(_ptr(out)), C.c_void_p(_stream())))
    return out


def generate_cart_query_points(args, device=None, rng: Rng = None) -> torch.Tensor:
    """The ``use_cart_query`` branch of engine_generation.py:251-256: uniform in the cartesian box,
    mapped to normalised polar coordinates, FoV-filtered -> float32 [n_kept, 3]."""
    device = _device(device, rng)
    n = int(args.eval.inference.num_query_points)
    lidar = args.dataset.lidar
    if not (lidar.norm_anisotropy or lidar.norm_isotropy):
        raise ValueError("one of norm_anisotropy / norm_isotropy is required")
    u = _uniform((3, n), device, rng)
    out = torch.empty(n, 3, device=device, dtype=torch.float32)
    cnt = torch.zeros(1, device=device, dtype=torch.int64)
    scratch = torch.empty(lib().rald_post_scratch_bytes(n), device=device, dtype=torch.uint8)
    check(lib().rald_query_uniform_cart(C.c_void_p(_ptr(u)), n, _d(lidar.pc_range_cart, 6, "pc_range_cart"), _d(lidar.pc_range, 6, "pc_range"),
                                        int(lidar.norm_anisotropy), int(lidar.norm_isotropy), C.c_void_p(_ptr(out)), C.c_void_p(_ptr(cnt)),
                                        C.c_void_p(_ptr(scratch)), C.c_void_p(_stream())))
    return out[:int(cnt.item())]


def norm_points(points: torch.Tensor, lidar_pc_range, norm_anisotropy: bool, norm_isotropy: bool) -> torch.Tensor:
    """utils/utils.py:77-104 on a float32 CUDA tensor [N,3]."""
    _need_cuda(points, "points")
    points = _f32c(points).reshape(-1, 3)
    out = torch.empty_like(points)
    check(lib().rald_query_norm_points(C.c_void_p(_ptr(points)), points.shape[0], _d(lidar_pc_range, 6, "pc_range"), int(norm_anisotropy),
                                       int(norm_isotropy), C.c_void_p(_ptr(out)), C.c_void_p(_stream())))
    return out


def remove_points_outside_fov(points: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:106-112 (plumbing: a torch mask on the device; the fused form is generate_cart_query_points)."""
    return points[((points > -1) & (points < 1)).all(dim=1)]


def aug_query_helper(helper_points: torch.Tensor, aug_num: int, pc_range, voxel_size, aug_bias_scale: int = 2, rng: Rng = None,
                     norm: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """datasets/utils/query_helper.py:3-42 -> float32 [aug_num, 3].  ``norm=(norm_anisotropy, norm_isotropy)``
    fuses the ``norm_points`` that engine_generation.py:295-296 applies next."""
    _need_cuda(helper_points, "helper_points")
    if helper_points.dim() != 2 or helper_points.shape[1] != 3:
        raise AssertionError("helper_points must be [N,3]")      # the reference asserts
    helper_points = _f32c(helper_points)
    dev = helper_points.device
    N, aug_num = helper_points.shape[0], int(aug_num)
    gen = aug_num - N
    sel = scales = u = None
    if gen > 0:
        if N == 0:
            raise ValueError("a must be greater than 0 unless no samples are taken")   # np.random.choice(0, ...) in the reference
        if rng is None:                                          # the reference's three draws, in its order
            sel = torch.from_numpy(np.random.choice(N, size=gen, replace=True).astype(np.int64)).to(dev)
            scales = torch.from_numpy(np.random.choice(np.arange(aug_bias_scale, step=1) + 1, size=gen).astype(np.int64)).to(dev)
            u = torch.from_numpy(np.random.rand(gen, 3)).to(dev)
        else:
            sel = torch.randint(0, N, (gen,), device=dev, generator=rng, dtype=torch.int64)
            scales = torch.randint(1, int(aug_bias_scale) + 1, (gen,), device=dev, generator=rng, dtype=torch.int64)
            u = torch.rand((gen, 3), dtype=torch.float64, device=dev, generator=rng)
    out = torch.empty(aug_num, 3, device=dev, dtype=torch.float32)
    aniso, iso = (bool(norm[0]), bool(norm[1])) if norm is not None else (False, False)
    p = lambda t: C.c_void_p(_ptr(t) if t is not None else 0)
    check(lib().rald_query_refine(p(helper_points), N, aug_num, p(sel), p(scales), p(u), _d(pc_range, 6, "pc_range"), _d(voxel_size, 3, "voxel_size"),
                                  int(aniso), int(iso), int(norm is not None), p(out), C.c_void_p(_stream())))
    return out


def refine_queries(pred_points: torch.Tensor, args, rng: Rng = None) -> torch.Tensor:
    """engine_generation.py:292-297: jittered copies of the positive (un-normalised polar) points,
    normalised again -> [refine_query_aug_num, 3] ready for ``vae.decode``."""
    inf, lidar = args.eval.inference, args.dataset.lidar
    return aug_query_helper(pred_points, int(inf.refine_query_aug_num), lidar.pc_range, lidar.voxel_size, inf.refine_query_scale, rng,
                            norm=(lidar.norm_anisotropy, lidar.norm_isotropy))
