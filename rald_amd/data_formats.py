"""The on-disk formats and the input contract either side of the hot path, so
that the build consumes the reference's artefacts:

  * radar cube ``.bin``: float32 ``[R, A, E, C_raw]`` (``load_radarcube``, Coloradar_dataset.py:420-430)
    and its normalisation + bilinear up-sampling to the network input ``[R, 64, 32, 2]``
    (``process_radar_data``, :432-475) - on the device through ``rald_radar_cube_prepare``;
  * latent cache ``.npz`` with ``res_tokens [M, latent_dim]`` float32 (written by
    engine_generation.cache_latents :409, read by ``load_cached_latent`` :477-483);
  * predicted latents ``.pt`` (``load_pred_latent`` :485-491, ``torch.save`` at engine_generation.py:222);
  * checkpoints ``checkpoint-<epoch>.pth`` = ``{model, model_ema, optimizer, epoch, scaler, args}``
    (utils/misc.py:293-318) and the EMA list ordered by ``named_parameters()`` (:360-363).
Only loaders that execute nothing from the file are used (``np.load(allow_pickle=False)``,
``torch.load(weights_only=True)``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

from ._handles import _f32c, _ptr, _stream
from ._lib import check, lib


# ---- radar cube -------------------------------------------------------------------------------
def load_radarcube(path, input_r_dim: int = 128, input_a_dim: int = 8, input_e_dim: int = 2) -> np.ndarray:
    cube = np.fromfile(str(path), dtype=np.float32)
    return cube.reshape(input_r_dim, input_a_dim, input_e_dim, -1)


def process_radar_data(radar_cube, *, norm_intensity: bool = True, max_intensity: float = 45.0, norm_dopp: bool = True,
                       max_dopp: float = 2.4958, upsample: bool = True, tgt_a_dim: int = 64, tgt_e_dim: int = 32,
                       device="cuda") -> torch.Tensor:
    """raw cube [R,A,E,C_raw] or [B,R,A,E,C_raw] (numpy or tensor) -> [(..B), R, tgt_A, tgt_E, 2] on the device.
    Defaults are the shipped radar config (configs/generation/*_eval.yml:70-81)."""
    t = torch.as_tensor(radar_cube, dtype=torch.float32)
    squeeze = t.dim() == 4
    if squeeze:
        t = t.unsqueeze(0)
    t = _f32c(t.to(device))
    if not t.is_cuda:
        raise RuntimeError("process_radar_data runs on the GPU; rald_amd has no CPU path")
    B, R, A, E, Cr = t.shape
    tA, tE = (tgt_a_dim, tgt_e_dim) if upsample else (A, E)
    out = torch.empty(B, R, tA, tE, 2, device=t.device, dtype=torch.float32)
    check(lib().rald_radar_cube_prepare(C.c_void_p(_ptr(t)), B, R, A, E, Cr, tA, tE, int(norm_intensity), float(max_intensity),
                                        int(norm_dopp), float(max_dopp), C.c_void_p(_ptr(out)), C.c_void_p(_stream())))
    return out[0] if squeeze else out


# ---- latent cache / predicted latents --------------------------------------------------------
def save_latent_cache(path, res_tokens: torch.Tensor) -> None:
    """np.savez(path, res_tokens=[M, latent_dim] float32)  (engine_generation.py:409)."""
    np.savez(str(path), res_tokens=res_tokens.detach().to(torch.float32).cpu().numpy())


def load_cached_latent(path) -> torch.Tensor:
    with np.load(str(path), allow_pickle=False) as f:
        return torch.from_numpy(f["res_tokens"])


def load_pred_latent(path) -> torch.Tensor:
    return torch.load(str(path), weights_only=True)


# ---- checkpoints --------------------------------------------------------------------------------
def load_checkpoint(path, model: torch.nn.Module, ema: bool = False, device=None) -> Tuple[Optional[list], Optional[List[torch.Tensor]], dict]:
    """utils/misc.py:324-365 for the eval path: strict load of checkpoint['model'];
    with ema=True also returns (model_params, ema_params) - the EMA list ordered by named_parameters()."""
    ckpt = torch.load(str(path), map_location="cpu", weights_only=True)
    model.load_state_dict(ckpt["model"], strict=True)
    if not ema:
        return None, None, ckpt
    ema_sd = ckpt["model_ema"]
    ema_params = [ema_sd[name].to(device) if device is not None else ema_sd[name] for name, _ in model.named_parameters()]
    return list(model.parameters()), ema_params, ckpt


def save_checkpoint(path, model: torch.nn.Module, ema_params: Optional[List[torch.Tensor]] = None, epoch: int = 0) -> None:
    """The tensor part of utils/misc.py:293-318 (optimizer / scaler / args belong to the training loop)."""
    ema_sd = None
    if ema_params is not None:
        ema_sd = {k: v.clone() for k, v in model.state_dict().items()}
        for i, (name, _) in enumerate(model.named_parameters()):
            ema_sd[name] = ema_params[i].detach().cpu()
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "model_ema": ema_sd, "epoch": epoch}, str(path))
