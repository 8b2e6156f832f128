"""Seeded synthetic inputs of the shapes the reference's data pipeline produces
(SURVEY.md §8d).  All draws come from CPU ``torch.Generator`` streams so the golden script,
the oracle tests, the GPU parity tests and bench.py see identical tensors.
"""
from __future__ import annotations

import torch


def latents(batch_seeds, n_latents: int = 512, channels: int = 32) -> torch.Tensor:
    """Initial sampler noise.  The reference seeds one torch.Generator(device) per sample
    (models_radar_generation.py:297-304, :446-447); device streams are not portable across
    CPU/CUDA/HIP, so 'identical noise seeds' = the CPU generator stream, which is what the
    reference itself produces when run on CPU (SURVEY.md §8b RNG)."""
    outs = []
    for s in batch_seeds:
        g = torch.Generator("cpu").manual_seed(int(s) % (1 << 32))
        outs.append(torch.randn([n_latents, channels], generator=g, dtype=torch.float32))
    return torch.stack(outs)


def radar_cube(batch: int, seed: int = 1234, rae=(128, 64, 32)) -> torch.Tensor:
    """[B,R,A,E,2] in U[0,1): real cubes are clipped to [0,45] dB and divided by 45
    (datasets/aligned_coloradar/Coloradar_dataset.py:447-451)."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.rand([batch, *rae, 2], generator=g, dtype=torch.float32)


def point_cloud(batch: int, n_points: int = 10000, seed: int = 2024) -> torch.Tensor:
    """[B,P,3] in U(-1,1)^3: real clouds are polar view-cone coordinates normalised per axis
    to [-1,1] (Coloradar_dataset.py:376-379)."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.rand([batch, n_points, 3], generator=g, dtype=torch.float32) * 2 - 1


def queries(batch: int, n_queries: int, seed: int = 4242) -> torch.Tensor:
    """Decoder query points, U(-1,1)^3 (utils/utils.py:171-175)."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.rand([batch, n_queries, 3], generator=g, dtype=torch.float32) * 2 - 1


def cond_tokens(batch: int, n_tokens: int = 64, dim: int = 512, seed: int = 777) -> torch.Tensor:
    """Stand-in radar condition tokens [B,64,C] for denoiser-only workloads."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn([batch, n_tokens, dim], generator=g, dtype=torch.float32)


def normal(shape, seed: int) -> torch.Tensor:
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(list(shape), generator=g, dtype=torch.float32)
