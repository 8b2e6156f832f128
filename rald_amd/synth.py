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


def structured_cloud(batch: int, n_points: int = 10000, seed: int = 3031) -> torch.Tensor:
    """[B,P,3] cloud with the structure real frustum scans have and U(-1,1)^3 lacks (stress input of the folded AE kernels,
    tests/golden/make_golden.py G18): 40 % of the points on three planes (x = 0.25, y = -0.5 and the z = +1 face), 10 % on
    the +-1 faces of the cube (one coordinate exactly +-1), 30 % uniform, then 20 % EXACT duplicates of earlier points."""
    g = torch.Generator("cpu").manual_seed(seed)
    out = []
    for _ in range(batch):
        n_dup = n_points // 5
        n_base = n_points - n_dup
        p = torch.rand([n_base, 3], generator=g, dtype=torch.float32) * 2 - 1
        n_plane, n_face = (2 * n_points) // 5, n_points // 10
        third = n_plane // 3
        p[:third, 0] = 0.25
        p[third:2 * third, 1] = -0.5
        p[2 * third:n_plane, 2] = 1.0
        axis = torch.randint(0, 3, [n_face], generator=g)
        sign = torch.randint(0, 2, [n_face], generator=g).float() * 2 - 1
        p[n_plane + torch.arange(n_face), axis] = sign
        src = torch.randint(0, n_base, [n_dup], generator=g)
        full = torch.cat([p, p[src]])
        out.append(full[torch.randperm(n_points, generator=g)])
    return torch.stack(out)


def structured_queries(batch: int, n_queries: int, seed: int = 3032) -> torch.Tensor:
    """[B,Q,3] decoder queries: a quarter on the +-1 faces, a quarter on the planes of `structured_cloud`, half uniform."""
    g = torch.Generator("cpu").manual_seed(seed)
    q = torch.rand([batch, n_queries, 3], generator=g, dtype=torch.float32) * 2 - 1
    n4 = n_queries // 4
    axis = torch.randint(0, 3, [batch, n4], generator=g)
    sign = torch.randint(0, 2, [batch, n4], generator=g).float() * 2 - 1
    for b in range(batch):
        q[b, torch.arange(n4), axis[b]] = sign[b]
        q[b, n4:n4 + n4 // 3, 0] = 0.25
        q[b, n4 + n4 // 3:n4 + 2 * (n4 // 3), 1] = -0.5
        q[b, n4 + 2 * (n4 // 3):2 * n4, 2] = 1.0
    return q
