"""Drop-in for the reference's ``model/models_ae.py``: same factory names (``kl_d512_m512_l32_mix``
..., looked up via ``models_ae.__dict__[name](N=...)``, main_generation.py:110), same
``KLAutoEncoder.encode / decode / forward`` signatures and return values (:351-432) and the same
``state_dict`` keys.  Arithmetic runs in librald_hip.so (include/rald_hip.h); no PyTorch compute
path exists.

Scope: query_type='mix' (the shipped config, configs/ae/*cone.yml:85) and 'learnable' (:325-326,
:378-379); query_type='point' needs torch_cluster.fps (a CUDA extension that is neither vendored
nor installed; SURVEY.md §8c 'parity unpinned') and raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import weights as _w
from ._handles import AeHandle
from ._lib import AeConfig
from .models_radar_generation import _HipBacked, build_param_tree


class DiagonalGaussianDistribution(object):
    """models_ae.py:141-179 (moments only; sampling happens inside rald_ae_encode)."""

    def __init__(self, mean, logvar, deterministic=False):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def mode(self):
        return self.mean


class KLAutoEncoder(_HipBacked):
    def __init__(self, *, depth=24, dim=512, queries_dim=512, output_dim=1, num_inputs=2048, num_latents=512,
                 latent_dim=64, heads=8, dim_head=64, weight_tie_layers=False, decoder_ff=False, query_type='point'):
        super().__init__()
        if query_type not in ('mix', 'learnable'):
            raise NotImplementedError(f"query_type={query_type!r}: 'mix' (the shipped config) and 'learnable' are built; "
                                      "'point' needs torch_cluster.fps")
        if weight_tie_layers or decoder_ff or output_dim != 1 or queries_dim != dim:
            raise NotImplementedError("only the create_autoencoder() configuration is built (models_ae.py:447-458)")
        self.depth, self.num_inputs, self.num_latents = depth, num_inputs, num_latents
        self.dim, self.latent_dim, self.heads, self.dim_head = dim, latent_dim, heads, dim_head
        self.query_type = query_type
        spec = _w.ae_spec(dim=dim, num_latents=num_latents, latent_dim=latent_dim, depth=depth, heads=heads,
                          dim_head=dim_head, query_type=query_type)
        build_param_tree(self, spec, buffers=("point_embed.basis",))
        self._hip = None
        self._hip_fp = None
        self._ctx_memo = None

    def _handle(self) -> AeHandle:
        fp = self._state_fingerprint()
        if self._hip is None or self._hip_fp != fp:
            cfg = AeConfig(dim=self.dim, num_latents=self.num_latents, latent_dim=self.latent_dim, depth=self.depth,
                           heads=self.heads, dim_head=self.dim_head, num_inputs=self.num_inputs,
                           query_type={'mix': 0, 'learnable': 1}[self.query_type])
            h = AeHandle(cfg)
            h.load(self.state_dict().items())
            self._hip, self._hip_fp, self._ctx_memo = h, fp, None
        return self._hip

    def encode(self, pc):
        """pc [B,N,3] -> (kl [B], z [B,M,latent_dim]); posterior noise from torch.randn on the CPU
        global RNG, exactly where the reference draws it (:153)."""
        B, N, D = pc.shape
        assert N == self.num_inputs
        eps = torch.randn(B, self.num_latents, self.latent_dim)
        kl, z = self._handle().encode(pc, eps)
        return kl, z

    def _context(self, x):
        """Latent stack + decoder context for latents x; memoised on the tensor OBJECT (+ its version; the entry keeps
        the tensor alive, see _HipBacked._memo_hit), so the reference's pattern of several decode() calls on the same
        `sampled_tokens` (engine_generation.py:204, :275, :300) runs the 24-layer stack once."""
        if not self._memo_hit(self._ctx_memo, x):
            self._ctx_memo = (x, x._version, self._handle().decode_latents(x))
        return self._ctx_memo[2]

    def decode(self, x, queries):
        """x [B,M,latent_dim], queries [B,Q,3] -> logits [B,Q,1] (:408-424)."""
        h = self._handle()
        return h.decode_queries(self._context(x), queries).unsqueeze(-1)

    def forward(self, pc, queries):
        kl, x = self.encode(pc)
        o = self.decode(x, queries).squeeze(-1)
        return {'logits': o, 'kl': kl}


class AutoEncoder(nn.Module):
    """models_ae.py:181 ('not actually used' in the reference) - needs torch_cluster.fps."""

    def __init__(self, **kw):
        super().__init__()
        raise NotImplementedError("the deterministic AutoEncoder needs torch_cluster.fps and is unused by the reference")


def create_autoencoder(dim=512, M=512, latent_dim=64, N=2048, determinisitc=False, query_type='point'):
    if determinisitc:
        return AutoEncoder()
    return KLAutoEncoder(depth=24, dim=dim, queries_dim=dim, output_dim=1, num_inputs=N, num_latents=M,
                         latent_dim=latent_dim, heads=8, dim_head=64, query_type=query_type)


# ---- factories (:461-512); the 'mix' and 'learnable' ones construct, the others need torch_cluster.fps (module docstring)
def kl_d512_m512_l512(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=512, N=N, determinisitc=False)


def kl_d512_m512_l64(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=64, N=N, determinisitc=False)


def kl_d512_m512_l32(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=32, N=N, determinisitc=False)


def kl_d512_m512_l32_learn(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=32, N=N, determinisitc=False, query_type='learnable')


def kl_d512_m512_l32_mix(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=32, N=N, determinisitc=False, query_type='mix')


def kl_d512_m512_l16(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=16, N=N, determinisitc=False)


def kl_d512_m512_l8(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=8, N=N, determinisitc=False)


def kl_d512_m512_l4(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=4, N=N, determinisitc=False)


def kl_d512_m512_l2(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=2, N=N, determinisitc=False)


def kl_d512_m512_l1(N=2048):
    return create_autoencoder(dim=512, M=512, latent_dim=1, N=N, determinisitc=False)


def ae_d512_m512(N=2048):
    return create_autoencoder(dim=512, M=512, N=N, determinisitc=True)


def ae_d512_m256(N=2048):
    return create_autoencoder(dim=512, M=256, N=N, determinisitc=True)


def ae_d512_m128(N=2048):
    return create_autoencoder(dim=512, M=128, N=N, determinisitc=True)


def ae_d512_m64(N=2048):
    return create_autoencoder(dim=512, M=64, N=N, determinisitc=True)


def ae_d256_m512(N=2048):
    return create_autoencoder(dim=256, M=512, N=N, determinisitc=True)


def ae_d128_m512(N=2048):
    return create_autoencoder(dim=128, M=512, N=N, determinisitc=True)


def ae_d64_m512(N=2048):
    return create_autoencoder(dim=64, M=512, N=N, determinisitc=True)
