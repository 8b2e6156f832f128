"""Parameter specs (names + shapes, in the reference's state_dict order) and the
deterministic name-seeded weight recipe.

The reference ships no weights (README.md:59 links Google Drive), so parity is pinned on
seeded random weights.  Every tensor is regenerated from (seed, name, shape) alone with a CPU
``torch.Generator`` - the stream is identical wherever the same torch build runs - so the
hundreds of MB of weights never have to be committed or travel.

State-dict key names/shapes follow the reference constructors:
  * DiT / EDMPrecond        model/models_radar_generation.py:171-213, :314-361
  * radar Encoder           model/models_radar_encoder.py:137-214
  * KLAutoEncoder ('mix')   model/models_ae.py:284-349
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import List, Sequence, Tuple

import numpy as np
import torch

Spec = List[Tuple[str, Tuple[int, ...]]]


# --------------------------------------------------------------------------------------
# specs
# --------------------------------------------------------------------------------------
def radar_encoder_spec(prefix: str = "radar_enc.", ch: int = 64, in_channels: int = 1,
                       z_channels: int = 16, ch_mult: Sequence[int] = (1, 1, 2, 2, 4),
                       num_res_blocks: int = 2) -> Spec:
    """models_radar_encoder.py:137-214 (Encoder.__init__).  attn_resolutions=((8,4,2),) with
    resolution (128,64,32) and 5 levels puts AttnBlocks on the last level only (:183-184)."""
    s: Spec = []

    def conv(name, cout, cin, k):
        s.append((f"{prefix}{name}.weight", (cout, cin, k, k, k)))
        s.append((f"{prefix}{name}.bias", (cout,)))

    def norm(name, c):
        s.append((f"{prefix}{name}.weight", (c,)))
        s.append((f"{prefix}{name}.bias", (c,)))

    def resblock(name, cin, cout):
        norm(f"{name}.norm1", cin)
        conv(f"{name}.conv1", cout, cin, 3)
        norm(f"{name}.norm2", cout)
        conv(f"{name}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{name}.nin_shortcut", cout, cin, 1)

    def attn(name, c):
        norm(f"{name}.norm", c)
        for p in ("q", "k", "v", "proj_out"):
            conv(f"{name}.{p}", c, c, 1)

    conv("conv_in", ch, in_channels, 3)
    nlev = len(ch_mult)
    in_ch_mult = (1,) + tuple(ch_mult)
    block_in = ch
    for lvl in range(nlev):
        block_in = ch * in_ch_mult[lvl]
        block_out = ch * ch_mult[lvl]
        has_attn = lvl == nlev - 1
        for b in range(num_res_blocks):
            resblock(f"down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if has_attn:
            for b in range(num_res_blocks):
                attn(f"down.{lvl}.attn.{b}", block_in)
        if lvl != nlev - 1:
            conv(f"down.{lvl}.downsample.conv", block_in, block_in, 3)
    resblock("mid.block_1", block_in, block_in)
    attn("mid.attn_1", block_in)
    resblock("mid.block_2", block_in, block_in)
    norm("norm_out", block_in)
    conv("conv_out", z_channels, block_in, 3)
    return s


def radar_decoder_spec(prefix: str = "decoder.", ch: int = 64, out_ch: int = 2, z_channels: int = 16,
                       ch_mult: Sequence[int] = (1, 1, 2, 2, 4), num_res_blocks: int = 2) -> Spec:
    """models_radar_encoder.py:243-332 (Decoder.__init__; attn_resolutions=() so only mid.attn_1).
    `up` is built from the last level down but stored by level index (insert(0), :326)."""
    s: Spec = []

    def conv(name, cout, cin, k):
        s.append((f"{prefix}{name}.weight", (cout, cin, k, k, k)))
        s.append((f"{prefix}{name}.bias", (cout,)))

    def norm(name, c):
        s.append((f"{prefix}{name}.weight", (c,)))
        s.append((f"{prefix}{name}.bias", (c,)))

    def resblock(name, cin, cout):
        norm(f"{name}.norm1", cin)
        conv(f"{name}.conv1", cout, cin, 3)
        norm(f"{name}.norm2", cout)
        conv(f"{name}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{name}.nin_shortcut", cout, cin, 1)

    nlev = len(ch_mult)
    block_in = ch * ch_mult[-1]
    conv("conv_in", block_in, z_channels, 3)
    resblock("mid.block_1", block_in, block_in)
    norm("mid.attn_1.norm", block_in)
    for p in ("q", "k", "v", "proj_out"):
        conv(f"mid.attn_1.{p}", block_in, block_in, 1)
    resblock("mid.block_2", block_in, block_in)
    ups = {}
    for lvl in reversed(range(nlev)):
        start = len(s)
        block_out = ch * ch_mult[lvl]
        for b in range(num_res_blocks + 1):
            resblock(f"up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(f"up.{lvl}.upsample.conv", block_in, block_in, 3)
        ups[lvl] = s[start:]
        del s[start:]
    for lvl in range(nlev):
        s.extend(ups[lvl])
    norm("norm_out", block_in)
    conv("conv_out", out_ch, block_in, 3)
    return s


def radar_autoencoder_spec(basic_channel: int = 64, embed_dim: int = 16, ch_mult: Sequence[int] = (1, 1, 2, 2, 4),
                           num_res_blocks: int = 2) -> Spec:
    """RadarAutoencoder (models_radar_encoder.py:366-379): Encoder(in_channels=2 default) + Decoder."""
    return (radar_encoder_spec("encoder.", ch=basic_channel, in_channels=2, z_channels=embed_dim, ch_mult=ch_mult,
                               num_res_blocks=num_res_blocks)
            + radar_decoder_spec("decoder.", ch=basic_channel, out_ch=2, z_channels=embed_dim, ch_mult=ch_mult,
                                 num_res_blocks=num_res_blocks))


def dit_spec(channels: int = 32, depth: int = 24, n_heads: int = 8, d_head: int = 64,
             t_channels: int = 256, context_dim: int | None = None,
             with_radar: bool = True, enc_hidden_ch: int = 64, enc_radar_ch: int = 16,
             radar_token_channel: int = 512, rae: Sequence[int] = (8, 4, 2),
             prefix: str = "model.") -> Spec:
    """EDMPrecond state_dict: LatentArrayTransformer under 'model.' (:336), then radar_enc,
    r/a/e embeddings and radar_token_project (:346-356)."""
    D = n_heads * d_head
    cdim = D if context_dim is None else context_dim
    s: Spec = [(f"{prefix}proj_in.weight", (D, channels))]
    for i in range(depth):
        p = f"{prefix}transformer_blocks.{i}."
        s += [
            (p + "attn1.to_q.weight", (D, D)), (p + "attn1.to_k.weight", (D, D)),
            (p + "attn1.to_v.weight", (D, D)), (p + "attn1.to_out.0.weight", (D, D)),
            (p + "attn1.to_out.0.bias", (D,)),
            (p + "ff.net.0.proj.weight", (8 * D, D)), (p + "ff.net.0.proj.bias", (8 * D,)),
            (p + "ff.net.2.weight", (D, 4 * D)), (p + "ff.net.2.bias", (D,)),
            (p + "attn2.to_q.weight", (D, D)), (p + "attn2.to_k.weight", (D, cdim)),
            (p + "attn2.to_v.weight", (D, cdim)), (p + "attn2.to_out.0.weight", (D, D)),
            (p + "attn2.to_out.0.bias", (D,)),
        ]
        for n in ("norm1", "norm2", "norm3"):
            s += [(p + f"{n}.linear.weight", (2 * D, D)), (p + f"{n}.linear.bias", (2 * D,))]
    s += [
        (f"{prefix}norm.weight", (D,)), (f"{prefix}norm.bias", (D,)),
        (f"{prefix}proj_out.weight", (channels, D)),
        (f"{prefix}map_layer0.weight", (D, t_channels)), (f"{prefix}map_layer0.bias", (D,)),
        (f"{prefix}map_layer1.weight", (D, D)), (f"{prefix}map_layer1.bias", (D,)),
    ]
    if with_radar:
        s += radar_encoder_spec("radar_enc.", ch=enc_hidden_ch, in_channels=1,
                                z_channels=enc_radar_ch)
        s += [
            ("radar_r_emb.weight", (rae[0], radar_token_channel)),
            ("radar_a_emb.weight", (rae[1], radar_token_channel)),
            ("radar_e_emb.weight", (rae[2], radar_token_channel)),
            ("radar_token_project.weight", (radar_token_channel, enc_radar_ch)),
            ("radar_token_project.bias", (radar_token_channel,)),
        ]
    return s


def ae_spec(dim: int = 512, num_latents: int = 512, latent_dim: int = 32, depth: int = 24,
            heads: int = 8, dim_head: int = 64, query_type: str = "mix") -> Spec:
    """KLAutoEncoder state_dict order (models_ae.py:308-349): cross_attend_blocks,
    point_embed, layers, [s_latents, d_latents, mix_attn_layer, query_proj], decoder_cross_attn,
    to_outputs, proj, mean_fc, logvar_fc.  heads*dim_head (=512) is the latent-stack inner dim
    regardless of `dim` (create_autoencoder hard-codes 8x64, :447-458)."""
    inner = heads * dim_head
    s: Spec = []

    def attn(p, qdim, cdim, inner_dim, ctx_norm):
        s.extend([
            (p + "fn.to_q.weight", (inner_dim, qdim)), (p + "fn.to_kv.weight", (2 * inner_dim, cdim)),
            (p + "fn.to_out.weight", (qdim, inner_dim)), (p + "fn.to_out.bias", (qdim,)),
            (p + "norm.weight", (qdim,)), (p + "norm.bias", (qdim,)),
        ])
        if ctx_norm:
            s.extend([(p + "norm_context.weight", (cdim,)), (p + "norm_context.bias", (cdim,))])

    def ff(p, d):
        s.extend([
            (p + "fn.net.0.weight", (8 * d, d)), (p + "fn.net.0.bias", (8 * d,)),
            (p + "fn.net.2.weight", (d, 4 * d)), (p + "fn.net.2.bias", (d,)),
            (p + "norm.weight", (d,)), (p + "norm.bias", (d,)),
        ])

    attn("cross_attend_blocks.0.", dim, dim, dim, True)      # heads=1, dim_head=dim (:309)
    ff("cross_attend_blocks.1.", dim)
    s += [("point_embed.basis", (3, 24)), ("point_embed.mlp.weight", (dim, 51)),
          ("point_embed.mlp.bias", (dim,))]
    # `self.layers` is registered (empty) before the query-type members and filled afterwards
    # (:319 vs :327-339), so its entries come first in state_dict order.
    for i in range(depth):
        attn(f"layers.{i}.0.", dim, dim, inner, False)
        ff(f"layers.{i}.1.", dim)
    if query_type == "mix":
        s += [("s_latents.weight", (num_latents, dim)), ("d_latents.weight", (num_latents, dim))]
        attn("mix_attn_layer.", dim, dim, inner, False)
        s += [("query_proj.weight", (dim, dim)), ("query_proj.bias", (dim,))]
    elif query_type == "learnable":
        s += [("latents.weight", (num_latents, dim))]
    attn("decoder_cross_attn.", dim, dim, dim, True)
    s += [("to_outputs.weight", (1, dim)), ("to_outputs.bias", (1,)),
          ("proj.weight", (dim, latent_dim)), ("proj.bias", (dim,)),
          ("mean_fc.weight", (latent_dim, dim)), ("mean_fc.bias", (latent_dim,)),
          ("logvar_fc.weight", (latent_dim, dim)), ("logvar_fc.bias", (latent_dim,))]
    return s


# --------------------------------------------------------------------------------------
# deterministic recipe
# --------------------------------------------------------------------------------------
def hash32(seed: int, name: str) -> int:
    return zlib.crc32(f"{seed}:{name}".encode()) & 0xFFFFFFFF


def point_embed_basis(hidden_dim: int = 48) -> torch.Tensor:
    """models_ae.py:115-124: e_k = 2^k * pi, k<8, block-diagonal over the 3 axes -> [3, 24]."""
    n = hidden_dim // 6
    e = torch.pow(2, torch.arange(n)).float() * np.pi
    z = torch.zeros(n)
    return torch.stack([torch.cat([e, z, z]), torch.cat([z, e, z]), torch.cat([z, z, e])])


def _std_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) of the seeded normal for one tensor."""
    if name.endswith("_emb.weight") or name.endswith("latents.weight"):
        return 0.0, 1.0                       # nn.Embedding default N(0,1)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return 0.0, 1.0 / math.sqrt(fan_in)   # keeps activations O(1) through 24 blocks
    if name.endswith(".weight"):
        return 1.0, 0.1                       # LayerNorm / GroupNorm gains
    return 0.0, 0.1                           # biases (and norm shifts)


def seeded_tensor(seed: int, name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    if name.endswith("point_embed.basis"):
        return point_embed_basis()
    g = torch.Generator("cpu").manual_seed(hash32(seed, name))
    mean, std = _std_for(name, tuple(shape))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std + mean


def make_state_dict(spec: Spec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU tensors for every entry of `spec` (note: proj_out is NOT zero here - a
    zero-initialised proj_out makes parity vacuous, SURVEY.md §8c 'Zero-init trap')."""
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in spec)


def spec_of_state_dict(sd) -> Spec:
    return [(k, tuple(v.shape)) for k, v in sd.items()]


# --------------------------------------------------------------------------------------
# stress variants of the seeded weights (tests/golden/make_golden.py G18 / G19): the plain recipe gives flat softmaxes and
# O(1) partial sums, which is where the folded fp16 tables and the fp16 x 2^-6 partial-sum slabs are least stressed
# --------------------------------------------------------------------------------------
AE_PEAKED_KEYS = ("decoder_cross_attn.fn.to_q.weight", "decoder_cross_attn.fn.to_kv.weight",
                  "cross_attend_blocks.0.fn.to_q.weight", "cross_attend_blocks.0.fn.to_kv.weight",
                  "mix_attn_layer.fn.to_q.weight", "mix_attn_layer.fn.to_kv.weight")


def stress_ae_state_dict(sd, scale: float = 4.0, embed_bias_offset: float = 0.5, out_bias=None):
    """Peaked attentions: the q / kv projections of the three point / query attentions times `scale` (logits times scale^2),
    a constant offset on the PointEmbed bias (a large bias-like term in the folded score tables), and optionally a given
    `to_outputs.bias` (chosen by the golden script so that the logits straddle 0)."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for k in AE_PEAKED_KEYS:
        if k in out:
            out[k] *= scale
    out["point_embed.mlp.bias"] += embed_bias_offset
    if out_bias is not None:
        out["to_outputs.bias"] = torch.tensor([float(out_bias)], dtype=torch.float32)
    return out


def stress_dit_state_dict(sd, scale: float = 8.0):
    """Large partial sums: every block's attn1 / attn2 `to_out.0.weight` and `ff.net.2.weight` times `scale`, so the per-head and
    split-K partial sums of the small-batch path (fp16 x 2^-6 slabs) and the residual stream grow by that factor."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for k in out:
        if k.endswith("to_out.0.weight") or k.endswith("ff.net.2.weight"):
            out[k] *= scale
    return out
