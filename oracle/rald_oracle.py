"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A plain fp32 PyTorch restatement of the RaLD hot path (radar-conditioned latent denoiser +
EDM/Heun sampler, set-latent autoencoder encode/decode, radar-spectrum encoder), written as
pure functions over a state dict.  It exists to CHECK the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it - never the product package ``rald_amd`` (tests/test_no_oracle_in_product.py
    enforces that);
  * it is pinned against golden vectors captured from the reference's own model code run on
    CPU in the build container (tests/golden/make_golden.py, tests/test_oracle_golden.py).

Every function cites the reference file:line (relative to the reference repo root) it
restates.  Nothing here is copied from the reference: the reference is an nn.Module tree,
this is a flat functional form keyed by checkpoint names.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, n, hd = t.shape
    return t.view(b, n, h, hd // h).permute(0, 2, 1, 3)          # [b,h,n,d]


def _attend(q, k, v, h: int) -> torch.Tensor:
    """softmax(q k^T * d^-1/2) v per head; models_radar_generation.py:66-75, models_ae.py:91-104."""
    q, k, v = _heads(q, h), _heads(k, h), _heads(v, h)
    d = q.shape[-1]
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1), v)
    b, hh, n, dd = out.shape
    return out.permute(0, 2, 1, 3).reshape(b, n, hh * dd)


# ======================================================================================
# denoiser (model/models_radar_generation.py)
# ======================================================================================
def positional_embedding(t: torch.Tensor, num_channels: int = 256, max_positions: int = 10000):
    """models_radar_generation.py:27-33 - cat[cos, sin] of outer(t, (1/max_pos)^(i/half))."""
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32) / half
    freqs = (1.0 / max_positions) ** freqs
    x = torch.outer(t, freqs.to(t.dtype))
    return torch.cat([x.cos(), x.sin()], dim=1)


def timestep_embed(sd: SD, c_noise: torch.Tensor, prefix: str = "model.") -> torch.Tensor:
    """:217-219 - [B'] -> [B',1,512]; SiLU after each of the two Linears."""
    e = positional_embedding(c_noise)[:, None]
    e = F.silu(_lin(sd, prefix + "map_layer0", e))
    return F.silu(_lin(sd, prefix + "map_layer1", e))


def ada_layer_norm(sd: SD, p: str, x: torch.Tensor, t_emb: torch.Tensor) -> torch.Tensor:
    """:127-131 - LN without affine, then *(1+scale)+shift; chunk order is (scale, shift);
    the module's SiLU is constructed but never applied."""
    emb = _lin(sd, p + ".linear", t_emb)
    scale, shift = emb.chunk(2, dim=2)
    return F.layer_norm(x, (x.shape[-1],)) * (1 + scale) + shift


def cross_attention(sd: SD, p: str, x, context=None, heads: int = 8):
    """:55-76 - to_q/to_k/to_v without bias, to_out.0 with bias."""
    ctx = x if context is None else context
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", ctx, bias=False)
    v = _lin(sd, p + ".to_v", ctx, bias=False)
    return _lin(sd, p + ".to_out.0", _attend(q, k, v, heads))


def geglu_ff(sd: SD, p_in: str, p_out: str, x):
    """:88-117 / models_ae.py:51-68 - a * gelu_erf(gate), a = first half."""
    a, gate = _lin(sd, p_in, x).chunk(2, dim=-1)
    return _lin(sd, p_out, a * F.gelu(gate))


def transformer_block(sd: SD, p: str, x, t_emb, context, heads: int = 8):
    """:165-169 (LayerScale / DropPath are Identity at init_values=0, drop_path=0)."""
    x = cross_attention(sd, p + "attn1", ada_layer_norm(sd, p + "norm1", x, t_emb), None, heads) + x
    x = cross_attention(sd, p + "attn2", ada_layer_norm(sd, p + "norm2", x, t_emb), context, heads) + x
    x = geglu_ff(sd, p + "ff.net.0.proj", p + "ff.net.2", ada_layer_norm(sd, p + "norm3", x, t_emb)) + x
    return x


def latent_transformer(sd: SD, x, t, cond, depth: int, heads: int = 8, prefix: str = "model.",
                       taps: Optional[dict] = None):
    """LatentArrayTransformer.forward :215-233."""
    t_emb = timestep_embed(sd, t, prefix)
    x = _lin(sd, prefix + "proj_in", x, bias=False)
    for i in range(depth):
        x = transformer_block(sd, f"{prefix}transformer_blocks.{i}.", x, t_emb, cond, heads)
        if taps is not None:
            taps[f"block{i}"] = x
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
    return _lin(sd, prefix + "proj_out", x, bias=False)


# ---- radar encoder (model/models_radar_encoder.py) -----------------------------------
def _gn(sd: SD, p: str, x):
    """Normalize :9-12 - GroupNorm(32 groups, eps 1e-6, affine)."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)                                     # :5-7


def _conv(sd: SD, p: str, x, stride=1, padding=1):
    return F.conv3d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resblock(sd: SD, p: str, x):
    """ResnetBlock.forward :82-100 with temb=None, dropout=0."""
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def _attnblock(sd: SD, p: str, x):
    """AttnBlock.forward :112-135 - single head over r*a*e tokens, scale c^-1/2."""
    h = _gn(sd, p + ".norm", x)
    q = _conv(sd, p + ".q", h, padding=0)
    k = _conv(sd, p + ".k", h, padding=0)
    v = _conv(sd, p + ".v", h, padding=0)
    b, c = q.shape[:2]
    q = q.reshape(b, c, -1).permute(0, 2, 1)
    k = k.reshape(b, c, -1)
    w = torch.bmm(q, k) * (int(c) ** -0.5)
    w = F.softmax(w, dim=2)
    v = v.reshape(b, c, -1)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(x.shape)
    return x + _conv(sd, p + ".proj_out", h, padding=0)


def radar_encoder(sd: SD, x, prefix: str = "radar_enc.", n_levels: int = 5, n_res: int = 2,
                  taps: Optional[dict] = None):
    """Encoder.forward :216-241.  x [B,Cin,R,A,E] -> [B,z,R/16,A/16,E/16]."""
    h = _conv(sd, prefix + "conv_in", x)
    if taps is not None:
        taps["conv_in"] = h
    for lvl in range(n_levels):
        for b in range(n_res):
            h = _resblock(sd, f"{prefix}down.{lvl}.block.{b}", h)
            if f"{prefix}down.{lvl}.attn.{b}.norm.weight" in sd:
                h = _attnblock(sd, f"{prefix}down.{lvl}.attn.{b}", h)
        if taps is not None:
            taps[f"level{lvl}"] = h
        if lvl != n_levels - 1:
            # Downsample :37-41 - zero-pad (0,1) on each spatial dim, conv k3 stride 2, no padding
            h = _conv(sd, f"{prefix}down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1, 0, 1)),
                      stride=2, padding=0)
    h = _resblock(sd, prefix + "mid.block_1", h)
    h = _attnblock(sd, prefix + "mid.attn_1", h)
    h = _resblock(sd, prefix + "mid.block_2", h)
    if taps is not None:
        taps["mid"] = h
    h = _swish(_gn(sd, prefix + "norm_out", h))
    return _conv(sd, prefix + "conv_out", h)


def radar_decoder(sd: SD, z, prefix: str = "decoder.", n_levels: int = 5, n_res: int = 2):
    """Decoder.forward :333-359 (attn_resolutions=() : only mid.attn_1).  z [B,z,R/16,A/16,E/16] -> [B,out_ch,R,A,E]."""
    h = _conv(sd, prefix + "conv_in", z)
    h = _resblock(sd, prefix + "mid.block_1", h)
    h = _attnblock(sd, prefix + "mid.attn_1", h)
    h = _resblock(sd, prefix + "mid.block_2", h)
    for lvl in reversed(range(n_levels)):
        for b in range(n_res + 1):
            h = _resblock(sd, f"{prefix}up.{lvl}.block.{b}", h)
        if lvl != 0:
            # Upsample :18-27 - nearest-neighbour x2, then conv k3
            h = _conv(sd, f"{prefix}up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    h = _swish(_gn(sd, prefix + "norm_out", h))
    return _conv(sd, prefix + "conv_out", h)


def radar_autoencoder_forward(sd: SD, cube: torch.Tensor):
    """RadarAutoencoder.forward :395-406: cube [B,R,A,E,2] -> {'pred': [B,R,A,E,2], 'latent': [B,z,R/16,A/16,E/16]}."""
    z = radar_encoder(sd, cube.permute(0, 4, 1, 2, 3), prefix="encoder.")
    return {"pred": radar_decoder(sd, z).permute(0, 2, 3, 4, 1), "latent": z}


def process_radar_cond(sd: SD, cube: torch.Tensor, unfreeze_radar_enc: bool = True):
    """EDMPrecond.process_radar_cond :363-407.  cube [B,R,A,E,2] -> tokens [B,R'A'E',C]
    (r-major, then a, then e)."""
    x = cube[..., 0:1]
    if unfreeze_radar_enc:
        x = radar_encoder(sd, x.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
    tok = _lin(sd, "radar_token_project", x)
    r, a, e = sd["radar_r_emb.weight"], sd["radar_a_emb.weight"], sd["radar_e_emb.weight"]
    tok = tok + r[None, :, None, None, :] + a[None, None, :, None, :] + e[None, None, None, :, :]
    return tok.reshape(tok.shape[0], -1, tok.shape[-1])


def edm_precond(sd: SD, x, sigma, cond_tokens, depth: int, sigma_data: float = 1.0,
                heads: int = 8):
    """EDMPrecond.forward :418-430 with the condition tokens already computed (hoisting
    process_radar_cond out of the call is bit-identical in eval mode)."""
    x = x.to(torch.float32)
    sigma = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1, 1, 1)
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    f = latent_transformer(sd, c_in * x, c_noise.flatten(), cond_tokens, depth, heads)
    return c_skip * x + c_out * f


def edm_sigma_schedule(num_steps: int = 18, sigma_min: float = 0.002, sigma_max: float = 80.0,
                       rho: float = 7.0) -> torch.Tensor:
    """edm_sampler :246-249 - fp32 Karras schedule with t_N = 0 appended."""
    i = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


def edm_sampler(denoise: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], latents,
                num_steps: int = 18, sigma_min: float = 0.002, sigma_max: float = 80.0,
                rho: float = 7.0):
    """edm_sampler :235-275 at the shipped S_churn=0: gamma=0, t_hat=t_cur and the in-loop
    randn_like is multiplied by exactly 0 (:258-260), so it is not drawn here.
    `denoise(x, sigma)` is one NFE."""
    t = edm_sigma_schedule(num_steps, sigma_min, sigma_max, rho)
    x_next = latents * t[0]
    for i in range(num_steps):
        t_cur, t_next = t[i], t[i + 1]
        x_hat = x_next
        d_cur = (x_hat - denoise(x_hat, t_cur)) / t_cur
        x_next = x_hat + (t_next - t_cur) * d_cur
        if i < num_steps - 1:
            d_prime = (x_next - denoise(x_next, t_next)) / t_next
            x_next = x_hat + (t_next - t_cur) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next


def dit_sample(sd: SD, cube, latents, depth: int, num_steps: int = 18):
    """EDMPrecond.sample :435-449 given the initial latents (drawn per sample from CPU
    generators by the caller, rald_amd.synth.latents)."""
    cond = process_radar_cond(sd, cube)
    return edm_sampler(lambda x, s: edm_precond(sd, x, s, cond, depth), latents, num_steps)


def edm_loss(sd: SD, y, cube_tokens, rnd_normal, noise, depth: int, p_mean=-1.2, p_std=1.2):
    """EDMLoss.__call__ :283-295 with the two random draws passed in."""
    sigma = (rnd_normal * p_std + p_mean).exp()
    weight = (sigma ** 2 + 1) / sigma ** 2
    d = edm_precond(sd, y + noise * sigma, sigma, cube_tokens, depth)
    return (weight * (d - y) ** 2).mean()


# ======================================================================================
# set-latent autoencoder (model/models_ae.py)
# ======================================================================================
def point_embed(sd: SD, pts: torch.Tensor):
    """PointEmbed.forward :128-138 - Linear(cat[sin(p.basis), cos(p.basis), p])."""
    proj = torch.einsum("bnd,de->bne", pts, sd["point_embed.basis"])
    feat = torch.cat([proj.sin(), proj.cos(), pts], dim=2)
    return _lin(sd, "point_embed.mlp", feat)


def _ln(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"])


def ae_attention(sd: SD, p: str, x, context=None, heads: int = 8):
    """PreNorm + Attention (:41-49, :84-105): LN on x always, LN on context iff the block has
    norm_context; to_kv is one Linear whose first half is k and second half v."""
    xn = _ln(sd, p + "norm", x)
    if context is None:
        ctx = xn
    elif (p + "norm_context.weight") in sd:
        ctx = _ln(sd, p + "norm_context", context)
    else:
        ctx = context
    q = _lin(sd, p + "fn.to_q", xn, bias=False)
    k, v = _lin(sd, p + "fn.to_kv", ctx, bias=False).chunk(2, dim=-1)
    return _lin(sd, p + "fn.to_out", _attend(q, k, v, heads))


def ae_ff(sd: SD, p: str, x):
    return geglu_ff(sd, p + "fn.net.0", p + "fn.net.2", _ln(sd, p + "norm", x))


def ae_encode_moments(sd: SD, pc: torch.Tensor, heads: int = 8):
    """KLAutoEncoder.encode :351-399 up to (mean, logvar): the 'mix' query (:380-387) or, when the state dict holds
    `latents.weight` instead of s_/d_latents, the 'learnable' one (:378-379)."""
    b = pc.shape[0]
    emb = point_embed(sd, pc)
    if "latents.weight" in sd:
        x = sd["latents.weight"][None].expand(b, -1, -1)
    else:
        s_q = sd["s_latents.weight"][None].expand(b, -1, -1)
        d_q = sd["d_latents.weight"][None].expand(b, -1, -1)
        d_q = ae_attention(sd, "mix_attn_layer.", d_q, emb, heads)       # no residual (:384)
        x = _lin(sd, "query_proj", s_q + d_q)
    x = ae_attention(sd, "cross_attend_blocks.0.", x, emb, heads=1) + x  # 1 head x dim (:309)
    x = ae_ff(sd, "cross_attend_blocks.1.", x) + x
    return _lin(sd, "mean_fc", x), _lin(sd, "logvar_fc", x)


def diag_gaussian(mean, logvar, eps):
    """DiagonalGaussianDistribution :141-163 - returns (z, kl[B])."""
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std = torch.exp(0.5 * logvar)
    var = torch.exp(logvar)
    z = mean + std * eps
    kl = 0.5 * torch.mean(mean.pow(2) + var - 1.0 - logvar, dim=[1, 2])
    return z, kl


def ae_encode(sd: SD, pc, eps):
    """encode :351-405 with the posterior noise `eps` passed in (the reference draws it from
    the CPU global RNG, :153)."""
    mean, logvar = ae_encode_moments(sd, pc)
    z, kl = diag_gaussian(mean, logvar, eps)
    return kl, z, mean, logvar


def ae_latent_stack(sd: SD, z, depth: int, heads: int = 8):
    """decode :410-414 - proj then depth x (self-attn + residual, FF + residual)."""
    x = _lin(sd, "proj", z)
    for i in range(depth):
        x = ae_attention(sd, f"layers.{i}.0.", x, None, heads) + x
        x = ae_ff(sd, f"layers.{i}.1.", x) + x
    return x


def ae_decode_queries(sd: SD, x, qpts):
    """decode :417-424 - PreNorm(q, ctx) 1-head cross-attention, no residual, decoder_ff=None."""
    qe = point_embed(sd, qpts)
    lat = ae_attention(sd, "decoder_cross_attn.", qe, x, heads=1)
    return _lin(sd, "to_outputs", lat)                                   # [B,Q,1]


def ae_decode(sd: SD, z, qpts, depth: int):
    return ae_decode_queries(sd, ae_latent_stack(sd, z, depth), qpts)
