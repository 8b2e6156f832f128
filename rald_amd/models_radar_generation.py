"""Drop-in for the reference's ``model/models_radar_generation.py``: same factory names
(``kl_d512_m512_l32_d24_edm`` ..., looked up via ``__dict__[name](configs=...)``,
main_generation.py:122), same ``EDMPrecond`` constructor / ``forward`` / ``sample`` /
``process_radar_cond`` signatures (:314-449), same ``edm_sampler`` (:235-275), ``EDMLoss``
(:277-295), ``StackedRandomGenerator`` (:297-311) and the same ``state_dict`` keys, so the
reference's checkpoints load with ``strict=True`` (utils/misc.py:346).

The modules hold fp32 ``nn.Parameter``s only; every forward runs the hand-written HIP kernels of
``librald_hip.so`` through the C-ABI (include/rald_hip.h).  There is no PyTorch compute path: on
a CPU tensor, or without the library, calls raise.
"""
from __future__ import annotations

import os

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import weights as _w
from ._handles import DitHandle
from ._lib import DitConfig


# ----------------------------------------------------------------------------------------------
# parameter containers with the reference's key names
# ----------------------------------------------------------------------------------------------
def _init_param(name: str, shape, fan_in=None) -> torch.Tensor:
    """Fresh-module initialisation like the reference's torch defaults: Linear / Conv weights AND biases uniform
    +-1/sqrt(fan_in) (`fan_in` of a bias = that of the weight it belongs to), ones/zeros for norms, N(0,1) for
    embeddings, and the zero-initialised proj_out (:198-201)."""
    if name.endswith("proj_out.weight") and "attn" not in name:
        return torch.zeros(shape)
    if name.endswith("_emb.weight") or name.endswith("latents.weight"):
        return torch.randn(shape)
    if len(shape) >= 2:
        bound = 1.0 / math.sqrt(int(np.prod(shape[1:])))
        return torch.empty(shape).uniform_(-bound, bound)
    is_norm = any(seg.startswith("norm") for seg in name.split(".")[:-1]) and not name.endswith("linear.bias")
    if is_norm:
        return torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
    if name.endswith(".bias") and fan_in:
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(shape).uniform_(-bound, bound)
    return torch.zeros(shape)


def build_param_tree(root: nn.Module, spec, buffers=()) -> None:
    """Registers every (dotted name, shape) of `spec` under `root` as nested plain nn.Modules so
    that ``root.state_dict()`` has exactly the reference's keys, in the reference's order."""
    spec = list(spec)
    shapes = {n: tuple(sh) for n, sh in spec}
    for name, shape in spec:
        parts = name.split(".")
        w = shapes.get(name[:-len("bias")] + "weight") if name.endswith(".bias") else None
        fan_in = int(np.prod(w[1:])) if w is not None and len(w) >= 2 else None
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        if name in buffers:
            mod.register_buffer(parts[-1], _w.seeded_tensor(0, name, shape))
        else:
            mod.register_parameter(parts[-1], nn.Parameter(_init_param(name, tuple(shape), fan_in)))


class _HipBacked(nn.Module):
    """Mixin: lazily creates the C handle on the parameters' device and re-packs the weights
    whenever a parameter/buffer changed (load_state_dict, optimizer step, .to())."""

    def _state_fingerprint(self):
        """(data_ptr, version) of every parameter and buffer: changes when weights are loaded, trained or moved.  A manual walk of
        the module tree: nn.Module.parameters() builds a dotted name per entry and costs 4x as much (2.8 vs 0.7 ms for the
        637 tensors of the 24-block model) - this runs on every forward()."""
        out = []

        def walk(mod):
            for t in mod._parameters.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
            for t in mod._buffers.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
            for c in mod._modules.values():
                if c is not None:
                    walk(c)
        walk(self)
        return tuple(out)

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    @staticmethod
    def _memo_hit(memo, t: torch.Tensor) -> bool:
        """A memo entry is (tensor, version, ...): it hits only for the SAME tensor object at the same version.  The
        entry holds the tensor, so the object stays alive and `is` cannot match a later tensor that the caching
        allocator placed at a recycled address (a (data_ptr, _version) key does: fresh `.to(device)` copies and
        tensors the library wrote through data_ptr() all sit at version 0)."""
        return memo is not None and memo[0] is t and memo[1] == t._version


# ----------------------------------------------------------------------------------------------
# LatentArrayTransformer (:171-233)
# ----------------------------------------------------------------------------------------------
def _qkv_code(name: str) -> int:
    codes = {"bf16": 0, "fp8": 1, "fp8_ff1": 2, "fp8_ff": 3}
    if name not in codes:
        raise ValueError("qkv_dtype must be 'bf16', 'fp8' (MXFP8 q/k/v), 'fp8_ff1' (+ GEGLU projection) or 'fp8_ff' (+ whole feed-forward)")
    return codes[name]


class LatentArrayTransformer(_HipBacked):
    def __init__(self, in_channels, t_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None,
                 out_channels=None, _owner=None):
        super().__init__()
        if dropout != 0.:
            raise NotImplementedError("dropout > 0 is not on the reference's shipped path")
        if out_channels is not None:
            raise NotImplementedError("out_channels is never set by the reference's factories")
        self.in_channels = in_channels
        self.t_channels = t_channels
        self.context_dim = context_dim
        self.depth, self.n_heads, self.d_head = depth, n_heads, d_head
        object.__setattr__(self, "_owner", _owner)      # EDMPrecond that shares its handle (not a submodule)
        self.qkv_dtype = os.environ.get("RALD_QKV_DTYPE", "bf16")   # "bf16" | "fp8" (MXFP8 q/k/v projections, BASELINE config #5)
        self._hip = None
        self._hip_fp = None
        spec = _w.dit_spec(channels=in_channels, depth=depth, n_heads=n_heads, d_head=d_head, t_channels=t_channels,
                           context_dim=context_dim, with_radar=False, prefix="")
        build_param_tree(self, spec)

    def _config(self, n_latents=512, n_cond_tokens=64) -> DitConfig:
        D = self.n_heads * self.d_head
        return DitConfig(n_latents=n_latents, channels=self.in_channels, depth=self.depth, n_heads=self.n_heads,
                         d_head=self.d_head, t_channels=self.t_channels,
                         context_dim=D if self.context_dim is None else self.context_dim,
                         n_cond_tokens=n_cond_tokens, with_radar_enc=0, enc_hidden_ch=64, enc_radar_ch=16,
                         radar_r=128, radar_a=64, radar_e=32, sigma_data=1.0, qkv_dtype=_qkv_code(self.qkv_dtype))

    def _handle(self, n_latents: int, n_cond_tokens: int) -> DitHandle:
        if self._owner is not None:
            return self._owner._handle()
        fp = (self._state_fingerprint(), n_latents, n_cond_tokens, self.qkv_dtype)
        if self._hip is None or self._hip_fp != fp:
            h = DitHandle(self._config(n_latents, n_cond_tokens))
            h.load(("model." + k, v) for k, v in self.state_dict().items())
            self._hip, self._hip_fp = h, fp
        return self._hip

    def forward(self, x, t, cond=None):
        """x [B,N,C], t [B'] (= c_noise; B' = 1 or B), cond [B,T,context_dim] -> F_x [B,N,C]."""
        if cond is None:
            raise NotImplementedError("the reference always passes radar condition tokens (cond)")
        if x.dim() != 3 or cond.dim() != 3:
            raise RuntimeError(f"x must be [B,N,C] and cond [B,T,context_dim], got {tuple(x.shape)} and {tuple(cond.shape)}")
        if x.shape[0] != cond.shape[0]:
            raise RuntimeError(f"x holds {x.shape[0]} samples but cond holds {cond.shape[0]}: one set of condition tokens per sample "
                               "(the reference's einsum 'b i d, b j d' fails the same way, models_radar_generation.py:66)")
        if x.shape[2] != self.in_channels:
            raise RuntimeError(f"x has {x.shape[2]} channels, the module was built for {self.in_channels}")
        h = self._handle(x.shape[1], cond.shape[1])
        t = torch.as_tensor(t, dtype=torch.float32).flatten().cpu()
        h.set_sigmas(torch.exp(4.0 * t.double()).tolist())      # c_noise = ln(sigma)/4  (:425)
        cache = h.encode_cond_tokens(cond)
        return h.denoise(x, cache, 0, per_sample=t.numel() > 1, raw_F=True)


# ----------------------------------------------------------------------------------------------
# sampler / loss / RNG helpers
# ----------------------------------------------------------------------------------------------
def edm_sampler(net, latents, class_labels=None, cond_type=None, randn_like=torch.randn_like,
                num_steps=18, sigma_min=0.002, sigma_max=80, rho=7,
                S_churn=0, S_min=0, S_max=float('inf'), S_noise=1):
    """Same signature as the reference (:235-240).  With the shipped S_churn=0 the whole loop
    (2*num_steps-1 NFEs + Heun updates) runs inside librald_hip.so with the radar condition
    encoded once; `randn_like` is never consumed because the reference multiplies it by exactly
    0 (:258-260).  S_churn > 0 is not on the reference's path and is rejected."""
    if S_churn != 0:
        raise NotImplementedError("S_churn > 0: the reference ships S_churn=0 (:239)")
    sigma_min = max(sigma_min, net.sigma_min)
    sigma_max = min(sigma_max, net.sigma_max)
    return net._sample_from(latents, class_labels, cond_type, num_steps, sigma_min, sigma_max, rho)


class _EdmDenoiseFn(torch.autograd.Function):
    """EDMPrecond.forward as an autograd node: forward = radar encoder + tokeniser + 24-block denoiser + EDM pre/post-conditioning
    through the HIP training kernels (rald_amd.train_encoder / train_dit, activations kept), backward = their hand-written
    backward passes.  The module's parameters are inputs of the node, so ``loss.backward()`` fills ``p.grad`` the ordinary way
    and ``DistributedDataParallel``'s reducer hooks fire on them (main_generation.py:157-159, utils/misc.py:255)."""

    @staticmethod
    def forward(ctx, module, x, sigma, cube, *params):
        tr = module._autograd_trainers()
        tokens = tr["enc"].forward(cube[..., 0:1].contiguous() if cube.shape[-1] != 1 else cube.contiguous())
        D, st = tr["dit"].forward_denoised(x, tokens, sigma)
        ctx.module, ctx.st = module, st
        return D

    @staticmethod
    def backward(ctx, dD):
        if ctx.st is None:
            raise RuntimeError("EDMPrecond.forward: backward through the same forward a second time - the saved activations "
                               "(~1 GiB per sample batch) are released by the first backward; retain_graph is not supported")
        tr = ctx.module._autograd_trainers()
        for sh in tr["shadow"].values():
            sh.grad = None
        dtok = tr["dit"].backward_denoised(ctx.st, dD.contiguous())
        tr["enc"].backward(dtok)
        ctx.st = None
        grads = tuple(tr["shadow"][n].grad if tr["shadow"][n].grad is not None else torch.zeros_like(tr["shadow"][n])
                      for n in tr["names"])
        return (None, None, None, None) + grads


class EDMLoss:
    """:277-295, the reference's call signature and value.  Under grad mode ``net(...)`` is differentiable (``_EdmDenoiseFn``):
    ``EDMLoss()(model, latents, cube, 'radar').backward()`` - the reference's own training loop, also under torch DDP - works
    unchanged.  ``rald_amd.train_dit.EdmTrainer`` is the fused alternative (loss + backward + clip / AdamW / EMA on flat
    storage, the two random draws passed explicitly)."""

    def __init__(self, P_mean=-1.2, P_std=1.2, sigma_data=1):
        self.P_mean, self.P_std, self.sigma_data = P_mean, P_std, sigma_data

    def __call__(self, net, inputs, labels=None, cond_type=None, augment_pipe=None):
        rnd_normal = torch.randn([inputs.shape[0], 1, 1], device=inputs.device)
        sigma = (rnd_normal * self.P_std + self.P_mean).exp()
        weight = (sigma ** 2 + self.sigma_data ** 2) / (sigma * self.sigma_data) ** 2
        y, _ = augment_pipe(inputs) if augment_pipe is not None else (inputs, None)
        n = torch.randn_like(y) * sigma
        D_yn = net(y + n, sigma, labels, cond_type)
        return (weight * ((D_yn - y) ** 2)).mean()


class StackedRandomGenerator:
    """:297-311.  Generators live on the CPU: device generator streams differ between CPU, CUDA
    and HIP, and 'identical noise seeds' against the reference's CPU path means the CPU stream
    (SURVEY.md §8b RNG); draws are moved to `device` afterwards."""

    def __init__(self, device, seeds):
        self.device = device
        self.generators = [torch.Generator("cpu").manual_seed(int(seed) % (1 << 32)) for seed in seeds]

    def randn(self, size, **kwargs):
        assert size[0] == len(self.generators)
        kwargs.pop("device", None)
        return torch.stack([torch.randn(size[1:], generator=gen, **kwargs) for gen in self.generators]).to(self.device)

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype, layout=input.layout)

    def randint(self, *args, size, **kwargs):
        assert size[0] == len(self.generators)
        kwargs.pop("device", None)
        return torch.stack([torch.randint(*args, size=size[1:], generator=gen, **kwargs) for gen in self.generators]).to(self.device)


# ----------------------------------------------------------------------------------------------
# EDMPrecond (:314-449)
# ----------------------------------------------------------------------------------------------
class EDMPrecond(_HipBacked):
    def __init__(self, n_latents=512, channels=8, use_fp16=False, sigma_min=0, sigma_max=float('inf'),
                 sigma_data=1, n_heads=8, d_head=64, depth=12, configs=None):
        super().__init__()
        if use_fp16:
            raise NotImplementedError("use_fp16 is never set by the reference (fp32 end to end, :318)")
        self.n_latents, self.channels, self.use_fp16 = n_latents, channels, use_fp16
        self.sigma_min, self.sigma_max, self.sigma_data = sigma_min, sigma_max, sigma_data
        self.configs = configs
        self.n_heads, self.d_head, self.depth = n_heads, d_head, depth
        self.unfreeze_radar_enc = self.configs.get('unfreeze_radar_enc', False)
        if configs.cond_type != 'radar':
            raise NotImplementedError("only cond_type='radar' exists in the reference (:341)")
        if not (self.unfreeze_radar_enc and self.configs.use_radar_enc):
            raise NotImplementedError("the shipped config trains the radar encoder jointly "
                                      "(use_radar_enc=True, unfreeze_radar_enc=True); the frozen-encoder "
                                      "route is not built yet")
        self.radar_token_channel = self.configs.radar_token_channel
        self.model = LatentArrayTransformer(in_channels=channels, t_channels=256, n_heads=n_heads, d_head=d_head,
                                            depth=depth, _owner=self)
        rest = _w.dit_spec(channels=channels, depth=0, with_radar=True, enc_hidden_ch=self.configs.enc_hidden_ch,
                           enc_radar_ch=self.configs.enc_radar_ch, radar_token_channel=self.radar_token_channel,
                           rae=(self.configs.enc_radar_r_dim, self.configs.enc_radar_a_dim, self.configs.enc_radar_e_dim))
        build_param_tree(self, [(n, s) for n, s in rest if not n.startswith("model.")])
        self.qkv_dtype = os.environ.get("RALD_QKV_DTYPE", "bf16")   # "bf16" | "fp8": see LatentArrayTransformer
        self._hip = None
        self._hip_fp = None
        self._cond_memo = None

    # -- handle -----------------------------------------------------------------------------
    def _config(self) -> DitConfig:
        c = self.configs
        return DitConfig(n_latents=self.n_latents, channels=self.channels, depth=self.depth, n_heads=self.n_heads,
                         d_head=self.d_head, t_channels=256, context_dim=self.radar_token_channel,
                         n_cond_tokens=c.enc_radar_r_dim * c.enc_radar_a_dim * c.enc_radar_e_dim, with_radar_enc=1,
                         enc_hidden_ch=c.enc_hidden_ch, enc_radar_ch=c.enc_radar_ch, radar_r=c.input_radar_r_dim,
                         radar_a=c.input_radar_a_dim, radar_e=c.input_radar_e_dim, sigma_data=float(self.sigma_data),
                         qkv_dtype=_qkv_code(self.qkv_dtype))

    def _handle(self) -> DitHandle:
        fp = (self._state_fingerprint(), self.qkv_dtype)
        if self._hip is None or self._hip_fp != fp:
            h = DitHandle(self._config())
            h.load(self.state_dict().items())
            self._hip, self._hip_fp, self._cond_memo = h, fp, None
        return self._hip

    def _cond(self, cube: torch.Tensor, h: "DitHandle | None" = None):
        """(tokens, cond cache) for a radar cube; memoised on the tensor OBJECT (+ its version) so a caller that
        invokes forward() in a loop with the same cube (the reference's sampler does, re-running
        the 287-GFLOP encoder every NFE) encodes it once.  See _memo_hit."""
        if not self._memo_hit(self._cond_memo, cube):
            tokens, cache = (h or self._handle()).encode_cond(cube)      # (h: the caller already paid the 1.6 ms fingerprint walk)
            self._cond_memo = (cube, cube._version, tokens, cache)
        return self._cond_memo[2], self._cond_memo[3]

    # -- reference API ------------------------------------------------------------------------
    def process_radar_cond(self, radar_cube):
        """(B,R,A,E,ch) -> (B, R'A'E', C) condition tokens (:363-407)."""
        return self._cond(radar_cube)[0]

    def _autograd_trainers(self):
        """The training kernels' view of this module for the autograd route: shadow Parameters that SHARE the real parameters'
        storage (the trainers accumulate into `.grad` of what they are given; the real `.grad`s belong to autograd), rebuilt
        when a parameter's storage moved, bf16 compute copies refreshed when a parameter's version did."""
        from .train_dit import DitTrainer
        from .train_encoder import EncoderTrainer
        named = list(self.named_parameters())
        ptrs = tuple(p.data_ptr() for _, p in named)
        vers = tuple(p._version for _, p in named)
        tr = self.__dict__.get("_ag_trainers")
        if tr is None or tr["ptrs"] != ptrs:
            shadow = {n: nn.Parameter(p.detach(), requires_grad=True) for n, p in named}
            tr = dict(shadow=shadow, names=[n for n, _ in named], ptrs=ptrs, vers=vers,
                      dit=DitTrainer({k[len("model."):]: v for k, v in shadow.items() if k.startswith("model.")}, self.depth,
                                     n_heads=self.n_heads, sigma_data=float(self.sigma_data)),
                      enc=EncoderTrainer({k: v for k, v in shadow.items() if k.startswith("radar_")}))
            self.__dict__["_ag_trainers"] = tr
        elif tr["vers"] != vers:
            tr["dit"].refresh_weights()
            tr["vers"] = vers
        return tr

    def forward(self, x, sigma, label_tokens=None, cond_type=None, force_fp32=False, **model_kwargs):
        if cond_type != 'radar':
            raise NotImplementedError("cond_type must be 'radar'")
        if label_tokens is None:
            raise RuntimeError("label_tokens (the radar cube [B,R,A,E,2]) is required: the reference's forward dereferences it too (:414)")
        if x.dim() != 3 or x.shape[0] != label_tokens.shape[0]:
            raise RuntimeError(f"x must be [B,{self.n_latents},{self.channels}] with one radar cube per sample; got x {tuple(x.shape)}, "
                               f"cube {tuple(label_tokens.shape)}")
        # Route like the reference uses the module: model.train() + grad mode = the training step (engine_generation.py:47, :93-104:
        # loss_scaler(loss, ...) -> loss.backward()); model.eval() / @torch.no_grad() = inference (engine_generation.py:141, :183)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if self.qkv_dtype != "bf16":
                raise NotImplementedError(f"qkv_dtype={self.qkv_dtype!r} is an inference mode; the differentiable route computes in bf16 "
                                          "(call model.eval() or wrap the call in torch.no_grad())")
            if x.requires_grad or label_tokens.requires_grad:
                raise NotImplementedError("gradients with respect to the noised latents / the radar cube are not built "
                                          "(the reference trains the parameters only: latents come from the frozen VAE)")
            sig = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)
            if sig.numel() == 1:
                sig = sig.expand(x.shape[0])
            if sig.numel() != x.shape[0]:
                raise RuntimeError("sigma must be a scalar or have one entry per sample")
            params = [p for _, p in self.named_parameters()]
            return _EdmDenoiseFn.apply(self, x.to(torch.float32), sig, label_tokens, *params)
        h = self._handle()
        _, cache = self._cond(label_tokens, h)
        sig = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1).cpu()
        if sig.numel() not in (1, x.shape[0]):
            raise RuntimeError("sigma must be a scalar or have one entry per sample")
        h.set_sigmas(sig.tolist())
        return h.denoise(x.to(torch.float32), cache, 0, per_sample=sig.numel() > 1, raw_F=False)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)

    def _sample_from(self, latents, cond, cond_type, num_steps, sigma_min, sigma_max, rho):
        if cond_type != 'radar':
            raise NotImplementedError("cond_type must be 'radar'")
        h = self._handle()
        _, cache = self._cond(cond, h)
        return h.sample(latents, cache, num_steps, float(sigma_min), float(sigma_max), float(rho))

    def _replica(self, i: int) -> DitHandle:
        """i-th independent C handle over the same parameters (0 = the module's own): a handle runs one stream at a time."""
        if i == 0:
            return self._handle()
        fp = (self._state_fingerprint(), self.qkv_dtype)
        reps = self.__dict__.setdefault("_replicas", {})
        if i not in reps or reps[i][1] != fp:
            h = DitHandle(self._config())
            h.load(self.state_dict().items())
            reps[i] = (h, fp)
        return reps[i][0]

    @torch.no_grad()
    def sample_concurrent(self, conds, batch_seeds=None, cond_type=None, num_steps=18):
        """`[self.sample(c, s, cond_type) for c, s in zip(conds, batch_seeds)]` - same values, bit for bit - with every
        condition batch on its own HIP stream and handle replica.  Not a reference API: the reference's evaluate loop
        (engine_generation.py:186-232) samples its batches one after the other.  All kernels of one launch run the same
        phase on every CU at once (MFMA loop, then the HBM-heavy epilogue) and small batches leave most CUs idle; independent
        streams fill both gaps (two batches of 64: +6 % sample.NFE/s)."""
        if cond_type != 'radar':
            raise NotImplementedError("cond_type must be 'radar'")
        conds = list(conds)
        seeds = list(batch_seeds) if batch_seeds is not None else [None] * len(conds)
        if len(seeds) != len(conds):
            raise ValueError("one seed tensor (or None) per condition batch")
        cur = torch.cuda.current_stream()
        streams = self.__dict__.setdefault("_streams", [])
        while len(streams) < len(conds):
            streams.append(torch.cuda.Stream())
        outs = []
        for i, (cond, sd) in enumerate(zip(conds, seeds)):
            if sd is None:
                sd = torch.arange(cond.shape[0])
            rnd = StackedRandomGenerator(cond.device, sd)
            latents = rnd.randn([cond.shape[0], self.n_latents, self.channels])
            h = self._replica(i)
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                _, cache = h.encode_cond(cond, want_tokens=False)
                # graph replays of different streams overlap less than eager launches do (measured: two batches of 8 take
                # 248 ms replayed, 223 ms eager, 241 ms one after the other), so only latency-bound batches replay a graph
                out = h.sample(latents, cache, num_steps, float(max(0.002, self.sigma_min)), float(min(80, self.sigma_max)), 7.0,
                               use_graph=None if cond.shape[0] <= 4 else False)
            latents.record_stream(streams[i])          # allocated on the caller's stream, consumed on the side stream
            out.record_stream(cur)                     # ... and the other way round
            outs.append(out)
        for st in streams[:len(conds)]:
            cur.wait_stream(st)
        return outs

    @torch.no_grad()
    def sample(self, cond, batch_seeds=None, cond_type=None):
        if cond is not None:
            batch_size, device = cond.shape[0], cond.device
            if batch_seeds is None:
                batch_seeds = torch.arange(batch_size)
        else:
            raise NotImplementedError("unconditional sampling is not a reference path (cond is always the radar cube)")
        rnd = StackedRandomGenerator(device, batch_seeds)
        latents = rnd.randn([batch_size, self.n_latents, self.channels])
        return edm_sampler(self, latents, cond, cond_type, randn_like=rnd.randn_like)


# ---- factories (:452-482) --------------------------------------------------------------------
def kl_d512_m512_l8_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=8, configs=configs)


def kl_d512_m512_l16_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=16, configs=configs)


def kl_d512_m512_l32_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=32, configs=configs)


def kl_d512_m512_l4_d24_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=4, depth=24, configs=configs)


def kl_d512_m512_l8_d24_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=8, depth=24, configs=configs)


def kl_d512_m512_l32_d24_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=32, depth=24, configs=configs)


def kl_d512_m512_l32_d18_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=32, depth=18, configs=configs)


def kl_d512_m512_l32_d12_edm(configs=None):
    return EDMPrecond(n_latents=512, channels=32, depth=12, configs=configs)
