"""Training step of the latent denoiser on the HIP kernels (the reference's
``train_one_epoch`` inner loop, engine_generation.py:90-110, for ``criterion = EDMLoss``):

    sigma = exp(1.2 * N - 1.2); noised = y + sigma * n            (models_radar_generation.py:283-290)
    D = c_skip * noised + c_out * F(c_in * noised, ln(sigma)/4, cond)   (:422-430)
    loss = mean((sigma^2 + 1)/sigma^2 * (D - y)^2)                 (:291-295)
    backward -> clip_grad_norm_(10) -> AdamW -> EMA               (engine_generation.py:96-110)

``DitTrainer`` owns the bf16 compute copies of the fp32 master weights (``FlatAdamW``'s flat storage),
runs forward + backward of ``LatentArrayTransformer`` with the kernels of ``train_ops`` (24 x
``block_forward`` / ``block_backward``, plus the timestep-embedding MLP, the AdaLN linears, proj_in / norm /
proj_out and the loss) and writes every gradient into the parameters' ``.grad`` views of the flat buffer.

``DitTrainer`` covers the transformer and the loss with the radar condition TOKENS as an input (their gradient
is returned); ``EdmTrainer`` adds the radar encoder + tokeniser (``train_encoder.EncoderTrainer``), which the
shipped config trains jointly (``unfreeze_radar_enc``), and the optimizer step: the reference's whole iteration.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import train_ops as TO
from ._handles import _stream, op_gemm_nt, op_layernorm
from ._lib import check, lib

_p = TO._p


def sgemm_acc(A, B, out, trans_a=False, trans_b=False, alpha=1.0, M=None, N=None, K=None):
    """out[m, n] += alpha * sum_k A(m, k) B(n, k) in fp32 (see rald_op_sgemm_acc)."""
    M = M if M is not None else (A.shape[1] if trans_a else A.shape[0])
    K = K if K is not None else (A.shape[0] if trans_a else A.shape[1])
    N = N if N is not None else (B.shape[1] if trans_b else B.shape[0])
    check(lib().rald_op_sgemm_acc(_p(A), A.stride(0), int(trans_a), _p(B), B.stride(0), int(trans_b), _p(out), out.stride(0), M, N, K, alpha,
                                  C.c_void_p(_stream())))
    return out


def linear_f32(x, W, b=None):
    """x [M, K] . W[N, K]^T (+ b) in fp32."""
    out = b.repeat(x.shape[0], 1).contiguous() if b is not None else torch.zeros(x.shape[0], W.shape[0], device=x.device)
    return sgemm_acc(x, W, out)


def silu(x):
    y = torch.empty_like(x)
    check(lib().rald_op_silu_fwd(_p(x), _p(y), x.numel(), C.c_void_p(_stream())))
    return y


def silu_bwd(x_pre, dy):
    dx = torch.empty_like(x_pre)
    check(lib().rald_op_silu_bwd(_p(x_pre), _p(dy), _p(dx), x_pre.numel(), C.c_void_p(_stream())))
    return dx


class DitTrainer:
    """Forward + backward of ``LatentArrayTransformer`` under ``EDMLoss``; gradients go to ``param.grad``."""

    def __init__(self, named_params: Dict[str, torch.nn.Parameter], depth: int, n_heads: int = 8, sigma_data: float = 1.0,
                 p_mean: float = -1.2, p_std: float = 1.2):
        self.P = named_params                      # names as in LatentArrayTransformer.state_dict() (no prefix)
        self.depth, self.H, self.D = depth, n_heads, n_heads * 64
        self.sigma_data, self.p_mean, self.p_std = sigma_data, p_mean, p_std
        dev = next(iter(named_params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("DitTrainer runs on the HIP device only (no CPU fallback)")
        self.dev = dev
        self.W = None
        self.refresh_weights()

    # -- bf16 compute copies of the master weights (call after every optimizer step) ----------------------
    _MATS = (("qkv", "attn1.to_q.weight", 3), ("o", "attn1.to_out.0.weight", 1), ("q2", "attn2.to_q.weight", 1), ("k2", "attn2.to_k.weight", 1),
             ("v2", "attn2.to_v.weight", 1), ("o2", "attn2.to_out.0.weight", 1), ("w1", "ff.net.0.proj.weight", 1), ("w2", "ff.net.2.weight", 1))
    _VECS = (("bo", "attn1.to_out.0.bias"), ("bo2", "attn2.to_out.0.bias"), ("b1", "ff.net.0.proj.bias"), ("b2", "ff.net.2.bias"))

    def _regular_layout(self):
        """(element offsets of block 0's matrices relative to block 0's first one, block stride) when every block's
        parameters sit at the same relative offsets of ONE fp32 buffer (``FlatAdamW``'s layout), else None."""
        if self.depth < 1:
            return None
        key = lambda i, n: self.P[f"transformer_blocks.{i}.{n}"]
        base0 = key(0, self._MATS[0][1])
        store = base0.untyped_storage().data_ptr()
        stride = (key(1, self._MATS[0][1]).data_ptr() - base0.data_ptr()) // 4 if self.depth > 1 else 0
        for i in range(self.depth):
            for _, n, mult in self._MATS:
                p = key(i, n)
                if p.untyped_storage().data_ptr() != store or not p.is_contiguous():
                    return None
                if p.data_ptr() - key(0, n).data_ptr() != i * stride * 4:
                    return None
            q, k, v = key(i, "attn1.to_q.weight"), key(i, "attn1.to_k.weight"), key(i, "attn1.to_v.weight")
            if k.data_ptr() != q.data_ptr() + q.numel() * 4 or v.data_ptr() != k.data_ptr() + k.numel() * 4:
                return None
        return stride

    def refresh_weights(self) -> None:
        sd = {k: p.data for k, p in self.P.items()}
        L = self.depth
        stride = self._regular_layout()
        if stride is None:                                  # arbitrary parameter tensors: per-tensor casts and transposes
            self.W = [TO.prepare_block_weights(sd, f"transformer_blocks.{i}.", self.dev) for i in range(L)]
        else:
            # one strided launch per matrix kind for all blocks: fp32 master -> bf16 [L, rows, cols] and its transpose
            self.W = [dict() for _ in range(L)]
            for key, n, mult in self._MATS:
                p0 = sd[f"transformer_blocks.0.{n}"]
                rows, cols = p0.shape[0] * mult, p0.shape[1]
                wt = TO.transpose(p0, rows, cols, cols, batch=L, stride_in=stride).view(L, cols, rows)          # W^T (bf16)
                w = TO.transpose(wt, cols, rows, rows, batch=L, stride_in=cols * rows).view(L, rows, cols)      # W   (bf16)
                for i in range(L):
                    self.W[i][key], self.W[i][key + "T"] = w[i], wt[i]
            for i in range(L):
                for key, n in self._VECS:
                    self.W[i][key] = sd[f"transformer_blocks.{i}.{n}"]
        self.w_mod = torch.cat([sd[f"transformer_blocks.{i}.norm{j}.linear.weight"] for i in range(L) for j in (1, 2, 3)], 0)
        self.b_mod = torch.cat([sd[f"transformer_blocks.{i}.norm{j}.linear.bias"] for i in range(L) for j in (1, 2, 3)], 0)

    def _grad(self, name: str) -> torch.Tensor:
        p = self.P[name]
        if p.grad is None:
            p.grad = torch.zeros_like(p.data)
        return p.grad

    def _block_grad_views(self, i: int):
        """fp32 destinations for block i's gradients, keyed like ``train_ops.prepare_block_weights``: the parameters'
        own ``.grad`` tensors (views of the optimizer's flat buffer), so the dW GEMMs accumulate in place.  The fused
        q/k/v gradient needs the three ``.grad`` views to be adjacent (they are in ``FlatAdamW``'s layout)."""
        pre, D = f"transformer_blocks.{i}.", self.D
        g = lambda n: self._grad(pre + n)
        views = {"o": g("attn1.to_out.0.weight"), "bo": g("attn1.to_out.0.bias"), "q2": g("attn2.to_q.weight"), "k2": g("attn2.to_k.weight"),
                 "v2": g("attn2.to_v.weight"), "o2": g("attn2.to_out.0.weight"), "bo2": g("attn2.to_out.0.bias"),
                 "w1": g("ff.net.0.proj.weight"), "b1": g("ff.net.0.proj.bias"), "w2": g("ff.net.2.weight"), "b2": g("ff.net.2.bias")}
        gq, gk, gv = g("attn1.to_q.weight"), g("attn1.to_k.weight"), g("attn1.to_v.weight")
        step = gq.numel() * 4
        pending = []
        if (gq.is_contiguous() and gk.data_ptr() == gq.data_ptr() + step and gv.data_ptr() == gk.data_ptr() + step
                and gq.untyped_storage().data_ptr() == gv.untyped_storage().data_ptr()):
            views["qkv"] = torch.as_strided(gq, (3 * D, D), (D, 1))
        else:
            pending = [(pre + f"attn1.to_{n}.weight", "qkv", slice(j * D, (j + 1) * D)) for j, n in enumerate("qkv")]
        return views, pending

    def edm_scalars(self, rnd_normal: torch.Tensor) -> torch.Tensor:
        """[B, 6] device table {sigma, c_in, c_noise, c_skip, c_out, weight} from the log-normal draw (:285-286, :422-425);
        tiny host math, like the reference's python arithmetic on [B,1,1] tensors."""
        return self.edm_scalars_from_sigma(torch.exp(rnd_normal.double().cpu().flatten() * self.p_std + self.p_mean))

    def edm_scalars_from_sigma(self, sigma: torch.Tensor) -> torch.Tensor:
        sd2 = self.sigma_data ** 2
        sigma = sigma.double().cpu().flatten()
        c_skip, c_out = sd2 / (sigma ** 2 + sd2), sigma * self.sigma_data / torch.sqrt(sigma ** 2 + sd2)
        c_in, c_noise = 1.0 / torch.sqrt(sd2 + sigma ** 2), torch.log(sigma) / 4
        weight = (sigma ** 2 + sd2) / (sigma * self.sigma_data) ** 2
        return torch.stack([sigma, c_in, c_noise, c_skip, c_out, weight], 1).to(device=self.dev, dtype=torch.float32).contiguous()

    def forward_backward(self, y: torch.Tensor, cond_tokens: torch.Tensor, rnd_normal: torch.Tensor, noise: torch.Tensor):
        """y [B, N, C] clean latents, cond_tokens [B, T, Cd], rnd_normal [B] and noise [B, N, C] = the two draws of
        EDMLoss (:285, :288).  Accumulates into every ``param.grad``; returns (loss 0-dim device double, dcond [B, T, Cd])."""
        return self.forward_backward_device(y, cond_tokens, self.edm_scalars(rnd_normal), noise)

    def forward_backward_device(self, y: torch.Tensor, cond_tokens: torch.Tensor, scal: torch.Tensor, noise: torch.Tensor):
        """Same with the per-sample scalars already on the device (``edm_scalars``): nothing here touches the host, so
        the whole call can be captured in a hipGraph (``GraphedTrainStep``)."""
        Bn, NL, Cc = y.shape
        M = Bn * NL
        f32 = dict(device=self.dev, dtype=torch.float32)
        y2 = y.reshape(M, Cc).to(**f32).contiguous()
        xn = (y2.view(Bn, NL * Cc) + noise.reshape(Bn, NL * Cc).to(**f32) * scal[:, 0][:, None]).view(M, Cc).contiguous()
        st = self.forward_core(xn, cond_tokens, scal, Bn, NL)
        # ---- loss and its gradient --------------------------------------------------------------------------
        coef3 = scal[:, 3:6].contiguous()
        loss = torch.zeros(1, device=self.dev, dtype=torch.float64)
        dF = torch.empty_like(st["F"])
        check(lib().rald_op_edm_loss_grad(_p(st["F"]), _p(xn), _p(y2), _p(coef3), NL * Cc, M * Cc, _p(dF), _p(None), _p(loss), C.c_void_p(_stream())))
        dcond = self.backward_core(st, dF)
        return loss[0], dcond

    def forward_denoised(self, x: torch.Tensor, cond_tokens: torch.Tensor, sigma: torch.Tensor):
        """EDMPrecond.forward (:412-430) with everything the backward pass needs kept: x [B, N, C] (noised input), sigma one
        value per sample -> (D_x [B, N, C] fp32, state).  ``backward_denoised(state, dD)`` then accumulates the parameter
        gradients of a scalar whose gradient with respect to D_x is ``dD`` (the autograd route: models_radar_generation)."""
        Bn, NL, Cc = x.shape
        scal = self.edm_scalars_from_sigma(sigma)
        xn = x.reshape(Bn * NL, Cc).to(device=self.dev, dtype=torch.float32).contiguous()
        st = self.forward_core(xn, cond_tokens, scal, Bn, NL)
        st["c_out"] = scal[:, 4].contiguous()
        D = scal[:, 3].view(Bn, 1, 1) * xn.view(Bn, NL, Cc) + scal[:, 4].view(Bn, 1, 1) * st["F"].view(Bn, NL, Cc)
        return D, st

    def backward_denoised(self, st, dD: torch.Tensor) -> torch.Tensor:
        Bn, NL = st["Bn"], st["NL"]
        dF = (dD.to(torch.float32).reshape(Bn, -1) * st["c_out"][:, None]).reshape(Bn * NL, -1).contiguous()     # D = c_skip x + c_out F
        return self.backward_core(st, dF)

    def forward_core(self, xn: torch.Tensor, cond_tokens: torch.Tensor, scal: torch.Tensor, Bn: int, NL: int):
        """timestep MLP -> 72 AdaLN linears -> proj_in -> blocks -> norm -> proj_out on xn [B*N, C]; returns the state dict
        (F [B*N, C] and every saved activation)."""
        P, D, H, L, dev = self.P, self.D, self.H, self.depth, self.dev
        Cc = xn.shape[1]
        T = cond_tokens.shape[1]
        M = Bn * NL
        f32 = dict(device=dev, dtype=torch.float32)
        c_in, c_noise = scal[:, 1], scal[:, 2].contiguous()
        xin = (xn.view(Bn, NL * Cc) * c_in[:, None]).view(M, Cc).contiguous()
        # ---- timestep embedding (:217-219) and the 72 AdaLN modulations (:127-131) ---------------------------
        pe = torch.empty(Bn, P["map_layer0.weight"].shape[1], **f32)
        check(lib().rald_op_posemb(_p(c_noise), _p(pe), Bn, pe.shape[1], C.c_void_p(_stream())))
        a0 = linear_f32(pe, P["map_layer0.weight"].data, P["map_layer0.bias"].data)
        e0 = silu(a0)
        a1 = linear_f32(e0, P["map_layer1.weight"].data, P["map_layer1.bias"].data)
        temb = silu(a1)
        mod = linear_f32(temb, self.w_mod, self.b_mod).view(Bn, L * 3, 2 * D)           # [B, 72, 1024]
        # ---- proj_in, blocks, norm, proj_out ------------------------------------------------------------------
        x = torch.zeros(M, D, **f32)
        sgemm_acc(xin, P["proj_in.weight"].data, x)
        cond16 = TO.cast_bf16(cond_tokens.reshape(Bn * T, -1).to(**f32).contiguous())
        saved = []
        for i in range(L):
            saved.append(TO.block_forward(self.W[i], x, mod[:, 3 * i:3 * i + 3], cond16, Bn, NL, H))
        ng, nb = P["norm.weight"].data, P["norm.bias"].data
        yn = op_layernorm(x, ng, nb, gstride=0, rows_per_group=1 << 30, add_one=0.0).float()      # [M, 512]
        F = torch.zeros(M, Cc, **f32)
        sgemm_acc(yn, P["proj_out.weight"].data, F)
        return dict(F=F, x_final=x, yn=yn, xin=xin, pe=pe, a0=a0, e0=e0, a1=a1, temb=temb, mod=mod, cond16=cond16, saved=saved,
                    Bn=Bn, NL=NL, T=T)

    def backward_core(self, st, dF: torch.Tensor) -> torch.Tensor:
        """Backward of forward_core from dF = d(scalar)/dF [B*N, C]: accumulates into every ``param.grad``, returns dcond [B, T, Cd]."""
        P, D, L, dev = self.P, self.D, self.depth, self.dev
        Bn, T = st["Bn"], st["T"]
        x_final, yn, xin, saved, cond16 = st["x_final"], st["yn"], st["xin"], st["saved"], st["cond16"]
        pe, a0, e0, a1, temb = st["pe"], st["a0"], st["e0"], st["a1"], st["temb"]
        M = x_final.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        ng = P["norm.weight"].data
        sgemm_acc(dF, yn, self._grad("proj_out.weight"), trans_a=True, trans_b=True)             # dW_out = dF^T . yn
        dyn = torch.zeros(M, D, **f32)
        sgemm_acc(dF, P["proj_out.weight"].data, dyn, trans_b=True)                              # dyn = dF . W_out
        dx = torch.zeros(M, D, **f32)
        dxb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)               # bf16(dx), kept current by every LayerNorm backward
        TO.ln_mod_bwd(x_final, dyn, ng, 0, 1 << 30, 0.0, dx, self._grad("norm.weight"), self._grad("norm.bias"), dx_bf16=dxb)
        dmod = torch.zeros(Bn, L * 3, 2 * D, **f32)
        dcond = torch.zeros(Bn * T, cond16.shape[1], **f32)
        for i in reversed(range(L)):
            grads, pending = self._block_grad_views(i)
            _, _, dc = TO.block_backward(self.W[i], saved[i], dx, dmod[:, 3 * i:3 * i + 3], grads, dxb=dxb)
            saved[i] = None
            dcond += dc
            for name, key, rows in pending:                 # q/k/v gradients not adjacent in memory: split the fused one
                self._grad(name).add_(grads[key][rows])
        sgemm_acc(dx, xin, self._grad("proj_in.weight"), trans_a=True, trans_b=True)             # dW_in = dx^T . xin
        # AdaLN linears: mod = temb . w_mod^T + b_mod
        dmod2 = dmod.view(Bn, L * 3 * 2 * D)
        gw = torch.zeros_like(self.w_mod)
        sgemm_acc(dmod2, temb, gw, trans_a=True, trans_b=True)                                   # [73728, 512] = dmod^T . temb
        gb = torch.zeros(1, dmod2.shape[1], **f32)
        TO.colsum(dmod2, gb[0])
        for idx, (i, j) in enumerate((i, j) for i in range(L) for j in (1, 2, 3)):
            self._grad(f"transformer_blocks.{i}.norm{j}.linear.weight").add_(gw[idx * 2 * D:(idx + 1) * 2 * D])
            self._grad(f"transformer_blocks.{i}.norm{j}.linear.bias").add_(gb[0, idx * 2 * D:(idx + 1) * 2 * D])
        dtemb = torch.zeros(Bn, D, **f32)
        sgemm_acc(dmod2, self.w_mod, dtemb, trans_b=True)                                        # dmod . w_mod
        da1 = silu_bwd(a1, dtemb)
        sgemm_acc(da1, e0, self._grad("map_layer1.weight"), trans_a=True, trans_b=True)
        TO.colsum(da1, self._grad("map_layer1.bias"))
        de0 = torch.zeros(Bn, D, **f32)
        sgemm_acc(da1, P["map_layer1.weight"].data, de0, trans_b=True)
        da0 = silu_bwd(a0, de0)
        sgemm_acc(da0, pe, self._grad("map_layer0.weight"), trans_a=True, trans_b=True)
        TO.colsum(da0, self._grad("map_layer0.bias"))
        return dcond.view(Bn, T, -1)


class GraphedTrainStep:
    """One training iteration with its ~2 200 kernel launches captured in two hipGraphs (torch.cuda.CUDAGraph on the
    current stream; the library only enqueues on the stream it is given):
      graph A = zero_grad + forward + backward (static input buffers),
      eager   = clip_grad_norm_ + fused AdamW/EMA (3 launches; the bias corrections depend on the step count),
      graph B = refresh of the bf16 weight copies and their transposes.
    Shapes are fixed at construction (the reference's training batches are fixed-size, drop_last=True)."""

    def __init__(self, trainer: DitTrainer, opt, Bn: int, NL: int, Cc: int, T: int, Cd: int):
        self.tr, self.opt = trainer, opt
        dev = trainer.dev
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.y, self.cond, self.noise, self.scal = z(Bn, NL, Cc), z(Bn, T, Cd), z(Bn, NL, Cc), z(Bn, 6)
        self.scal[:, 0] = 1.0
        self.scal[:, 5] = 1.0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                          # warm-up outside capture (lazy allocations, function attributes)
            opt.zero_grad()
            trainer.forward_backward_device(self.y, self.cond, self.scal, self.noise)
            trainer.refresh_weights()
            opt.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_refresh = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_refresh):
            trainer.refresh_weights()                          # trainer.W now lives in the graph's pool, rewritten by every replay
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb):
            opt.zero_grad()
            self.loss, self.dcond = trainer.forward_backward_device(self.y, self.cond, self.scal, self.noise)
        self.g_refresh.replay()

    def __call__(self, y, cond_tokens, rnd_normal, noise, max_norm: float = 10.0, ema_rate: Optional[float] = 0.999, pre_scale: float = 1.0,
                 reducer=None):
        """Returns (loss, total gradient norm) as 0-dim device tensors.  ``reducer`` (``GradReducer``) exchanges the flat
        gradient between ranks after the backward graph; pass its ``pre_scale`` = 1/world."""
        self.y.copy_(y, non_blocking=True)
        self.cond.copy_(cond_tokens, non_blocking=True)
        self.noise.copy_(noise, non_blocking=True)
        self.scal.copy_(self.tr.edm_scalars(rnd_normal), non_blocking=True)
        self.g_fb.replay()
        if reducer is not None:
            reducer.start()
            pre_scale = reducer.finish()
        norm = self.opt.clip_grad_norm_(max_norm, pre_scale=pre_scale)
        self.opt.step(ema_rate=ema_rate)
        self.g_refresh.replay()
        return self.loss, norm


class EdmTrainer:
    """The reference's training iteration for ``EDMPrecond`` with the radar encoder trained jointly (the shipped
    ``unfreeze_radar_enc: true``; engine_generation.py:74-110): cube -> radar encoder -> condition tokens -> denoiser ->
    EDMLoss, backward through all of it, then clip / AdamW / EMA on the flat storage.

        opt = FlatAdamW(model.parameters(), lr=..., ema=True)
        trainer = EdmTrainer(model, opt)
        loss, grad_norm = trainer.step(latents, radar_cube, rnd_normal, noise)

    ``rnd_normal`` [B] and ``noise`` [B, N, C] are EDMLoss's two draws (models_radar_generation.py:285, :288); pass
    ``None`` to draw them here (CPU generator for the log-normal, device generator for the noise)."""

    def __init__(self, model, opt, reducer=None):
        from .train_encoder import EncoderTrainer
        named = dict(model.named_parameters())
        self.model, self.opt, self.reducer = model, opt, reducer
        depth = 1 + max(int(k.split(".")[2]) for k in named if k.startswith("model.transformer_blocks."))
        self.dit = DitTrainer({k[len("model."):]: p for k, p in named.items() if k.startswith("model.")}, depth,
                              sigma_data=float(getattr(model, "sigma_data", 1.0)))
        self.enc = EncoderTrainer({k: p for k, p in named.items() if k.startswith("radar_")})

    def _model_range(self):
        """[lo, hi) of the flat gradient that holds the transformer's gradients (everything under ``model.``): final once
        ``DitTrainer.forward_backward`` returns, i.e. BEFORE the radar encoder's backward pass - the longest part of the iteration
        (44 of 80 ms at B = 8) - under which their exchange between the ranks then runs."""
        if getattr(self, "_mrange", None) is None:
            base = self.opt.flat_g.data_ptr()
            los = [((p.grad.data_ptr() - base) // 4, p.numel()) for k, p in self.model.named_parameters() if k.startswith("model.")]
            self._mrange = (min(l for l, _ in los), max(l + n for l, n in los))
            # the range must hold the transformer's gradients and nothing else: were another parameter's gradient interleaved, its bucket
            # would travel before that gradient is final
            if sum(n for _, n in los) != self._mrange[1] - self._mrange[0]:
                raise RuntimeError("EdmTrainer: the gradients of model.* are not one contiguous range of the flat gradient buffer")
        return self._mrange

    def forward_backward(self, y, cube, rnd_normal, noise):
        tokens = self.enc.forward(cube[..., 0:1].contiguous() if cube.shape[-1] != 1 else cube)
        loss, dtok = self.dit.forward_backward(y, tokens, rnd_normal, noise)
        if self.reducer is not None:
            self.reducer.mark_range(*self._model_range())          # the transformer's buckets travel under the encoder's backward
        self.enc.backward(dtok)
        return loss

    def step(self, y, cube, rnd_normal=None, noise=None, max_norm: float = 10.0, ema_rate: Optional[float] = 0.999):
        if rnd_normal is None:
            rnd_normal = torch.randn(y.shape[0])
        if noise is None:
            noise = torch.randn(y.shape, device=y.device)
        self.opt.zero_grad()
        if self.reducer is not None:
            self.reducer.start()
        loss = self.forward_backward(y, cube, rnd_normal, noise)
        pre = self.reducer.finish() if self.reducer is not None else 1.0
        norm = self.opt.clip_grad_norm_(max_norm, pre_scale=pre)
        self.opt.step(ema_rate=ema_rate)
        self.dit.refresh_weights()
        return loss, norm
