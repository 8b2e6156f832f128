"""Drop-in for the reference's ``model/models_radar_encoder.py`` as far as the hot path reaches:
``RadarAutoencoder`` with the reference's factories (``ae_ch64_mult5_n2_d16`` ..., looked up via
``models_radar_encoder.__dict__[name]()``, main_generation.py:134) and ``state_dict`` keys
(``encoder.*`` + ``decoder.*``, so checkpoints load with ``strict=True``), whose
``encode`` / ``_encode`` (:382-393, the frozen-encoder route of engine_generation.py:87, :191) run
the HIP radar-spectrum encoder and whose ``decode`` / ``forward`` (:386-388, :395-406: the reconstruction used to
pre-train the radar autoencoder) run the HIP decoder - the same implicit-GEMM Conv3d / GroupNorm / attention kernels,
with a nearest-neighbour up-sampling kernel between the levels.  Inference only (not differentiable).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import weights as _w
from ._handles import _f32c, _need_cuda, _ptr, _stream
from ._lib import check, lib
from .models_radar_generation import _HipBacked, build_param_tree


class RadarAutoencoder(_HipBacked):
    def __init__(self, *, basic_channel=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, embed_dim=16):
        super().__init__()
        if tuple(ch_mult) != (1, 1, 2, 2, 4) or num_res_blocks != 2:
            raise NotImplementedError("only the reference's ch_mult=(1,1,2,2,4), num_res_blocks=2 is built")
        if basic_channel % 64 != 0:
            raise NotImplementedError("basic_channel must be a multiple of 64 (MFMA K-step); the reference ships 64 and 128")
        self.basic_channel, self.embed_dim = basic_channel, embed_dim
        build_param_tree(self, _w.radar_autoencoder_spec(basic_channel, embed_dim))
        self._hip = None
        self._hip_fp = None

    def _handle(self):
        fp = self._state_fingerprint()
        if self._hip is None or self._hip_fp != fp:
            if self._hip is not None:
                lib().rald_radar_destroy(self._hip)
            h = C.c_void_p()
            check(lib().rald_radar_create(self.basic_channel, self.embed_dim, 2, 128, 64, 32, C.byref(h)))
            for k, v in self.state_dict().items():
                t = _f32c(v)
                if k.startswith("encoder."):
                    check(lib().rald_radar_load_weight(h, k[len("encoder."):].encode(), C.c_void_p(_ptr(t)), t.numel()))
                elif k.startswith("decoder."):
                    check(lib().rald_radar_load_decoder_weight(h, k[len("decoder."):].encode(), C.c_void_p(_ptr(t)), t.numel()))
            check(lib().rald_radar_finalize(h))
            self._hip, self._hip_fp = h, fp
        return self._hip

    def __del__(self):
        try:
            if self._hip is not None:
                lib().rald_radar_destroy(self._hip)
        except Exception:
            pass

    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        """cube [B,R,A,E,2] -> [B,R/16,A/16,E/16,embed_dim]  (:390-393)."""
        _need_cuda(x, "radar cube")
        x = _f32c(x)
        if x.shape[1:] != (128, 64, 32, 2):
            raise RuntimeError(f"radar cube must be [B,128,64,32,2], got {tuple(x.shape)}")
        B = x.shape[0]
        z = torch.empty(B, 8, 4, 2, self.embed_dim, device=x.device, dtype=torch.float32)
        check(lib().rald_radar_encode(self._handle(), C.c_void_p(_ptr(x)), B, C.c_void_p(_ptr(z)), C.c_void_p(_stream())))
        return z

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,2,R,A,E] (channels first, as Encoder.forward takes it) -> [B,embed_dim,R/16,A/16,E/16] (:382-384)."""
        return self._encode(x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)

    def _decode_cl(self, z_cl: torch.Tensor) -> torch.Tensor:
        """z [B,R/16,A/16,E/16,embed_dim] (channels last) -> reconstruction [B,R,A,E,2]."""
        _need_cuda(z_cl, "radar latent")
        z_cl = _f32c(z_cl)
        if z_cl.shape[1:] != (8, 4, 2, self.embed_dim):
            raise RuntimeError(f"radar latent must be [B,8,4,2,{self.embed_dim}] (channels last), got {tuple(z_cl.shape)}")
        B = z_cl.shape[0]
        out4 = torch.empty(B, 128, 64, 32, 4, device=z_cl.device, dtype=torch.float32)      # 2 channels + the kernel's zero padding
        check(lib().rald_radar_decode(self._handle(), C.c_void_p(_ptr(z_cl)), B, C.c_void_p(_ptr(out4)), C.c_void_p(_stream())))
        return out4[..., :2].contiguous()

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B,embed_dim,R/16,A/16,E/16] (channels first, as Decoder.forward takes it) -> [B,2,R,A,E]  (:386-388)."""
        return self._decode_cl(z.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)

    @torch.no_grad()
    def forward(self, inputs: torch.Tensor):
        """inputs [B,R,A,E,2] -> {'pred': [B,R,A,E,2], 'latent': [B,embed_dim,R/16,A/16,E/16]}  (:395-406)."""
        z_cl = self._encode(inputs)
        return {"pred": self._decode_cl(z_cl), "latent": z_cl.permute(0, 4, 1, 2, 3)}


def create_autoencoder(basic_channel=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, embed_dim=16):
    return RadarAutoencoder(basic_channel=basic_channel, ch_mult=ch_mult, num_res_blocks=num_res_blocks, embed_dim=embed_dim)


def ae_ch128_mult5_n2_d16():
    return create_autoencoder(basic_channel=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, embed_dim=16)


def ae_ch64_mult5_n2_d16():
    return create_autoencoder(basic_channel=64, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, embed_dim=16)


def ae_ch16_mult5_n2_d16():
    return create_autoencoder(basic_channel=16, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, embed_dim=16)
