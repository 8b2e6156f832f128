"""Data formats either side of the path: radar cube .bin -> network input, latent
cache .npz, predicted-latent .pt, checkpoint dict.  The radar preprocessing is pinned to the output of the
reference's own ColoRadarDataset.process_radar_data (g10)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def test_oracle_radar_preprocessing_vs_reference_golden():
    from oracle import post_oracle as P
    g = load_golden("g10_radar_cube.npz")
    out = P.process_radar_data(g["raw"].numpy())
    assert out.shape == (128, 64, 32, 2)
    assert np.array_equal(out[::4], g["out_r4"].numpy())
    assert abs(out.astype(np.float64).sum() - float(g["out_sum"])) < 1e-9 * abs(float(g["out_abs_sum"]))


@pytest.mark.gpu
def test_hip_radar_preprocessing_vs_reference_golden(tmp_path):
    from rald_amd import data_formats as F
    g = load_golden("g10_radar_cube.npz")
    raw = g["raw"].numpy()
    path = tmp_path / "cube.bin"
    raw.astype(np.float32).tofile(path)                          # the reference's on-disk layout
    cube = F.load_radarcube(path)
    assert cube.shape == (128, 8, 2, 3)
    out = F.process_radar_data(cube)
    assert out.shape == (128, 64, 32, 2)
    ref = g["out_r4"].numpy()
    err = np.abs(out[::4].cpu().numpy() - ref).max()
    print("radar preprocessing max abs err", err)
    assert err < 2e-6                                            # fp32 lerp; value range [-2.1, 2.1]
    assert abs(float(out.double().sum()) - float(g["out_sum"])) < 1e-6 * float(g["out_abs_sum"])
    batched = F.process_radar_data(np.stack([raw, raw[::-1].copy()]))
    assert batched.shape == (2, 128, 64, 32, 2) and torch.equal(batched[0], out)
    noup = F.process_radar_data(raw, upsample=False)
    assert noup.shape == (128, 8, 2, 2)
    assert np.allclose(noup[..., 0].cpu().numpy(), np.clip(raw[..., 0], 0, 45) / 45, atol=1e-7)


def test_latent_cache_and_checkpoint_round_trip(tmp_path):
    """.npz latent cache readable the way the reference reads it (np.load(...)['res_tokens']); checkpoint
    dict with the reference's keys; EMA list ordered by named_parameters()."""
    from rald_amd import data_formats as F, models_ae as A, weights
    z = torch.randn(512, 32)
    F.save_latent_cache(tmp_path / "frame_0.npz", z)
    assert np.array_equal(np.load(tmp_path / "frame_0.npz")["res_tokens"], z.numpy())
    assert torch.equal(F.load_cached_latent(tmp_path / "frame_0.npz"), z)
    torch.save(z.unsqueeze(0), tmp_path / "pred.pt")
    assert torch.equal(F.load_pred_latent(tmp_path / "pred.pt"), z.unsqueeze(0))
    m = A.create_autoencoder(dim=256, M=128, latent_dim=32, N=1000, query_type="mix")
    m.load_state_dict(weights.make_state_dict(weights.spec_of_state_dict(m.state_dict()), 3))
    ema = [p.detach() * 0.5 for p in m.parameters()]
    F.save_checkpoint(tmp_path / "checkpoint-7.pth", m, ema_params=ema, epoch=7)
    m2 = A.create_autoencoder(dim=256, M=128, latent_dim=32, N=1000, query_type="mix")
    params, ema2, ckpt = F.load_checkpoint(tmp_path / "checkpoint-7.pth", m2, ema=True)
    assert ckpt["epoch"] == 7 and set(ckpt) >= {"model", "model_ema", "epoch"}
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    assert all(torch.equal(a, b) for a, b in zip(ema, ema2)) and len(params) == len(ema2)
