"""Pin the CPU oracle (oracle/rald_oracle.py) against the golden vectors captured from the
reference's own model code (tests/golden/make_golden.py).  CPU only, fp32: rel-L2 <= 1e-5
(SURVEY.md §8d 'fp32 kernels vs oracle')."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, rel_l2
from oracle import rald_oracle as O
from rald_amd import synth, weights

TOL = 1e-5


@pytest.fixture(scope="module")
def sd_d2():
    return weights.make_state_dict(weights.dit_spec(depth=2), seed=0)


@pytest.fixture(scope="module")
def sd_dit():
    return weights.make_state_dict(weights.dit_spec(depth=24), seed=0)


def test_specs_match_reference_state_dict_keys():
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    as_list = lambda spec: [[n, list(s)] for n, s in spec]
    assert as_list(weights.dit_spec(depth=24)) == keys["dit"]
    assert as_list(weights.ae_spec()) == keys["ae"]
    assert as_list(weights.ae_spec(dim=256, num_latents=128)) == keys["ae_tiny"]


def test_g1_ops(sd_d2):
    g = load_golden("g1_ops.npz")
    x = synth.normal([2, 32, 512], 11)
    ctx = synth.cond_tokens(2, 64, 512, seed=12)
    pe = O.positional_embedding(torch.tensor([0.3]))
    assert rel_l2(pe, g["pos_emb"]) < TOL
    t_emb = O.timestep_embed(sd_d2, torch.tensor([0.3]))
    assert rel_l2(t_emb, g["t_emb"]) < TOL
    p = "model.transformer_blocks.0."
    assert rel_l2(O.ada_layer_norm(sd_d2, p + "norm1", x, t_emb), g["adaln"]) < TOL
    assert rel_l2(O.cross_attention(sd_d2, p + "attn1", x), g["self_attn"]) < TOL
    assert rel_l2(O.cross_attention(sd_d2, p + "attn2", x, ctx), g["cross_attn"]) < TOL
    assert rel_l2(O.geglu_ff(sd_d2, p + "ff.net.0.proj", p + "ff.net.2", x), g["ff"]) < TOL
    assert rel_l2(O.transformer_block(sd_d2, p, x, t_emb, ctx), g["block"]) < TOL


def test_g1_depth2_precond_and_100step_sampler(sd_d2):
    g = load_golden("g1_depth2.npz")
    cube = synth.radar_cube(2)
    cond = O.process_radar_cond(sd_d2, cube)
    assert rel_l2(cond, g["cond"]) < TOL
    x = synth.latents([0, 1])
    sig = torch.tensor([1.5, 0.05]).reshape(2, 1, 1)
    assert rel_l2(O.edm_precond(sd_d2, x, sig, cond, depth=2), g["d_x"]) < TOL
    s = O.edm_sampler(lambda xx, ss: O.edm_precond(sd_d2, xx, ss, cond, depth=2),
                      synth.latents([0, 1]), num_steps=100)
    assert rel_l2(s, g["sample100"]) < 1e-4     # 199 compounding fp32 NFEs


def test_g2_transformer(sd_dit):
    g = load_golden("g2_transformer.npz")
    x = synth.latents([0, 1])
    t = torch.tensor([0.25, -1.0])
    out = O.latent_transformer(sd_dit, x, t, synth.cond_tokens(2), depth=24)
    assert rel_l2(out, g["out"]) < TOL


def test_g2_transformer_context_dim_1024():
    sd = weights.make_state_dict(weights.dit_spec(depth=24, context_dim=1024, with_radar=False,
                                                  prefix=""), seed=0)
    g = load_golden("g2_transformer_ctx1024.npz")
    out = O.latent_transformer(sd, synth.latents([0, 1]), torch.tensor([0.25, -1.0]),
                               synth.cond_tokens(2, 64, 1024, seed=778), depth=24, prefix="")
    assert rel_l2(out, g["out"]) < TOL


def test_g7_radar_encoder_and_g3_precond(sd_dit):
    g7 = load_golden("g7_radar_encoder.npz")
    g3 = load_golden("g3_precond.npz")
    cube = synth.radar_cube(2)
    taps = {}
    z = O.radar_encoder(sd_dit, cube[..., 0:1].permute(0, 4, 1, 2, 3), taps=taps)
    assert rel_l2(z, g7["z"]) < TOL
    for name, (mean, amax) in zip(g7["stage_names"], g7["stage_stats"]):
        h = taps[str(name)]
        assert abs(h.mean().item() - mean) < 1e-5 + 1e-4 * abs(mean)
        assert abs(h.abs().max().item() - amax) < 1e-4 * amax
    tokens = O.process_radar_cond(sd_dit, cube)
    assert rel_l2(tokens, g3["cond_tokens"]) < TOL
    x = synth.latents([0, 1])
    for s in (80.0, 1.0, 0.002):
        d = O.edm_precond(sd_dit, x * max(s, 1.0), torch.tensor(s), tokens, depth=24)
        assert rel_l2(d, g3[f"d_sigma_{s}"]) < TOL


@pytest.mark.slow
def test_g4_sample18_as_shipped(sd_dit):
    """EDMPrecond.sample, B=2, seeds {0,1}, 18 Heun steps (35 NFE) - the reference run
    re-encodes the radar cube inside every NFE; the oracle hoists it (bit-identical)."""
    g = load_golden("g4_sample18.npz")
    s = O.dit_sample(sd_dit, synth.radar_cube(2), synth.latents([0, 1]), depth=24)
    assert rel_l2(s, g["sample"]) < 1e-4


def test_g5_autoencoder():
    sd = weights.make_state_dict(weights.ae_spec(), seed=0)
    g = load_golden("g5_ae.npz")
    pc = synth.point_cloud(2, 10000)
    kl, z, mean, logvar = O.ae_encode(sd, pc, g["eps"])
    assert rel_l2(mean, g["mean"]) < TOL and rel_l2(logvar, g["logvar"]) < TOL
    assert rel_l2(z, g["z"]) < TOL and rel_l2(kl, g["kl"]) < TOL
    logits = O.ae_decode(sd, g["z"], synth.queries(2, 4096), depth=24)
    assert rel_l2(logits, g["logits"]) < TOL


def test_g5_tiny_autoencoder():
    """BASELINE config #1: create_autoencoder(dim=256, M=128, N=1000, 'mix'), B=2, CPU."""
    sd = weights.make_state_dict(weights.ae_spec(dim=256, num_latents=128), seed=0)
    g = load_golden("g5_ae_tiny.npz")
    kl, z, _, _ = O.ae_encode(sd, synth.point_cloud(2, 1000), g["eps"])
    logits = O.ae_decode(sd, z, synth.queries(2, 1000), depth=24).squeeze(-1)
    assert rel_l2(kl, g["kl"]) < TOL
    assert rel_l2(logits, g["logits"]) < TOL


def test_g6_edm_loss(sd_d2):
    g = load_golden("g6_edmloss.npz")
    cond = O.process_radar_cond(sd_d2, synth.radar_cube(2))
    y = synth.normal([2, 512, 32], 21)
    loss = O.edm_loss(sd_d2, y, cond, g["rnd_normal"], g["noise"], depth=2)
    assert abs(loss.item() - g["loss"].item()) < 1e-5 * abs(g["loss"].item())


def test_g8_radar_autoencoder_encode_and_keys():
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_radar_ae.json")))["ae_ch64_mult5_n2_d16"]
    spec = weights.radar_autoencoder_spec(64)
    assert [[n, list(s)] for n, s in spec] == keys
    sd = weights.make_state_dict(spec, 0)
    g = load_golden("g8_radar_autoencoder.npz")
    cube = synth.radar_cube(2)
    z = O.radar_encoder(sd, cube.permute(0, 4, 1, 2, 3), prefix="encoder.").permute(0, 2, 3, 4, 1)
    assert rel_l2(z, g["z"]) < TOL


def test_g14_chain_sample_then_decode():
    """BASELINE config #4 as the reference chains it (engine_generation.py:195 -> :204 -> :229-232): the reference's
    sampler output (G4) decoded by the reference's autoencoder, and three batch-1 frames through a depth-2 denoiser."""
    g = load_golden("g14_chain.npz")
    sd_ae = weights.make_state_dict(weights.ae_spec(), seed=0)
    s18 = load_golden("g4_sample18.npz")["sample"]
    logits = O.ae_decode(sd_ae, s18, synth.queries(2, 4096, seed=4243), depth=24).squeeze(-1)
    assert rel_l2(logits, g["logits"]) < TOL
    sd = weights.make_state_dict(weights.dit_spec(depth=2), seed=0)
    qf = synth.queries(1, 2048, seed=4244)
    for i, cs in enumerate(g["frame_cube_seeds"]):
        s = O.dit_sample(sd, synth.radar_cube(1, seed=int(cs)), synth.latents([0]), depth=2)
        assert rel_l2(s[0], g["frame_samples"][i]) < 1e-4
        lg = O.ae_decode(sd_ae, s, qf, depth=24).squeeze(-1)[0]
        assert rel_l2(lg, g["frame_logits"][i]) < 1e-4
    # the frames really differ through the condition alone (same sampler seed)
    assert rel_l2(g["frame_samples"][0], g["frame_samples"][1]) > 5e-2


def test_g15_learnable_query_autoencoder_and_keys():
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    spec = weights.ae_spec(query_type="learnable")
    assert [[n, list(s)] for n, s in spec] == keys["ae_learnable"]
    sd = weights.make_state_dict(spec, seed=0)
    g = load_golden("g15_ae_learnable.npz")
    kl, z, mean, logvar = O.ae_encode(sd, synth.point_cloud(2, 10000), g["eps"])
    assert rel_l2(mean, g["mean"]) < TOL and rel_l2(logvar, g["logvar"]) < TOL
    assert rel_l2(z, g["z"]) < TOL and rel_l2(kl, g["kl"]) < TOL
    assert rel_l2(O.ae_decode(sd, g["z"], synth.queries(2, 4096), depth=24), g["logits"]) < TOL


def test_g16_edm_loss_gradient_norm(sd_d2):
    """EDMLoss forward + backward through denoiser AND radar encoder (autograd of the oracle) against the reference's
    total gradient norm and the norms of its three parameter groups."""
    g6, g = load_golden("g6_edmloss.npz"), load_golden("g16_edmloss_grad.npz")
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "point_embed.basis") for k, v in sd_d2.items()}
    with torch.enable_grad():
        cond = O.process_radar_cond(sd, synth.radar_cube(2))
        loss = O.edm_loss(sd, synth.normal([2, 512, 32], 21), cond, g6["rnd_normal"], g6["noise"], depth=2)
        loss.backward()
    assert abs(loss.item() - g["loss"].item()) < 1e-5 * abs(g["loss"].item())
    sq = {"model": 0.0, "radar_enc": 0.0, "tokeniser": 0.0}
    for n, v in sd.items():
        if v.grad is not None:
            sq["model" if n.startswith("model.") else ("radar_enc" if n.startswith("radar_enc.") else "tokeniser")] += float(v.grad.double().pow(2).sum())
    total = sum(sq.values()) ** 0.5
    assert abs(total - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    for name, ref in zip(g["group_names"], g["group_norms"]):
        assert abs(sq[str(name)] ** 0.5 - float(ref)) < 1e-4 * float(ref) + 1e-7


@pytest.mark.parametrize("tag", ["plain", "peaked"])
def test_g18_autoencoder_on_structured_cloud(tag):
    """Round 3 stress vectors: structured cloud (planes, exact duplicates, +-1 faces), plain and peaked attentions."""
    g = load_golden("g18_ae_stress.npz")
    sd = weights.make_state_dict(weights.ae_spec(), seed=0)
    if tag == "peaked":
        sd = weights.stress_ae_state_dict(sd, out_bias=float(g["peaked_out_bias"]))
    kl, z, mean, logvar = O.ae_encode(sd, synth.structured_cloud(2, 10000), g["eps"])
    assert rel_l2(mean, g[f"{tag}_mean"]) < TOL and rel_l2(logvar, g[f"{tag}_logvar"]) < TOL
    assert rel_l2(z, g[f"{tag}_z"]) < TOL and rel_l2(kl, g[f"{tag}_kl"]) < TOL
    logits = O.ae_decode(sd, g[f"{tag}_z"], synth.structured_queries(2, 4096), depth=24).squeeze(-1)
    ref = g[f"{tag}_logits"]
    assert float((logits - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))      # (peaked logits straddle 0: absolute bound)
    if tag == "peaked":
        assert 0.4 < float((ref > 0).float().mean()) < 0.6


def test_g19_denoiser_with_scaled_output_projections():
    g = load_golden("g19_dit_stress.npz")
    sd = weights.stress_dit_state_dict(weights.make_state_dict(weights.dit_spec(depth=2, with_radar=False, prefix=""), 0))
    cond = synth.cond_tokens(2, seed=781)
    for sigma in (80.0, 1.0):
        x = synth.latents([5, 6]) * max(sigma, 1.0)
        s = torch.tensor(sigma)
        c_skip, c_out, c_in, c_noise = 1 / (s ** 2 + 1), s / (s ** 2 + 1).sqrt(), 1 / (1 + s ** 2).sqrt(), s.log() / 4
        for B in (1, 2):
            F = O.latent_transformer(sd, c_in * x[:B], c_noise.flatten(), cond[:B], depth=2, prefix="")
            assert rel_l2(c_skip * x[:B] + c_out * F, g[f"d_sigma{int(sigma)}_B{B}"]) < TOL


@pytest.mark.skipif(os.environ.get("RALD_LONG_TESTS", "0") != "1", reason="~2 min of CPU: RALD_LONG_TESTS=1 (the GPU tests compare the HIP path with these reference vectors directly)")
def test_g13_g17_long_sampler_horizons():
    for fname, depth, steps in (("g13_sample1000.npz", 2, 1000), ("g17_sample100_depth24.npz", 24, 100)):
        sd = weights.make_state_dict(weights.dit_spec(depth=depth), 0)
        cond = O.process_radar_cond(sd, synth.radar_cube(1))
        s = O.edm_sampler(lambda xx, ss: O.edm_precond(sd, xx, ss, cond, depth=depth), synth.latents([0]), num_steps=steps)
        assert rel_l2(s, load_golden(fname)["sample"]) < 1e-4


def test_g20_radar_autoencoder_forward_and_decode():
    """RadarAutoencoder.forward / decode (reconstruction) against the reference."""
    g = load_golden("g20_radar_autoencoder_forward.npz")
    sd = weights.make_state_dict(weights.radar_autoencoder_spec(64), 0)
    out = O.radar_autoencoder_forward(sd, synth.radar_cube(1))
    assert rel_l2(out["latent"], g["latent"]) < TOL
    pred = out["pred"]
    assert rel_l2(pred[:, ::4, ::4, ::4], g["pred_s4"]) < TOL
    assert abs(float(pred.double().abs().sum()) - float(g["pred_abs_sum"])) < 1e-4 * float(g["pred_abs_sum"])
    assert abs(float(pred.double().pow(2).sum()) - float(g["pred_sq_sum"])) < 1e-4 * float(g["pred_sq_sum"])
