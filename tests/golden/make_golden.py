#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own model code
(/root/reference/model/*.py) on CPU, fp32, with the deterministic name-seeded weights of
rald_amd.weights and the seeded inputs of rald_amd.synth.

Runs only in the build container (the reference never travels).  The reference has no tests
or fixtures of its own, so these outputs are what pins the oracle.

Import recipe: two third-party names the model files import are absent here
(timm.models.layers.DropPath, torch_cluster.fps); they are replaced by in-memory stubs -
DropPath = identity (exact in eval / at drop_path=0), fps = raises (only reached by
query_type='point', not the shipped 'mix' config).

Usage:  python tests/golden/make_golden.py [--only g2,g5] [--skip-long]
"""
import argparse
import json
import os
import sys
import time
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rald_amd import synth, weights  # noqa: E402


def import_reference():
    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert (not self.training) or self.p == 0
            return x

    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = DropPath
    sys.modules["timm"] = types.ModuleType("timm")
    sys.modules["timm.models"] = types.ModuleType("timm.models")
    sys.modules["timm.models.layers"] = tl
    tc = types.ModuleType("torch_cluster")

    def fps(*a, **k):
        raise NotImplementedError("torch_cluster.fps is not available (query_type='point' is out of scope)")

    tc.fps = fps
    sys.modules["torch_cluster"] = tc
    sys.path.insert(0, "/root/reference")
    from model import models_ae, models_radar_encoder, models_radar_generation
    return models_radar_generation, models_ae, models_radar_encoder


class Cfg(dict):
    __getattr__ = dict.__getitem__


CFG = Cfg(cond_type="radar", use_radar_enc=True, unfreeze_radar_enc=True,
          enc_radar_r_dim=8, enc_radar_a_dim=4, enc_radar_e_dim=2, enc_radar_ch=16,
          enc_hidden_ch=64, input_radar_r_dim=128, input_radar_a_dim=64, input_radar_e_dim=32,
          radar_token_channel=512)


def seed_module(m, seed):
    spec = weights.spec_of_state_dict(m.state_dict())
    m.load_state_dict(weights.make_state_dict(spec, seed), strict=True)
    m.eval()
    return spec


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-long", action="store_true")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    want = lambda k: (not only) or (k in only)
    torch.set_num_threads(os.cpu_count())
    G, A, R = import_reference()
    t00 = time.time()

    with torch.no_grad():
        # ---------------- key lists (checkpoint-compat contract) -------------
        if want("keys"):
            dit = G.kl_d512_m512_l32_d24_edm(configs=CFG)
            ae = A.kl_d512_m512_l32_mix(N=10000)
            tiny = A.create_autoencoder(dim=256, M=128, latent_dim=32, N=1000, query_type="mix")
            keys = {"dit": weights.spec_of_state_dict(dit.state_dict()),
                    "ae": weights.spec_of_state_dict(ae.state_dict()),
                    "ae_tiny": weights.spec_of_state_dict(tiny.state_dict())}
            with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
                json.dump(keys, f)
            print("  wrote state_dict_keys.json")
            del dit, ae, tiny

        # ---------------- G1: per-op vectors on a depth-2 denoiser ----------------------
        if want("g1"):
            m = G.EDMPrecond(n_latents=512, channels=32, depth=2, configs=CFG)
            seed_module(m, 0)
            x = synth.normal([2, 32, 512], 11)
            ctx = synth.cond_tokens(2, 64, 512, seed=12)
            c_noise = torch.tensor([0.3], dtype=torch.float32)
            pe = m.model.map_noise(c_noise)
            t_emb = torch.nn.functional.silu(m.model.map_layer1(
                torch.nn.functional.silu(m.model.map_layer0(pe[:, None]))))
            blk = m.model.transformer_blocks[0]
            save("g1_ops.npz", pos_emb=pe, t_emb=t_emb,
                 adaln=blk.norm1(x, t_emb), self_attn=blk.attn1(x), cross_attn=blk.attn2(x, context=ctx),
                 ff=blk.ff(x), block=blk(x, t_emb, context=ctx))
            # depth-2 full forward with per-sample sigma (training-style [B,1,1]) and taps
            xin = synth.latents([0, 1])
            cond = m.process_radar_cond(synth.radar_cube(2))
            sig = torch.tensor([1.5, 0.05]).reshape(2, 1, 1)
            m.process_radar_cond = lambda cube: cond
            d = m(xin, sig, synth.radar_cube(2), "radar")
            s100 = G.edm_sampler(m, synth.latents([0, 1]), synth.radar_cube(2), "radar", num_steps=100)
            save("g1_depth2.npz", d_x=d, cond=cond, sample100=s100)
            del m

        # ---------------- G2: full-depth LatentArrayTransformer --------------------------
        if want("g2"):
            m = G.kl_d512_m512_l32_d24_edm(configs=CFG)
            seed_module(m, 0)
            x = synth.latents([0, 1])
            cond = synth.cond_tokens(2)
            t = torch.tensor([0.25, -1.0], dtype=torch.float32)
            save("g2_transformer.npz", out=m.model(x, t, cond=cond))
            del m
            lt = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64,
                                          depth=24, context_dim=1024)
            seed_module(lt, 0)
            cond1024 = synth.cond_tokens(2, 64, 1024, seed=778)
            save("g2_transformer_ctx1024.npz", out=lt(x, t, cond=cond1024))
            del lt

        # ---------------- G7 + G3: radar encoder, EDMPrecond.forward ----------------------
        if want("g3"):
            m = G.kl_d512_m512_l32_d24_edm(configs=CFG)
            seed_module(m, 0)
            cube = synth.radar_cube(2)
            xr = cube[..., 0:1].permute(0, 4, 1, 2, 3)
            z = m.radar_enc(xr)
            # per-stage statistics of the encoder (hooks on the reference modules)
            stats = {}
            h = m.radar_enc.conv_in(xr)
            stats["conv_in"] = [h.mean().item(), h.abs().max().item()]
            for lvl in range(5):
                for b in range(2):
                    h = m.radar_enc.down[lvl].block[b](h, None)
                    if len(m.radar_enc.down[lvl].attn) > 0:
                        h = m.radar_enc.down[lvl].attn[b](h)
                stats[f"level{lvl}"] = [h.mean().item(), h.abs().max().item()]
                if lvl != 4:
                    h = m.radar_enc.down[lvl].downsample(h)
            save("g7_radar_encoder.npz", z=z, stage_names=np.array(list(stats.keys())),
                 stage_stats=np.array(list(stats.values()), dtype=np.float64))
            tokens = m.process_radar_cond(cube)
            x = synth.latents([0, 1])
            outs = {}
            t0 = time.time()
            for s in (80.0, 1.0, 0.002):
                outs[f"d_sigma_{s}"] = m(x * max(s, 1.0), torch.tensor(s), cube, "radar")
            print(f"  3x EDMPrecond.forward: {time.time() - t0:.1f}s")
            save("g3_precond.npz", cond_tokens=tokens, **outs)
            # ---------------- G4: the sampler as shipped (18 Heun steps = 35 NFE) -------
            if want("g4") and not args.skip_long:
                t0 = time.time()
                s18 = m.sample(cond=cube, batch_seeds=None, cond_type="radar")
                print(f"  EDMPrecond.sample B=2 (as shipped): {time.time() - t0:.1f}s")
                save("g4_sample18.npz", sample=s18)
            del m

        # ---------------- G5: autoencoder -------------------------------------------------
        if want("g5"):
            ae = A.kl_d512_m512_l32_mix(N=10000)
            seed_module(ae, 0)
            pc = synth.point_cloud(2, 10000)
            torch.manual_seed(99)                      # posterior noise: CPU global RNG (:153)
            kl, z = ae.encode(pc)
            torch.manual_seed(99)
            eps = torch.randn(2, 512, 32)              # the same draw the reference consumed
            q = synth.queries(2, 4096)
            logits = ae.decode(z, q)
            # moments for finer-grained checks
            mean_hook, logvar_hook = {}, {}
            h1 = ae.mean_fc.register_forward_hook(lambda m_, i, o: mean_hook.setdefault("v", o))
            h2 = ae.logvar_fc.register_forward_hook(lambda m_, i, o: logvar_hook.setdefault("v", o))
            torch.manual_seed(99)
            ae.encode(pc)
            h1.remove(); h2.remove()
            save("g5_ae.npz", kl=kl, z=z, eps=eps, mean=mean_hook["v"], logvar=logvar_hook["v"],
                 logits=logits)
            del ae
            tiny = A.create_autoencoder(dim=256, M=128, latent_dim=32, N=1000, query_type="mix")
            seed_module(tiny, 0)
            pc = synth.point_cloud(2, 1000)
            q = synth.queries(2, 1000)
            torch.manual_seed(7)
            out = tiny(pc, q)
            torch.manual_seed(7)
            eps = torch.randn(2, 128, 32)
            save("g5_ae_tiny.npz", logits=out["logits"], kl=out["kl"], eps=eps)

        # ---------------- G6: EDMLoss value (training parity anchor) ----------------------
        if want("g6"):
            m = G.EDMPrecond(n_latents=512, channels=32, depth=2, configs=CFG)
            seed_module(m, 0)
            cube = synth.radar_cube(2)
            cond = m.process_radar_cond(cube)
            m.process_radar_cond = lambda c: cond
            y = synth.normal([2, 512, 32], 21)
            torch.manual_seed(5)
            loss = G.EDMLoss()(m, y, cube, "radar")
            torch.manual_seed(5)
            rnd = torch.randn([2, 1, 1])
            noise = torch.randn_like(y)
            save("g6_edmloss.npz", loss=loss, rnd_normal=rnd, noise=noise)
    if want("g8"):
        golden_radar_autoencoder()
    if want("g9"):
        golden_postprocess()
    if want("g10"):
        golden_radar_cube()
    if want("g11"):
        golden_queries()
    if want("g12"):
        golden_optim()
    if want("g14"):
        golden_chain()
    if want("g15"):
        golden_ae_learnable()
    if want("g16"):
        golden_edmloss_grad()
    if "g13" in only or "g17" in only:       # long horizons, minutes of CPU: on request only
        golden_long_horizon(only)
    if want("g18"):
        golden_stress_ae()
    if want("g19"):
        golden_stress_dit()
    if want("g20"):
        golden_radar_autoencoder_forward()
    print(f"done in {time.time() - t00:.0f}s")



def golden_radar_autoencoder():
    """G8: RadarAutoencoder._encode (models_radar_encoder.py:390-393), ae_ch64_mult5_n2_d16."""
    G, A, R = import_reference()
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        m = R.ae_ch64_mult5_n2_d16()
    spec = seed_module(m, 0)
    with open(os.path.join(HERE, "state_dict_keys_radar_ae.json"), "w") as f:
        json.dump({"ae_ch64_mult5_n2_d16": spec}, f)
    with torch.no_grad():
        z = m._encode(synth.radar_cube(2))
    save("g8_radar_autoencoder.npz", z=z)



def golden_radar_autoencoder_forward():
    """G20: RadarAutoencoder.forward (models_radar_encoder.py:395-406) and .decode (:386-388) of
    ae_ch64_mult5_n2_d16 on one seeded cube.  The reconstruction [1,128,64,32,2] is 2 MB: every 4th voxel per axis is stored plus
    whole-tensor checksums; the latent is stored whole (it is also the decoder-only input)."""
    G, A, R = import_reference()
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        m = R.ae_ch64_mult5_n2_d16()
    seed_module(m, 0)
    with torch.no_grad():
        out = m(synth.radar_cube(1))
        pred, z = out["pred"], out["latent"]
        dec = m.decode(z).permute(0, 2, 3, 4, 1)
        assert torch.equal(dec, pred)
    save("g20_radar_autoencoder_forward.npz", latent=z, pred_s4=pred[:, ::4, ::4, ::4], pred_sum=np.float64(pred.double().sum()),
         pred_abs_sum=np.float64(pred.double().abs().sum()), pred_sq_sum=np.float64(pred.double().pow(2).sum()))


def golden_postprocess():
    """G9: the reference's post-processing helpers on seeded synthetic logits / queries / surface
    (utils/utils.py inverse_norm_points, cal_metrics; dataset_preprocessor/lidar.py polar2cartesian -
    the latter module needs `easydict` at import, stubbed in memory like the other absent packages)."""
    sys.path.insert(0, "/root/reference")
    ed = types.ModuleType("easydict"); ed.EasyDict = dict; sys.modules.setdefault("easydict", ed)
    import importlib.util
    import utils.utils as U
    spec = importlib.util.spec_from_file_location("ref_lidar", "/root/reference/dataset_preprocessor/lidar.py")
    L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
    pc_range = [0, -90, -20, 15.8, 90, 20]                       # configs/generation/*_eval.yml:52
    Q = 60000
    q = synth.queries(1, Q, seed=31)[0].numpy()
    # occupancy-like logits: positive in a thin shell so ~4 % of the queries are occupied
    rr = np.linalg.norm(q, axis=1)
    logits = (0.06 - np.abs(rr - 0.9)).astype(np.float32) * 10 + 0.05 * synth.normal([Q], 32).numpy()
    surface = synth.point_cloud(1, 10000, seed=33)[0].numpy()
    ind = np.where(logits > 0)[0]
    pred_polar = U.inverse_norm_points(q[ind], pc_range, True, False)
    gt_polar = U.inverse_norm_points(surface, pc_range, True, False)
    pred, gt = L.polar2cartesian(pred_polar), L.polar2cartesian(gt_polar)
    cd = U.cal_metrics(y_pred=pred, y_gt=gt)
    iso = U.inverse_norm_points(q[:1000], pc_range, False, True)
    labels = (synth.normal([2, 4096], 34).numpy() > 0.3).astype(np.float32)
    outs = synth.normal([2, 4096], 35).numpy()
    predb = (outs >= 0).astype(np.float32)
    acc = (predb == labels).sum(1) / labels.shape[1]
    iou = (predb * labels).sum(1) / ((predb + labels) > 0).sum(1) + 1e-5
    save("g9_postprocess.npz", n_pos=np.int64(len(ind)), ind_head=ind[:64], ind_tail=ind[-64:], pred_head=pred[:256], pred_tail=pred[-256:],
         pred_sum=pred.astype(np.float64).sum(0), gt_head=gt[:256], cd=np.float64(cd), iso_head=iso, acc=acc.astype(np.float32),
         iou=iou.astype(np.float32))



def golden_radar_cube():
    """G10: ColoRadarDataset.process_radar_data (Coloradar_dataset.py:432-475) on a seeded raw cube
    [128,8,2,3]; the dataset module imports once `easydict` is stubbed in memory."""
    sys.path.insert(0, "/root/reference")

    class ED(dict):
        __getattr__ = dict.__getitem__
    ed = types.ModuleType("easydict"); ed.EasyDict = ED; sys.modules["easydict"] = ed
    from datasets.aligned_coloradar.Coloradar_dataset import ColoRadarDataset
    cfg = ED(radar=ED(input_r_dim=128, input_a_dim=8, input_e_dim=2, upsample=True, tgt_r_dim=128, tgt_a_dim=64, tgt_e_dim=32,
                      norm_intensity=True, max_intensity=45, norm_dopp=True, max_dopp=2.4958))
    fake_self = types.SimpleNamespace(config=cfg)
    g = torch.Generator("cpu").manual_seed(41)
    raw = torch.empty(128, 8, 2, 3)
    raw[..., 0] = torch.rand(128, 8, 2, generator=g) * 70 - 10          # dB, some below 0 and above 45
    raw[..., 1] = torch.randn(128, 8, 2, generator=g) * 1.5             # doppler m/s
    raw[..., 2] = (torch.rand(128, 8, 2, generator=g) > 0.3).float()    # validity mask
    out = ColoRadarDataset.process_radar_data(fake_self, raw.numpy().copy())
    # the full output is 2 MB: keep every 4th range bin plus whole-tensor checksums
    save("g10_radar_cube.npz", raw=raw, out_r4=out[::4], out_sum=np.float64(out.astype(np.float64).sum()),
         out_abs_sum=np.float64(np.abs(out).astype(np.float64).sum()))



def golden_queries():
    """G11: the reference's query generation / refine helpers under np.random.seed (utils/utils.py
    generate_query_points, norm_points, remove_points_outside_fov; datasets/utils/query_helper.py
    aug_query_helper; dataset_preprocessor/lidar.py cartesian2polar; the use_cart_query chain is
    engine_generation.py:251-256 restated with the reference's own functions)."""
    sys.path.insert(0, "/root/reference")

    class ED(dict):
        __getattr__ = dict.__getitem__
    ed = types.ModuleType("easydict"); ed.EasyDict = ED; sys.modules.setdefault("easydict", ed)
    import importlib.util
    import utils.utils as U

    def by_path(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        return m
    L = by_path("ref_lidar", "/root/reference/dataset_preprocessor/lidar.py")
    QH = by_path("ref_query_helper", "/root/reference/datasets/utils/query_helper.py")
    pc_range = [0, -90, -20, 15.8, 90, 20]                       # configs/generation/*_eval.yml:52
    pc_range_cart = [0, -15.8, -5.4, 15.8, 15.8, 5.4]            # :46 (commented alternative of the shipped file)
    voxel = [0.05, 0.25, 0.5]                                    # :54
    out = {}
    for tag, aniso, iso in (("aniso", True, False), ("iso", False, True)):
        args = ED(eval=ED(inference=ED(num_query_points=20000)),
                  dataset=ED(lidar=ED(pc_range=pc_range, pc_range_cart=pc_range_cart, norm_anisotropy=aniso, norm_isotropy=iso)))
        np.random.seed(101)
        g = U.generate_query_points(args).astype("float32")
        out[f"uniform_{tag}_head"] = g[:1024]
        out[f"uniform_{tag}_sum"] = g.astype(np.float64).sum(0)
        np.random.seed(102)
        gc = U.generate_query_points(args, coordinate_type="cart")
        gc = U.inverse_norm_points(gc, pc_range_cart, aniso, iso)
        gp = L.cartesian2polar(gc)
        gp = U.norm_points(gp, pc_range, aniso, iso)
        gp = U.remove_points_outside_fov(gp).astype("float32")
        out[f"cart_{tag}_n"] = np.int64(len(gp))
        out[f"cart_{tag}_head"] = gp[:1024]
        out[f"cart_{tag}_tail"] = gp[-256:]
        out[f"cart_{tag}_sum"] = gp.astype(np.float64).sum(0)
        # refine: helper points = un-normalised polar positives, as engine_generation.py:288-297
        helper = U.inverse_norm_points(synth.queries(1, 2000, seed=51)[0].numpy(), pc_range, True, False)
        np.random.seed(103)
        ref = QH.aug_query_helper(helper, 5000, pc_range, np.array(voxel), 10)
        out[f"refine_{tag}"] = U.norm_points(ref, pc_range, aniso, iso)
        if aniso:
            out["refine_raw"] = ref
        np.random.seed(104)
        out[f"refine_trunc_{tag}"] = U.norm_points(QH.aug_query_helper(helper, 1000, pc_range, np.array(voxel), 10), pc_range, aniso, iso)
    save("g11_queries.npz", **out)


def golden_optim():
    """G12: three iterations of the reference's optimizer step on CPU with torch's own pieces -
    torch.nn.utils.clip_grad_norm_(params, 10) (utils/misc.py:262), torch.optim.AdamW(params, lr)
    (main_generation.py:161, all other arguments default) and update_ema(rate=0.999)
    (engine_generation.py:29-40, whose two lines are restated here: the module itself needs open3d)."""
    shapes = [(128, 96), (1024,), (3, 3, 3, 8, 16), (7,), (1,)]
    params = [torch.nn.Parameter(synth.normal(list(s), 900 + i)) for i, s in enumerate(shapes)]
    ema = [p.detach().clone() for p in params]
    opt = torch.optim.AdamW(params, lr=2.5e-4)
    norms = []
    for it in range(3):
        scale = (40.0, 0.01, 3.0)[it]                        # first step clips, the others do not
        for i, p in enumerate(params):
            p.grad = synth.normal(list(p.shape), 1000 + 10 * it + i) * scale
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 10.0)))
        opt.step()
        for t, s in zip(ema, params):
            t.detach().mul_(0.999).add_(s.detach(), alpha=1 - 0.999)
    out = {f"p{i}": p.detach() for i, p in enumerate(params)}
    out.update({f"ema{i}": e for i, e in enumerate(ema)})
    st = opt.state_dict()["state"]
    out.update({f"m{i}": st[i]["exp_avg"] for i in range(len(params))})
    out.update({f"v{i}": st[i]["exp_avg_sq"] for i in range(len(params))})
    save("g12_optim.npz", norms=np.array(norms, np.float64), **out)


def golden_long_horizon(only):
    """Long sampler horizons FROM THE REFERENCE (VERDICT r02 missing #3), radar condition hoisted out of the loop (bit-identical,
    SURVEY.md section 0 row 9; as golden_chain does):
      G13  BASELINE config #5: edm_sampler(num_steps=1000) = 1999 NFE, depth-2 model, B = 1 (round 2: generated by the oracle);
      G17  BASELINE config #3: edm_sampler(num_steps=100) = 199 NFE through the SHIPPED depth (24 blocks), B = 1."""
    G, A, R = import_reference()
    with torch.no_grad():
        for key, depth, steps, fname in (("g13", 2, 1000, "g13_sample1000.npz"), ("g17", 24, 100, "g17_sample100_depth24.npz")):
            if key not in only:
                continue
            m = G.EDMPrecond(n_latents=512, channels=32, depth=depth, configs=CFG)
            seed_module(m, 0)
            cube = synth.radar_cube(1)
            cond = m.process_radar_cond(cube)
            m.process_radar_cond = lambda c, _cond=cond: _cond
            t0 = time.time()
            s = G.edm_sampler(m, synth.latents([0]), cube, "radar", num_steps=steps).to(torch.float32)
            print(f"  reference edm_sampler depth {depth}, {steps} steps: {time.time() - t0:.0f}s")
            save(fname, sample=s)
            del m


def golden_stress_ae():
    """G18: the reference's KLAutoEncoder on inputs / weights that stress the folded kernels:
      a) a STRUCTURED cloud (synth.structured_cloud: planes, exact duplicates, points on the +-1 faces) and structured queries,
         plain seeded weights;
      b) the same inputs with PEAKED attentions (weights.stress_ae_state_dict: q / kv projections of the three point / query
         attentions x 4, PointEmbed bias + 0.5) and `to_outputs.bias` set to minus the median logit, so that the occupancy decision
         `logit > 0` (engine_generation.py:229-232) splits the queries in half instead of being all-positive."""
    G, A, R = import_reference()
    with torch.no_grad():
        ae = A.kl_d512_m512_l32_mix(N=10000)
        spec = seed_module(ae, 0)
        sd0 = weights.make_state_dict(spec, 0)
        pc = synth.structured_cloud(2, 10000)
        q = synth.structured_queries(2, 4096)
        out = {}
        for tag, sd in (("plain", sd0), ("peaked", weights.stress_ae_state_dict(sd0))):
            ae.load_state_dict(sd, strict=True)
            mean_hook, logvar_hook = {}, {}
            h1 = ae.mean_fc.register_forward_hook(lambda m_, i, o: mean_hook.setdefault("v", o))
            h2 = ae.logvar_fc.register_forward_hook(lambda m_, i, o: logvar_hook.setdefault("v", o))
            torch.manual_seed(99)
            kl, z = ae.encode(pc)
            h1.remove(); h2.remove()
            logits = ae.decode(z, q).squeeze(-1)
            if tag == "peaked":
                bias = float(sd["to_outputs.bias"][0]) - float(logits.median())
                logits = logits - float(logits.median())
                out["peaked_out_bias"] = np.float32(bias)
            out.update({f"{tag}_kl": kl, f"{tag}_z": z, f"{tag}_mean": mean_hook["v"], f"{tag}_logvar": logvar_hook["v"], f"{tag}_logits": logits})
            print(f"  {tag}: logits in [{float(logits.min()):.2f}, {float(logits.max()):.2f}], {float((logits > 0).float().mean()) * 100:.1f} % positive, "
                  f"|mean| max {float(mean_hook['v'].abs().max()):.2f}")
        torch.manual_seed(99)
        out["eps"] = torch.randn(2, 512, 32)
        save("g18_ae_stress.npz", **out)


def golden_stress_dit():
    """G19 (VERDICT r02 next #2 iii): a depth-2 denoiser whose to_out / ff.net.2 weights are 8 x the seeded ones
    (weights.stress_dit_state_dict) at the sampler's first noise level sigma = 80 and at sigma = 1, B = 1 and B = 2 - the batches
    whose per-head / split-K partial sums travel as fp16 x 2^-6 slabs on the HIP side.  Condition tokens given (LatentArrayTransformer
    + EDM pre / post-conditioning restated from EDMPrecond.forward :418-430 on the reference's own transformer)."""
    G, A, R = import_reference()
    with torch.no_grad():
        lt = G.LatentArrayTransformer(in_channels=32, t_channels=256, n_heads=8, d_head=64, depth=2)
        spec = seed_module(lt, 0)
        lt.load_state_dict(weights.stress_dit_state_dict(weights.make_state_dict(spec, 0)), strict=True)
        cond = synth.cond_tokens(2, seed=781)
        out = {}
        for sigma in (80.0, 1.0):
            x = synth.latents([5, 6]) * max(sigma, 1.0)
            s = torch.tensor(sigma)
            c_skip, c_out, c_in, c_noise = 1 / (s ** 2 + 1), s / (s ** 2 + 1).sqrt(), 1 / (1 + s ** 2).sqrt(), s.log() / 4
            for B in (1, 2):
                F = lt(c_in * x[:B], c_noise.flatten(), cond=cond[:B])
                out[f"d_sigma{int(sigma)}_B{B}"] = c_skip * x[:B] + c_out * F
            print(f"  sigma {sigma}: |F| max {float(F.abs().max()):.1f}, rms {float(F.pow(2).mean().sqrt()):.2f}")
        save("g19_dit_stress.npz", **out)


def golden_chain():
    """G14 (BASELINE config #4, engine_generation.py:195 -> :204/:275/:300 -> :229-232): the reference's own chain
    radar cube -> EDMPrecond.sample -> KLAutoEncoder.decode -> logits (occupied iff > 0).
      * full depth: the sampler output of G4 (EDMPrecond.sample as shipped, depth 24, B = 2; regenerating it bit for bit is
        what `--only g3,g4` does) is decoded by the reference's kl_d512_m512_l32_mix on 4 096 seeded queries;
      * three frames at batch 1 (the reference's eval_batch_size) through a depth-2 denoiser: different cubes, the SAME
        sampler seed (batch_seeds=None -> seed 0 for every frame), so the frames differ only through the radar condition -
        the vectors behind the test that frees frame k's tensors before frame k+1 is allocated.  The radar encoder is
        hoisted out of the sampler loop here (bit-identical; as G1 does)."""
    G, A, R = import_reference()
    with torch.no_grad():
        ae = A.kl_d512_m512_l32_mix(N=10000)
        seed_module(ae, 0)
        s18 = torch.from_numpy(np.load(os.path.join(HERE, "g4_sample18.npz"))["sample"])
        q = synth.queries(2, 4096, seed=4243)
        logits = ae.decode(s18, q).squeeze(-1)
        m = G.EDMPrecond(n_latents=512, channels=32, depth=2, configs=CFG)
        seed_module(m, 0)
        raw_cond = m.process_radar_cond
        frames, flogits = [], []
        qf = synth.queries(1, 2048, seed=4244)
        for cube_seed in (1234, 555, 909):
            cube = synth.radar_cube(1, seed=cube_seed)
            cond = raw_cond(cube)
            m.process_radar_cond = lambda c, _cond=cond: _cond
            s = m.sample(cond=cube, batch_seeds=None, cond_type="radar").to(torch.float32)
            frames.append(s[0])
            flogits.append(ae.decode(s, qf).squeeze(-1)[0])
        save("g14_chain.npz", logits=logits, frame_samples=torch.stack(frames), frame_logits=torch.stack(flogits),
             frame_cube_seeds=np.array([1234, 555, 909], np.int64))


def golden_ae_learnable():
    """G15: query_type='learnable' (models_ae.py:325-326, :378-379; factory kl_d512_m512_l32_learn): encode moments / z /
    kl and decode logits, plus its state_dict key list."""
    G, A, R = import_reference()
    with torch.no_grad():
        ae = A.create_autoencoder(dim=512, M=512, latent_dim=32, N=10000, query_type="learnable")
        spec = seed_module(ae, 0)
        path = os.path.join(HERE, "state_dict_keys.json")
        keys = json.load(open(path))
        keys["ae_learnable"] = spec
        with open(path, "w") as f:
            json.dump(keys, f)
        pc = synth.point_cloud(2, 10000)
        mean_hook, logvar_hook = {}, {}
        h1 = ae.mean_fc.register_forward_hook(lambda m_, i, o: mean_hook.setdefault("v", o))
        h2 = ae.logvar_fc.register_forward_hook(lambda m_, i, o: logvar_hook.setdefault("v", o))
        torch.manual_seed(99)
        kl, z = ae.encode(pc)
        h1.remove(); h2.remove()
        torch.manual_seed(99)
        eps = torch.randn(2, 512, 32)
        logits = ae.decode(z, synth.queries(2, 4096))
        save("g15_ae_learnable.npz", kl=kl, z=z, eps=eps, mean=mean_hook["v"], logvar=logvar_hook["v"], logits=logits)


def golden_edmloss_grad():
    """G16 (the grad-norm SURVEY.md section 8c asks for beside G6): EDMLoss forward + backward through the depth-2
    EDMPrecond WITH its jointly trained radar encoder (not hoisted: gradients flow through it), same model / inputs / seed
    as G6; the total gradient norm (what clip_grad_norm_ returns, utils/misc.py:262) and the norms of the three parameter
    groups."""
    G, A, R = import_reference()
    m = G.EDMPrecond(n_latents=512, channels=32, depth=2, configs=CFG)
    seed_module(m, 0)
    m.train()                      # the reference trains in train mode; dropout is 0 and drop_path 0, so eval == train here
    cube = synth.radar_cube(2)
    y = synth.normal([2, 512, 32], 21)
    torch.manual_seed(5)
    loss = G.EDMLoss()(m, y, cube, "radar")
    loss.backward()
    sq = {"model": 0.0, "radar_enc": 0.0, "tokeniser": 0.0}
    for n, p in m.named_parameters():
        g2 = float(p.grad.double().pow(2).sum()) if p.grad is not None else 0.0
        key = "model" if n.startswith("model.") else ("radar_enc" if n.startswith("radar_enc.") else "tokeniser")
        sq[key] += g2
    total = float(np.sqrt(sum(sq.values())))
    save("g16_edmloss_grad.npz", loss=loss.detach(), grad_norm=np.float64(total),
         group_names=np.array(list(sq.keys())), group_norms=np.sqrt(np.array(list(sq.values()), np.float64)))


if __name__ == "__main__":
    main()
