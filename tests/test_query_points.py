"""Query generation + refine.  CPU: the numpy oracle against outputs of the
reference's own functions under np.random.seed (g11).  GPU: the HIP kernels (rald_query_*) replaying
the same numpy draws - bit-exact for the uniform / refine / norm arithmetic (add, mul, div, clip in
numpy's dtype promotion), <= 1 float32 ulp for the cartesian chain (device atan2/asin vs libm) - plus
the device-generator variant checked through distribution properties."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from rald_amd import synth

PC_RANGE = [0, -90, -20, 15.8, 90, 20]
PC_RANGE_CART = [0, -15.8, -5.4, 15.8, 15.8, 5.4]
VOXEL = [0.05, 0.25, 0.5]
TAGS = (("aniso", True, False), ("iso", False, True))


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def _args(n, aniso, iso, aug_num=5000, scale=10):
    return _ns(eval=_ns(inference=_ns(num_query_points=n, refine_query_aug_num=aug_num, refine_query_scale=scale)),
               dataset=_ns(lidar=_ns(pc_range=PC_RANGE, pc_range_cart=PC_RANGE_CART, voxel_size=VOXEL, norm_anisotropy=aniso,
                                     norm_isotropy=iso)))


def _helper():
    from oracle import post_oracle as P
    return P.inverse_norm_points(synth.queries(1, 2000, seed=51)[0].numpy(), PC_RANGE, True, False)


def _refine_draws(N, aug_num, scale):
    gen = aug_num - N
    sel = np.random.choice(N, size=gen, replace=True)
    scales = np.random.choice(np.arange(scale, step=1) + 1, size=gen)
    return sel, scales, np.random.rand(gen, 3)


def test_oracle_vs_reference_golden():
    from oracle import post_oracle as P
    g = load_golden("g11_queries.npz")
    helper = _helper()
    for tag, aniso, iso in TAGS:
        np.random.seed(101)
        q = P.queries_from_uniform(np.random.random_sample((3, 20000)), PC_RANGE, aniso, iso).astype(np.float32)
        assert np.array_equal(q[:1024], g[f"uniform_{tag}_head"].numpy())
        assert np.array_equal(q.astype(np.float64).sum(0), g[f"uniform_{tag}_sum"].numpy())
        np.random.seed(102)
        c = P.cart_queries_from_uniform(np.random.random_sample((3, 20000)), PC_RANGE_CART, PC_RANGE, aniso, iso)
        assert len(c) == int(g[f"cart_{tag}_n"])
        assert np.array_equal(c[:1024], g[f"cart_{tag}_head"].numpy()) and np.array_equal(c[-256:], g[f"cart_{tag}_tail"].numpy())
        np.random.seed(103)
        raw = P.aug_query_helper_from_draws(helper, 5000, PC_RANGE, VOXEL, *_refine_draws(2000, 5000, 10))
        if aniso:
            assert np.array_equal(raw, g["refine_raw"].numpy())
        assert np.array_equal(P.norm_points(raw, PC_RANGE, aniso, iso), g[f"refine_{tag}"].numpy())
        np.random.seed(104)                                          # N >= aug_num: truncation, no draws
        t = P.norm_points(P.aug_query_helper_from_draws(helper, 1000, PC_RANGE, VOXEL, None, None, None), PC_RANGE, aniso, iso)
        assert np.array_equal(t, g[f"refine_trunc_{tag}"].numpy())


@pytest.mark.gpu
def test_hip_queries_vs_reference_golden():
    from rald_amd import query_points as QP
    g = load_golden("g11_queries.npz")
    helper = torch.from_numpy(_helper()).cuda()
    for tag, aniso, iso in TAGS:
        args = _args(20000, aniso, iso)
        np.random.seed(101)
        q = QP.generate_query_points(args).cpu().numpy()
        assert q.dtype == np.float32 and q.shape == (20000, 3)
        assert np.array_equal(q[:1024], g[f"uniform_{tag}_head"].numpy())                       # bit-exact
        assert np.array_equal(q.astype(np.float64).sum(0), g[f"uniform_{tag}_sum"].numpy())
        np.random.seed(102)
        c = QP.generate_cart_query_points(args).cpu().numpy()
        assert c.shape[0] == int(g[f"cart_{tag}_n"])
        for got, want in ((c[:1024], g[f"cart_{tag}_head"].numpy()), (c[-256:], g[f"cart_{tag}_tail"].numpy())):
            ulp = np.spacing(np.abs(want).astype(np.float32))
            print(tag, "cart chain: max diff in ulp", (np.abs(got - want) / ulp).max(), "mismatching", (got != want).mean())
            assert np.all(np.abs(got - want) <= ulp)
        assert np.allclose(c.astype(np.float64).sum(0), g[f"cart_{tag}_sum"].numpy(), rtol=1e-9, atol=1e-6)
        np.random.seed(103)
        r = QP.refine_queries(helper, args).cpu().numpy()
        assert np.array_equal(r, g[f"refine_{tag}"].numpy())                                    # bit-exact
        if aniso:
            np.random.seed(103)
            raw = QP.aug_query_helper(helper, 5000, PC_RANGE, VOXEL, 10).cpu().numpy()
            assert np.array_equal(raw, g["refine_raw"].numpy())
        np.random.seed(104)
        t = QP.refine_queries(helper, _args(20000, aniso, iso, aug_num=1000)).cpu().numpy()
        assert np.array_equal(t, g[f"refine_trunc_{tag}"].numpy())
        n = QP.norm_points(helper, PC_RANGE, aniso, iso).cpu().numpy()
        from oracle import post_oracle as P
        assert np.array_equal(n, P.norm_points(helper.cpu().numpy(), PC_RANGE, aniso, iso))


@pytest.mark.gpu
def test_hip_queries_full_size_properties_and_device_rng():
    """The shipped sizes (num_query_points = refine_query_aug_num = 500 000, scale 10) with a device
    generator: box / clip bounds, helper points preserved, jitter within scale * voxel, uniform moments;
    and the numpy-stream path against the oracle at full size."""
    from oracle import post_oracle as P
    from rald_amd import query_points as QP
    args = _args(500000, True, False, aug_num=500000)
    rng = torch.Generator("cuda").manual_seed(7)
    q = QP.generate_query_points(args, rng=rng)
    assert q.shape == (500000, 3) and q.dtype == torch.float32
    assert float(q.min()) >= -1 and float(q.max()) <= 1
    assert torch.all(q.mean(0).abs() < 5e-3) and torch.all((q.var(0) - 1 / 3).abs() < 5e-3)
    np.random.seed(9)
    u = np.random.random_sample((3, 500000))
    np.random.seed(9)
    assert np.array_equal(QP.generate_query_points(args).cpu().numpy(), P.queries_from_uniform(u, PC_RANGE, True, False).astype(np.float32))
    np.random.seed(9)
    assert np.array_equal(QP.generate_cart_query_points(_args(500000, False, True)).cpu().numpy().shape,
                          P.cart_queries_from_uniform(u, PC_RANGE_CART, PC_RANGE, False, True).shape)
    helper = torch.from_numpy(_helper()).cuda()
    r = QP.aug_query_helper(helper, 500000, PC_RANGE, VOXEL, 10, rng=rng)
    assert torch.equal(r[:2000], helper)
    lo, hi = torch.tensor(PC_RANGE[:3], device="cuda"), torch.tensor(PC_RANGE[3:], device="cuda")
    assert torch.all(r >= lo) and torch.all(r <= hi)
    # every augmented point lies within scale * voxel of SOME helper point (checked on a sample, per axis box)
    s = r[2000:][torch.randperm(498000, device="cuda")[:512]]
    d = (s[:, None, :] - helper[None, :, :]).abs()
    bound = torch.tensor(VOXEL, device="cuda") * 10 + 1e-4
    clipped = ((s <= lo + 1e-6) | (s >= hi - 1e-6))[:, None, :]
    assert torch.all(((d <= bound) | clipped).all(dim=2).any(dim=1))
    rn = QP.refine_queries(helper, args, rng=rng)
    assert float(rn.min()) >= -1 - 1e-6 and float(rn.max()) <= 1 + 1e-6
    np.random.seed(11)
    draws = _refine_draws(2000, 500000, 10)
    np.random.seed(11)
    want = P.norm_points(P.aug_query_helper_from_draws(helper.cpu().numpy(), 500000, PC_RANGE, VOXEL, *draws), PC_RANGE, True, False)
    assert np.array_equal(QP.refine_queries(helper, args).cpu().numpy(), want)


@pytest.mark.gpu
def test_hip_queries_edge_cases():
    from rald_amd import query_points as QP
    args = _args(0, True, False, aug_num=0)
    assert QP.generate_query_points(args).shape == (0, 3)
    assert QP.generate_cart_query_points(args).shape == (0, 3)
    empty = torch.empty(0, 3, device="cuda")
    assert QP.norm_points(empty, PC_RANGE, True, False).shape == (0, 3)
    assert QP.refine_queries(empty, args).shape == (0, 3)
    with pytest.raises(ValueError):                                  # np.random.choice(0, ...) raises in the reference
        QP.aug_query_helper(empty, 10, PC_RANGE, VOXEL, 2)
    with pytest.raises(ValueError):
        QP.uniform_queries(10, PC_RANGE, False, False, device="cuda")
    with pytest.raises(ValueError):
        QP.generate_query_points(args, coordinate_type="spherical")
    with pytest.raises(RuntimeError):
        QP.norm_points(torch.zeros(4, 3), PC_RANGE, True, False)     # CPU tensor: no fallback
    one = torch.tensor([[3.0, 10.0, -5.0]], device="cuda")
    out = QP.aug_query_helper(one, 1, PC_RANGE, VOXEL, 2)             # N == aug_num: copy
    assert torch.equal(out, one)
    # neither flag: norm_points returns zeros, like np.zeros_like in the reference
    assert torch.count_nonzero(QP.norm_points(one, PC_RANGE, False, False)) == 0


@pytest.mark.gpu
def test_infer_point_cloud_tail_vs_oracle_replay():
    """engine_generation.py:250-322 on the device (queries + helper points -> decode -> positives -> refine
    -> decode -> positives -> cartesian -> Chamfer) against a replay in which every NON-decode step is the
    numpy oracle fed with the same numpy draws; the decode logits come from the same HIP autoencoder (its
    own parity is tests/test_gpu_ae.py), so the comparison isolates the tail."""
    from oracle import post_oracle as P
    from rald_amd import engine_generation as E, models_ae as A, weights
    z = synth.latents([3])[:, :128].contiguous().cuda()

    def build(bias_shift):
        m = A.create_autoencoder(dim=256, M=128, latent_dim=32, N=1000, query_type="mix")
        sd = weights.make_state_dict(weights.spec_of_state_dict(m.state_dict()), 0)
        sd["to_outputs.bias"] = sd["to_outputs.bias"] - bias_shift
        m.load_state_dict(sd, strict=True)
        return m.cuda().eval()
    # random weights give one-signed logits: centre them so that a few per cent of the queries are occupied
    probe = build(0.0).decode(z, synth.queries(1, 4096, seed=60).cuda()).flatten()
    vae = build(float(torch.quantile(probe, 0.95)))
    helper_norm = synth.queries(1, 700, seed=61)[0]
    surface = synth.point_cloud(1, 3000, seed=62)[0]
    args = _args(30000, True, False, aug_num=20000, scale=10)
    args.eval.inference.query_helper = True
    args.eval.inference.refine_query = True
    args.dataset.lidar.view_cone_mode = True
    np.random.seed(5)
    res = E.infer_point_cloud(vae, z, args, helper_points=helper_norm, surface=surface.cuda())

    np.random.seed(5)
    grid = P.queries_from_uniform(np.random.random_sample((3, 30000)), PC_RANGE, True, False).astype(np.float32)
    grid = np.concatenate((grid, helper_norm.numpy()), axis=0)
    logits = vae.decode(z, torch.from_numpy(grid).cuda()[None]).squeeze(-1)[0].cpu().numpy()
    pred = P.inverse_norm_points(grid[np.where(logits > 0)[0]], PC_RANGE, True, False)
    assert 0 < len(pred) < 20000                                     # the refine pass has to draw
    refined = P.norm_points(P.aug_query_helper_from_draws(pred, 20000, PC_RANGE, VOXEL, *_refine_draws(len(pred), 20000, 10)),
                            PC_RANGE, True, False)
    logits_r = vae.decode(z, torch.from_numpy(refined).cuda()[None]).squeeze(-1)[0].cpu().numpy()
    pred = P.polar2cartesian(P.inverse_norm_points(refined[np.where(logits_r > 0)[0]], PC_RANGE, True, False))
    gt = P.polar2cartesian(P.inverse_norm_points(surface.numpy(), PC_RANGE, True, False))
    got = res["pred"].cpu().numpy()
    assert res["n_queries"] == 30700 + 20000
    assert got.shape == pred.shape and len(pred) > 0
    assert np.allclose(got, pred, rtol=0, atol=8e-6)                 # fp32 cos/sin, as in test_postprocess.py
    cd = P.chamfer(pred, gt)
    print("chamfer device / oracle:", res["cd"], cd)
    assert abs(res["cd"] - cd) < 1e-6 * cd
