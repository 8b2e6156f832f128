"""Decode post-processing: positives compaction + inverse normalisation +
polar->cartesian, Chamfer distance, accuracy/IoU.  CPU: the numpy oracle against outputs of the
reference's own helper functions (g9).  GPU: the HIP kernels (through rald_post_*) against the same
golden and against the oracle on edge cases.  Indices are bit-exact; coordinates agree to fp32
rounding of cos/sin (<= 8e-6 m at r <= 15.8 m); Chamfer to 1e-6 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rald_amd import synth

PC_RANGE = [0, -90, -20, 15.8, 90, 20]
Q = 60000


def _inputs():
    q = synth.queries(1, Q, seed=31)[0].numpy()
    rr = np.linalg.norm(q, axis=1)
    logits = (0.06 - np.abs(rr - 0.9)).astype(np.float32) * 10 + 0.05 * synth.normal([Q], 32).numpy()
    surface = synth.point_cloud(1, 10000, seed=33)[0].numpy()
    return logits, q, surface


def test_oracle_vs_reference_golden():
    from oracle import post_oracle as P
    g = load_golden("g9_postprocess.npz")
    logits, q, surface = _inputs()
    pred, ind = P.occupied_points(logits, q, PC_RANGE, True, False, True)
    assert len(ind) == int(g["n_pos"])
    assert np.array_equal(ind[:64], g["ind_head"]) and np.array_equal(ind[-64:], g["ind_tail"])
    assert np.array_equal(pred[:256], g["pred_head"].numpy()) and np.array_equal(pred[-256:], g["pred_tail"].numpy())
    gt = P.polar2cartesian(P.inverse_norm_points(surface, PC_RANGE, True, False))
    assert np.array_equal(gt[:256], g["gt_head"].numpy())
    assert abs(P.chamfer(pred, gt) - float(g["cd"])) < 1e-9 * float(g["cd"])
    assert np.array_equal(P.inverse_norm_points(q[:1000], PC_RANGE, False, True), g["iso_head"].numpy())
    labels = (synth.normal([2, 4096], 34).numpy() > 0.3).astype(np.float32)
    acc, iou = P.accuracy_iou(synth.normal([2, 4096], 35).numpy(), labels)
    assert np.allclose(acc, g["acc"].numpy(), rtol=0, atol=1e-7) and np.allclose(iou, g["iou"].numpy(), rtol=1e-6)


@pytest.mark.gpu
def test_hip_postprocess_vs_reference_golden():
    from rald_amd import postprocess as PP
    g = load_golden("g9_postprocess.npz")
    logits, q, surface = _inputs()
    pts, idx = PP.occupied_points(torch.from_numpy(logits).cuda(), torch.from_numpy(q).cuda(), PC_RANGE, True, False, True,
                                  return_index=True)
    assert pts.shape[0] == int(g["n_pos"])
    idx = idx.cpu().numpy()
    assert np.array_equal(idx[:64], g["ind_head"]) and np.array_equal(idx[-64:], g["ind_tail"])      # order preserved, bit-exact
    assert np.all(np.diff(idx) > 0)
    p = pts.cpu().numpy()
    print("max |coord diff| head/tail (m):", np.abs(p[:256] - g["pred_head"].numpy()).max(), np.abs(p[-256:] - g["pred_tail"].numpy()).max())
    # fp32 cos/sin of the device libm vs numpy's differ by <= ~2 ulp; at r <= 15.8 m that is <= 8e-6 m
    assert np.allclose(p[:256], g["pred_head"].numpy(), rtol=0, atol=8e-6)
    assert np.allclose(p[-256:], g["pred_tail"].numpy(), rtol=0, atol=8e-6)
    assert np.allclose(p.astype(np.float64).sum(0), g["pred_sum"].numpy(), rtol=1e-6)
    gt = PP.polar2cartesian(PP.inverse_norm_points(torch.from_numpy(surface).cuda(), PC_RANGE, True, False))
    assert np.allclose(gt[:256].cpu().numpy(), g["gt_head"].numpy(), rtol=0, atol=8e-6)
    cd = PP.cal_metrics(pts, gt)
    print("chamfer", cd, "ref", float(g["cd"]))
    assert abs(cd - float(g["cd"])) < 1e-6 * float(g["cd"])
    iso = PP.inverse_norm_points(torch.from_numpy(q[:1000]).cuda(), PC_RANGE, False, True)
    assert np.array_equal(iso.cpu().numpy(), g["iso_head"].numpy())                                   # pure mul/add: bit-exact
    labels = (synth.normal([2, 4096], 34) > 0.3).float()
    acc, iou = PP.accuracy_iou(synth.normal([2, 4096], 35).cuda(), labels.cuda())
    assert np.allclose(acc.cpu().numpy(), g["acc"].numpy(), atol=1e-7) and np.allclose(iou.cpu().numpy(), g["iou"].numpy(), rtol=1e-6)


@pytest.mark.gpu
def test_hip_postprocess_edge_cases():
    """No positives (cal_metrics -> inf), all positives, ragged Q (not a multiple of the 1024-query
    compaction block), a single query; full-size Q = 1.2 M checked through its invariants."""
    from oracle import post_oracle as P
    from rald_amd import postprocess as PP
    q = synth.queries(1, 5000, seed=5)[0]
    none = PP.occupied_points(torch.full((5000,), -1.0).cuda(), q.cuda(), PC_RANGE, True, False, True)
    assert none.shape == (0, 3) and PP.cal_metrics(none, q.cuda()) == float("inf")
    allp, idx = PP.occupied_points(torch.ones(5000).cuda(), q.cuda(), PC_RANGE, True, False, False, return_index=True)
    assert allp.shape == (5000, 3) and torch.equal(idx.cpu(), torch.arange(5000))
    assert np.array_equal(allp.cpu().numpy(), P.inverse_norm_points(q.numpy(), PC_RANGE, True, False))
    one = PP.occupied_points(torch.tensor([0.5]).cuda(), q[:1].cuda(), PC_RANGE, True, False, False)
    assert one.shape == (1, 3)
    for n in (1023, 1025, 4097):
        lg = synth.normal([n], n)
        pts, idx = PP.occupied_points(lg.cuda(), synth.queries(1, n, seed=n)[0].cuda(), PC_RANGE, True, False, True, return_index=True)
        assert np.array_equal(idx.cpu().numpy(), np.where(lg.numpy() > 0)[0])
    # full size: 1.2 M queries, count / order / chamfer symmetry invariants
    Qf = 1200000
    lg = synth.normal([Qf], 77) - 1.5
    qf = synth.queries(1, Qf, seed=78)[0]
    pts, idx = PP.occupied_points(lg.cuda(), qf.cuda(), PC_RANGE, True, False, True, return_index=True)
    ref_idx = np.where(lg.numpy() > 0)[0]
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    gt = PP.polar2cartesian(PP.inverse_norm_points(synth.point_cloud(1, 10000, seed=79)[0].cuda(), PC_RANGE, True, False))
    assert PP.cal_metrics(gt, gt) == 0.0
    cd_ab, cd_ba = PP.cal_metrics(pts, gt), PP.cal_metrics(gt, pts)
    assert abs(cd_ab - cd_ba) < 1e-12 * cd_ab                       # Chamfer is symmetric in its arguments
    sub = np.random.default_rng(0).choice(len(ref_idx), 3000, replace=False)
    cd_sub = P.chamfer(pts.cpu().numpy()[np.sort(sub)], gt.cpu().numpy())
    assert abs(PP.cal_metrics(pts[torch.from_numpy(np.sort(sub)).cuda()], gt) - cd_sub) < 1e-9 * cd_sub
