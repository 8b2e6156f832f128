"""CPU ORACLE (test infrastructure only) for the decode post-processing row: a numpy restatement of
utils/utils.py:50-75 (inverse_norm_points), :116-142 (cal_metrics / chamfer_distance),
dataset_preprocessor/lidar.py:57-63 (polar2cartesian) and engine_generation.py:229-243, :283-292.
Pinned against outputs of the reference's own functions (tests/golden/g9_postprocess.npz)."""
import numpy as np


def inverse_norm_points(points, pc_range, norm_anisotropy, norm_isotropy):
    off = [(pc_range[3 + i] + pc_range[i]) / 2 for i in range(3)]
    sc = [(pc_range[3 + i] - pc_range[i]) / 2 for i in range(3)]
    out = np.zeros_like(points)
    if norm_anisotropy:
        for i in range(3):
            out[:, i] = points[:, i] * sc[i] + off[i]
    if norm_isotropy:
        out[:, :3] = points[:, :3] * max(sc) + np.array(off)
    return out


def polar2cartesian(points):
    r, az, el = points[:, 0], -np.deg2rad(points[:, 1]), np.deg2rad(points[:, 2])
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], axis=1)


def occupied_points(logits, queries, pc_range, aniso, iso, view_cone=True):
    """np.where(output > 0) -> grid[ind] -> inverse_norm_points -> polar2cartesian (:287-318)."""
    ind = np.where(logits > 0)[0]
    pts = inverse_norm_points(queries[ind], pc_range, aniso, iso)
    return (polar2cartesian(pts) if view_cone else pts), ind


def chamfer(y_pred, y_gt, chunk=2048):
    """Exact nearest-neighbour Chamfer (what cal_metrics gets from its two cKDTrees), brute force in float64."""
    if len(y_pred) == 0:
        return np.inf
    a, b = y_pred.astype(np.float64), y_gt.astype(np.float64)

    def mean_min(p, q):
        tot = 0.0
        for i in range(0, len(p), chunk):
            d = ((p[i:i + chunk, None, :] - q[None, :, :]) ** 2).sum(-1)
            tot += np.sqrt(d.min(1)).sum()
        return tot / len(p)

    return 0.5 * mean_min(b, a) + 0.5 * mean_min(a, b)


def accuracy_iou(outputs, labels):
    pred = (outputs >= 0).astype(np.float32)
    acc = (pred == labels).astype(np.float32).sum(1) / labels.shape[1]
    inter = (pred * labels).sum(1)
    union = ((pred + labels) > 0).sum(1)
    return acc, inter * 1.0 / union + 1e-5


def process_radar_data(raw, norm_intensity=True, max_intensity=45, norm_dopp=True, max_dopp=2.4958, tgt=(64, 32)):
    """ColoRadarDataset.process_radar_data (Coloradar_dataset.py:432-475): raw [R,A,E,C] -> [R,tA,tE,2].
    The up-sampling is the same public torch call the reference makes (F.interpolate bilinear, align_corners=True)."""
    import torch
    import torch.nn.functional as F
    raw = np.array(raw, dtype=np.float32, copy=True)
    out = np.zeros(raw.shape[:3] + (2,), dtype=np.float32)
    if norm_intensity:
        out[..., 0] = np.clip(raw[..., 0], 0, max_intensity) / max_intensity
    out[..., 1] = raw[..., 1] * raw[..., -1]
    if norm_dopp:
        out[..., 1] = out[..., 1] / max_dopp
    chans = [F.interpolate(torch.from_numpy(out[..., c]).unsqueeze(0), size=tgt, mode="bilinear", align_corners=True).squeeze(0).numpy()
             for c in range(2)]
    return np.stack(chans, axis=-1)
