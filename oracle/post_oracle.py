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


# ---- query generation + refine --------------------------------------------
def _offset_scale(pc_range):
    off = [(pc_range[3 + a] + pc_range[a]) / 2 for a in range(3)]
    scale = [(pc_range[3 + a] - pc_range[a]) / 2 for a in range(3)]
    return off, scale


def norm_points(points, pc_range, norm_anisotropy, norm_isotropy):
    """utils/utils.py:77-104 (dtype promotion as numpy does it: python-float scalars keep the array's
    dtype, the isotropic branch goes through a float64 offset array)."""
    off, scale = _offset_scale(pc_range)
    out = np.zeros_like(points)
    if norm_anisotropy:
        for a in range(3):
            out[:, a] = (points[:, a] - off[a]) / scale[a]
    if norm_isotropy:
        out[:, :3] = (points[:, :3] - np.array(off)) / max(scale)
    return out


def cartesian2polar(points):
    """dataset_preprocessor/lidar.py:49-55."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    return np.stack([r, -np.rad2deg(np.arctan2(y, x)), np.rad2deg(np.arcsin(z / r))], axis=1)


def remove_points_outside_fov(points):
    """utils/utils.py:106-112."""
    return points[np.all((points > -1) & (points < 1), axis=1)]


def query_box(pc_range, norm_anisotropy, norm_isotropy):
    """x/y/z_min, x/y/z_max of generate_query_points (utils/utils.py:157-169)."""
    if not (norm_anisotropy or norm_isotropy):
        raise ValueError("one of norm_anisotropy / norm_isotropy is required")       # NameError in the reference
    _, scale = _offset_scale(pc_range)
    lo, hi = [-1.0] * 3, [1.0] * 3
    if norm_isotropy:
        lo = [-(s / max(scale)) for s in scale]
        hi = [s / max(scale) for s in scale]
    return lo, hi


def queries_from_uniform(u3n, pc_range, norm_anisotropy, norm_isotropy):
    """generate_query_points (utils/utils.py:147-175) with the uniforms made explicit: u3n [3,n] float64
    in draw order; np.random.uniform(lo, hi, n) == lo + (hi - lo) * random_sample(n).  float64 out."""
    lo, hi = query_box(pc_range, norm_anisotropy, norm_isotropy)
    return np.stack([lo[a] + (hi[a] - lo[a]) * u3n[a] for a in range(3)], axis=1)


def cart_queries_from_uniform(u3n, pc_range_cart, pc_range, norm_anisotropy, norm_isotropy):
    """engine_generation.py:251-256 (use_cart_query), float32 out."""
    g = queries_from_uniform(u3n, pc_range_cart, norm_anisotropy, norm_isotropy)
    g = inverse_norm_points(g, pc_range_cart, norm_anisotropy, norm_isotropy)
    g = norm_points(cartesian2polar(g), pc_range, norm_anisotropy, norm_isotropy)
    return remove_points_outside_fov(g).astype(np.float32)


def aug_query_helper_from_draws(helper_points, aug_num, pc_range, voxel_size, sel, scales, u):
    """datasets/utils/query_helper.py:3-42 with its three random draws made explicit
    (sel = np.random.choice(N, gen), scales = np.random.choice(1..aug_bias_scale, gen), u = np.random.rand(gen, 3))."""
    N = helper_points.shape[0]
    out = np.zeros((aug_num, 3), np.float32)
    if N >= aug_num:
        out[:] = helper_points[:aug_num]
        return out
    biases = (u * 2 - 1) * (np.asarray(voxel_size, np.float64) * scales[:, None])
    aug = np.clip(helper_points[sel] + biases, pc_range[:3], pc_range[3:])
    out[:N] = helper_points
    out[N:] = aug
    return out
