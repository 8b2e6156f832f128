"""Optimizer step of the training loop: clip_grad_norm_ -> AdamW -> update_ema.
CPU: the oracle restatement against torch's own AdamW / clip_grad_norm_ (g12), flat-layout host logic,
and the bucketed gradient all-reduce over gloo (world 2).  GPU: the fused HIP step (rald_optim_*)
against the same golden.  Tolerance: 2e-6 relative (fp32; torch's CPU lerp/addcmul may contract to FMA,
the device kernel rounds after every op) - three steps from identical starts."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from rald_amd import synth

SHAPES = [(128, 96), (1024,), (3, 3, 3, 8, 16), (7,), (1,)]
SCALES = (40.0, 0.01, 3.0)
LR = 2.5e-4


def _grads(it):
    return [synth.normal(list(s), 1000 + 10 * it + i) * SCALES[it] for i, s in enumerate(SHAPES)]


def _close(a, b, tol=2e-6):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max()) <= tol * max(1e-30, float(b.abs().max()))


def test_oracle_vs_torch_adamw_golden():
    from oracle import optim_oracle as OO
    g = load_golden("g12_optim.npz")
    p = [synth.normal(list(s), 900 + i) for i, s in enumerate(SHAPES)]
    ema = [t.clone() for t in p]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    for it in range(3):
        total, gr = OO.clip_grad_norm_(_grads(it), 10.0)
        assert abs(float(total) - float(g["norms"][it])) <= 1e-6 * float(g["norms"][it])
        for i in range(len(p)):
            p[i], m[i], v[i] = OO.adamw_step(p[i], gr[i], m[i], v[i], it + 1, lr=LR)
            ema[i] = OO.update_ema(ema[i], p[i], 0.999)
    for i in range(len(p)):
        assert _close(p[i], g[f"p{i}"]) and _close(ema[i], g[f"ema{i}"])
        assert _close(m[i], g[f"m{i}"]) and _close(v[i], g[f"v{i}"])


def test_flat_layout_alignment():
    from rald_amd.train_utils import flat_layout
    offs, total = flat_layout(SHAPES)
    assert offs == [0, 12288, 13312, 16768, 16776] and total == 16780
    assert all(o % 4 == 0 for o in offs) and total % 4 == 0
    assert flat_layout([]) == ([], 0)


def _reduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from rald_amd.train_utils import GradReducer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 10007 * 4
    flat = synth.normal([n], 70 + rank)
    red = GradReducer(flat, bucket_bytes=16384 * 4)
    assert red.bounds[0] == (n - 16384, n) and red.bounds[-1][0] == 0
    assert sum(b - a for a, b in red.bounds) == n
    red.start()
    launched = [red.mark_ready(n), red.mark_ready(n - 16384 - 5), red.mark_ready(7000), red.mark_ready(7000)]
    pre = red.finish()
    # interval form (EdmTrainer: the transformer's gradients are final before the encoder's, which sit at the END of the buffer):
    # a bucket leaves as soon as a union of finished ranges covers it, in any order, and exactly once
    flat2 = synth.normal([n], 80 + rank)
    red2 = GradReducer(flat2, bucket_bytes=16384 * 4)
    red2.start()
    launched += [red2.mark_range(0, 100), red2.mark_range(100, 7260), red2.mark_range(n - 5000, n), red2.mark_range(7000, n - 5000), red2.mark_range(0, n)]
    red2.finish()
    q.put((rank, launched, pre, flat.numpy().copy(), flat2.numpy().copy()))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29640 + os.getpid() % 200
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    want = (synth.normal([10007 * 4], 70) + synth.normal([10007 * 4], 71)).numpy()
    want2 = (synth.normal([10007 * 4], 80) + synth.normal([10007 * 4], 81)).numpy()
    for rank, launched, pre, flat, flat2 in res:
        assert launched == [0, 1, 1, 0, 0, 1, 0, 2, 0]        # a bucket goes out only once it is complete, whatever the order of the ranges
        assert pre == 0.5
        assert np.array_equal(flat, want)                     # SUM over ranks, every element exactly once
        assert np.array_equal(flat2, want2)


@pytest.mark.gpu
def test_hip_flat_adamw_vs_torch_adamw_golden():
    from rald_amd.train_utils import FlatAdamW
    g = load_golden("g12_optim.npz")
    params = [torch.nn.Parameter(synth.normal(list(s), 900 + i).cuda()) for i, s in enumerate(SHAPES)]
    opt = FlatAdamW(params, lr=LR, ema=True)
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in params)
    for it in range(3):
        opt.zero_grad()
        for p, gr in zip(params, _grads(it)):
            p.grad.add_(gr.cuda())                            # a backward pass accumulates into the flat views
        norm = opt.clip_grad_norm_(10.0)
        assert abs(float(norm) - float(g["norms"][it])) <= 1e-6 * float(g["norms"][it])
        opt.step(ema_rate=0.999)
    sd = opt.state_dict()
    for i, p in enumerate(params):
        assert _close(p.detach().cpu(), g[f"p{i}"]) and _close(opt.ema_params[i].cpu(), g[f"ema{i}"])
        assert _close(sd["state"][i]["exp_avg"].cpu(), g[f"m{i}"]) and _close(sd["state"][i]["exp_avg_sq"].cpu(), g[f"v{i}"])
        assert float(sd["state"][i]["step"]) == 3.0
    # state_dict round trip + a stand-alone EMA update + pre_scale (1/world) folding
    opt2 = FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1.0, ema=True)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and opt2.param_groups[0]["lr"] == LR
    assert torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    before = opt.flat_ema.clone()
    opt.update_ema(0.9)
    assert _close(opt.flat_ema.cpu(), (before * 0.9 + opt.flat_p * (1 - 0.9)).cpu(), 1e-6)
    opt.zero_grad()
    params[0].grad.fill_(2.0)
    n_half = opt.clip_grad_norm_(0.0, pre_scale=0.5)          # max_norm 0: no clipping, only the 1/world scale
    assert abs(float(n_half) - 0.5 * 2.0 * (128 * 96) ** 0.5) < 1e-3
    opt.step(write_back_grads=True)
    assert torch.all(params[0].grad == 1.0)


@pytest.mark.gpu
def test_hip_flat_adamw_full_model_size_and_errors():
    """183.8 M parameters (the denoiser + radar encoder of the shipped config): one fused pass; property
    checks at full size - zero gradient leaves only the decoupled decay, moments stay zero."""
    from rald_amd.train_utils import FlatAdamW
    n = 183_800_000
    p = torch.nn.Parameter(torch.full((n,), 0.5, device="cuda"))
    opt = FlatAdamW([p], lr=1e-3, ema=True)
    norm = opt.clip_grad_norm_(10.0)
    opt.step(ema_rate=0.999)
    torch.cuda.synchronize()
    assert float(norm) == 0.0
    want = np.float32(0.5) * np.float32(1 - 1e-3 * 1e-2)
    assert float(p.detach().min()) == float(p.detach().max()) == float(want)
    assert float(opt.exp_avg.abs().max()) == 0.0 and float(opt.exp_avg_sq.abs().max()) == 0.0
    assert abs(float(opt.flat_ema[12345]) - (0.5 * 0.999 + float(want) * 0.001)) < 1e-7
    with pytest.raises(RuntimeError):
        FlatAdamW([torch.nn.Parameter(torch.zeros(4))])           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        FlatAdamW([])
    with pytest.raises(RuntimeError):
        FlatAdamW([torch.nn.Parameter(torch.zeros(4, device="cuda"))]).update_ema(0.9)
