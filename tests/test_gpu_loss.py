"""GPU parity of the multi-loss kernels against the CPU oracle and the golden fixtures.

All loss arithmetic is fp32; bound 1e-4 relative on every loss part and 1e-4 relative L2 on the
gradient with respect to the prediction (the stated fp32 bound; measured ~1e-6).
"""
import numpy as np
import pytest
import torch

from oracle import loss_oracle
from oracle import weights as W
from tests.test_oracle_golden import LOSS_CFGS, LOSS_SHAPES, loss_inputs, sub

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("si", [0, 1, 2])
def test_multiloss_matches_oracle_and_golden(golden, si):
    from unet_amd.losses import MultiLossFunction
    g = golden("loss")
    p, y, w = loss_inputs(si)
    for wname, wt in (("none", None), ("rand", w), ("zero", torch.zeros_like(w))):
        for cname, kw in LOSS_CFGS.items():
            cfg = loss_oracle.preset(**kw)
            crit = MultiLossFunction(cfg).cuda()
            pd = p.cuda().requires_grad_(True)
            total, parts = crit(pd, y.cuda(), None if wt is None else wt.cuda())
            total.backward()
            po = p.clone().requires_grad_(True)
            o_total, o_parts = loss_oracle.multi_loss(cfg, po, y, wt)
            o_total.backward()
            key = f"s{si}.{wname}.{cname}"
            for k in ("pixel", "gradient", "spectral", "total"):
                assert abs(parts[k] - o_parts[k]) <= 1e-4 * max(abs(o_parts[k]), 1e-6), (key, k, parts, o_parts)
            got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
            assert np.allclose(got, g[key + ".parts"], rtol=1e-4, atol=1e-9), key
            assert abs(total.item() - o_total.item()) <= 1e-4 * abs(o_total.item())
            e = rel(pd.grad, po.grad)
            assert e < 1e-4, (key, e)
            assert rel(sub(pd.grad.cpu(), 2048), g[key + ".grad"]) < 1e-4, key


@pytest.mark.parametrize("shape", [(2, 19, 30), (1, 9, 7), (2, 20, 36)])
def test_multiloss_widths_with_and_without_the_four_pixel_kernels(shape):
    """Image widths that are not a multiple of four take the one-pixel-per-thread kernels, the others (every golden
    shape among them) the four-pixel ones: both against the oracle, every preset, with and without weights."""
    from unet_amd.losses import MultiLossFunction
    B, H, Wd = shape
    _, y, w = W.synthetic_batch(B, 1, H, Wd, seed=31)
    p = W.synthetic_batch(B, 1, H, Wd, seed=32)[1] * 0.7 + 0.3 * y
    for wt in (None, w):
        for cname, kw in LOSS_CFGS.items():
            cfg = loss_oracle.preset(**kw)
            crit = MultiLossFunction(cfg).cuda()
            pd = p.cuda().requires_grad_(True)
            total, parts = crit(pd, y.cuda(), None if wt is None else wt.cuda())
            total.backward()
            po = p.clone().requires_grad_(True)
            o_total, o_parts = loss_oracle.multi_loss(cfg, po, y, wt)
            o_total.backward()
            for k in ("pixel", "gradient", "spectral", "total"):
                assert abs(parts[k] - o_parts[k]) <= 1e-4 * max(abs(o_parts[k]), 1e-6), (shape, cname, k)
            assert rel(pd.grad, po.grad) < 1e-4, (shape, cname)


def test_sub_losses_and_errors():
    from unet_amd.losses import GradientLoss, MultiLossFunction, SpectralLoss, TIDWeightedLoss
    p, y, w = loss_inputs(0)
    pd, yd, wd = p.cuda(), y.cuda(), w.cuda()
    assert abs(TIDWeightedLoss("mse", True, 1.2)(pd, yd, wd).item()
               - loss_oracle.pixel_loss(p, y, w, "mse", True, 1.2).item()) < 1e-5
    assert abs(TIDWeightedLoss("mae", False)(pd, yd, wd).item()
               - loss_oracle.pixel_loss(p, y, w, "mae", False).item()) < 1e-5
    for m in ("sobel", "diff"):
        assert abs(GradientLoss(m)(pd, yd, wd).item() - loss_oracle.gradient_loss(p, y, w, m).item()) < 1e-4
        assert abs(GradientLoss(m)(pd, yd).item() - loss_oracle.gradient_loss(p, y, None, m).item()) < 1e-4
    ref = loss_oracle.spectral_loss(p, y, w, True, 2.5).item()
    assert abs(SpectralLoss(True, 2.5)(pd, yd, wd).item() - ref) < 1e-4 * abs(ref)
    with pytest.raises(ValueError):
        TIDWeightedLoss("huber")
    with pytest.raises(NameError):
        SpectralLoss(False)(pd, yd, wd)
    with pytest.raises(RuntimeError):
        MultiLossFunction(loss_oracle.preset("all"))(p, y, w)          # CPU tensors: no fallback


def test_full_size_linearity_property():
    """At the bench size (64 x 256 x 256) the oracle is too slow for every config; check the
    size-independent property that the loss of (pred, target) equals the loss of (pred - c, target - c)
    for pixel and gradient parts, and that scaling the upstream gradient scales dpred."""
    from unet_amd.losses import MultiLossFunction
    x, y, w = W.synthetic_batch(64, 1, 256, 256, seed=3)
    p = (0.6 * y + 0.4 * x).cuda()
    yd, wd = y.cuda(), w.cuda()
    crit = MultiLossFunction(loss_oracle.preset("weighted_mse_grad")).cuda()
    a, pa = crit(p, yd, wd)
    b, pb = crit(p - 0.75, yd - 0.75, wd)
    assert abs(pa["pixel"] - pb["pixel"]) < 1e-5 * pa["pixel"]
    # the Sobel stencil is shift-invariant only in the interior (zero padding): compare loosely
    assert abs(pa["gradient"] - pb["gradient"]) < 0.05 * pa["gradient"]
    crit = MultiLossFunction(loss_oracle.preset("all")).cuda()
    p1 = p.clone().requires_grad_(True)
    t1, _ = crit(p1, yd, wd)
    t1.backward()
    p2 = p.clone().requires_grad_(True)
    t2, _ = crit(p2, yd, wd)
    (3.0 * t2).backward()
    assert rel(p2.grad, 3.0 * p1.grad) < 1e-6
    # one sample against the oracle at full resolution
    cfg = loss_oracle.preset("all")
    o, parts = loss_oracle.multi_loss(cfg, p[:2].cpu(), y[:2], w[:2])
    t, gp = crit(p[:2], yd[:2], wd[:2])
    assert abs(t.item() - o.item()) < 1e-4 * abs(o.item())


def test_loss_parts_behave_like_the_reference_dict():
    """The parts dict fetches its floats lazily but reads like a plain dict of floats."""
    import json
    import pickle
    from unet_amd.losses import MultiLossFunction
    x, y, w = W.synthetic_batch(2, 1, 16, 16, seed=3)
    crit = MultiLossFunction(loss_oracle.preset("all")).cuda()
    total, parts = crit(x.cuda(), y.cuda(), w.cuda())
    assert isinstance(parts, dict) and set(parts) == {"pixel", "gradient", "spectral", "total"}
    plain = dict(parts)
    assert all(isinstance(v, float) for v in plain.values())
    assert abs(plain["total"] - total.item()) < 1e-6 * max(1.0, abs(total.item()))
    _, again = crit(x.cuda(), y.cuda(), w.cuda())
    assert again == plain and {**again} == plain and json.loads(json.dumps(again)) == plain
    assert pickle.loads(pickle.dumps(again)) == plain and again.copy() == plain and again.get("pixel") == plain["pixel"]
    acc = {k: 0.0 for k in plain}
    _, third = crit(x.cuda(), y.cuda(), w.cuda())
    for k, v in third.items():      # the trainer's accumulation loop
        acc[k] += v
    assert acc == plain
