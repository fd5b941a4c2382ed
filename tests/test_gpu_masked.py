"""Training-mode forward + BACKWARD with live Dropout2d / DropPath factors: the path a real training step runs.

The factors are the ones a seeded run of the reference draws (oracle.draw_dropout_masks / draw_drop_path make the
same torch calls in the same order), handed to the modules through their override hooks.  Expected numbers:
tests/golden/masked.npz, frozen from the reference in float64 with those factors patched into its Dropout2d /
DropPath modules (training.py:530; swin_unet_model.py:139-158, 371-372), and the float64 oracle run here.

Bounds as in test_gpu_unet.py / test_gpu_swin.py: float32 outputs 1e-4, gradients 1e-4 on the narrow UNet cases,
1e-2 on the 48-feature one, 2e-2 on the Swin-UNet (batch statistics over a handful of elements); bfloat16 by the
reference's own bf16-autocast distance for the same case (bf16_bounds).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import swin_oracle, unet_oracle
from oracle import weights as W
from tests.test_gpu_unet import CASES, NARROW, _cosine, bf16_bounds, rel, sub

pytestmark = pytest.mark.gpu

UNET_SEEDS = {"small": 1234, "ragged": 77, "f48": 5}
SWIN_SEEDS = {"s64": (2, 64, 64, 16), "s256": (1, 256, 256, 17), "s24": (2, 24, 24, 3)}


def _heads(grads, names):
    return np.concatenate([sub(grads[n], 16) for n in names])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["small", "ragged", "f48"])
def test_unet_backward_with_live_dropout(golden, name, dt):
    from unet_amd import OptimizedUNet
    g = golden("masked")
    pre = f"unet.{name}"
    cin, f, B, H, Wd = CASES[name]
    seed = UNET_SEEDS[name]
    assert int(g[f"{pre}.seed"]) == seed
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
    torch.manual_seed(seed)
    masks = unet_oracle.draw_dropout_masks(B, f)
    assert [int((k == 0).sum()) for k in masks] == g[f"{pre}.dropped"].tolist()
    m = OptimizedUNet(cin, 1, f, compute_dtype=dt)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.dropout_masks_override = masks
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    loss = ((out - y.cuda()) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o_out, o_loss, o_g, o_stats = unet_oracle.unet_fwd_bwd(sd64, x.double(), lambda o: ((o - y.double()) ** 2).mean(),
                                                            [k.double() for k in masks])
    names = [str(s) for s in g[f"{pre}.param_names"]]
    mine = dict(m.named_parameters())
    assert names == list(mine)
    mg = {n: p.grad for n, p in mine.items()}
    flat = torch.cat([mg[n].double().cpu().reshape(-1) for n in names])
    want = torch.cat([o_g[n].reshape(-1) for n in names])
    e_out, e_g, e_dx = rel(out, o_out), rel(flat, want), rel(xd.grad, o_g["__input__"])
    worst = max((rel(mg[n], o_g[n]), n) for n in names)
    cos, dxcos = _cosine(flat, want), _cosine(xd.grad, o_g["__input__"])
    print(f"unet {name} {dt} live dropout: out {e_out:.2e} grads {e_g:.2e} (cos {cos:.4f}) worst {worst[0]:.2e} "
          f"{worst[1]} dx {e_dx:.2e}")
    # a dropped channel is exactly zero in the output of its block: its BatchNorm parameters get no gradient
    # through the block's consumers only if every sample dropped it; checked through the oracle comparison below
    if dt == torch.float32:
        gtol = 1e-4 if name in NARROW else 1e-2
        assert e_out < 1e-4 and abs(loss.item() - float(g[f"{pre}.loss"])) < 1e-4 * float(g[f"{pre}.loss"])
        assert rel(sub(out, 16384), g[f"{pre}.train_out"]) < 1e-4
        assert e_g < gtol and e_dx < gtol and worst[0] < 3 * gtol, (e_g, e_dx, worst)
        norms = np.array([mg[n].double().norm().item() for n in names])
        assert rel(norms, g[f"{pre}.grad_norms"]) < gtol
        assert rel(_heads(mg, names), g[f"{pre}.grad_heads"]) < 3 * gtol
        assert rel(sub(xd.grad, 4096), g[f"{pre}.grad_input"]) < gtol
        msd = m.state_dict()
        stats = np.concatenate([sub(msd[str(k)], 8) for k in g[f"{pre}.stat_names"]])
        assert rel(stats, g[f"{pre}.stats"]) < 1e-5
    else:
        b_out, b_g, b_cos, b_dx, b_dxcos = bf16_bounds(golden("bf16"), pre)
        assert e_out < b_out and e_g < b_g and cos > b_cos, (e_out, b_out, e_g, b_g, cos, b_cos)
        assert e_dx < b_dx and dxcos > b_dxcos, (e_dx, b_dx, dxcos, b_dxcos)


def _swin_run(dt, name, defer=None, monkeypatch=None, fused_mlp_bwd=True):
    from unet_amd import swin as swin_mod
    from tests.test_gpu_swin import build
    B, H, Wd, seed = SWIN_SEEDS[name]
    if defer is not None:
        monkeypatch.setattr(swin_mod, "DEFER_LN_SUMS", defer)
    monkeypatch.setattr(swin_mod, "FUSED_MLP_BWD", fused_mlp_bwd)
    m, sd, cfg = build(dt)
    m.train()
    torch.manual_seed(seed)
    dp = swin_oracle.draw_drop_path(cfg, B)
    m.drop_path_override = dp
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    tgt = F.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False)
    loss = ((out - tgt.float().cuda()) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    return m, sd, cfg, dp, x, tgt, xd, out, loss


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["s64", "s256", "s24"])
def test_swin_backward_with_live_droppath(golden, monkeypatch, name, dt):
    """Default backward (LayerNorm parameter-gradient sums deferred to one launch, the DropPath scaling of a gradient
    written as a second output of the LayerNorm backward that produces it) and the plain one: same bits; both against
    the float64 oracle and the frozen reference numbers."""
    g = golden("masked")
    pre = f"swin.{name}"
    B, H, Wd, seed = SWIN_SEEDS[name]
    assert int(g[f"{pre}.seed"]) == seed
    if dt == torch.bfloat16 and f"{pre}.out_rel" not in golden("bf16").files:
        pytest.skip("no bf16-autocast run of the reference frozen for this case")
    m, sd, cfg, dp, x, tgt, xd, out, loss = _swin_run(dt, name, True, monkeypatch)
    assert [0 if k is None else int((k == 0).sum()) for pair in dp for k in pair] == g[f"{pre}.dropped"].tolist()
    # deferred sums + scaled second output against the plain backward: same bits (the one-launch data-gradient
    # chain of the MLP branch, which exists only with deferred sums, is switched off for this comparison: it sums a
    # token's channels in another order) ...
    ma, _, _, _, _, _, xda, outa, _ = _swin_run(dt, name, True, monkeypatch, fused_mlp_bwd=False)
    m2, _, _, _, _, _, xd2, out2, _ = _swin_run(dt, name, False, monkeypatch, fused_mlp_bwd=False)
    assert torch.equal(outa, out2) and torch.equal(xda.grad, xd2.grad)
    for (n, p), (_, p2) in zip(ma.named_parameters(), m2.named_parameters()):
        assert torch.equal(p.grad, p2.grad), n
    # ... and the default backward agrees with it to rounding
    assert torch.equal(out, out2)
    ga = torch.cat([p.grad.double().reshape(-1) for p in m.parameters()])
    gb = torch.cat([p.grad.double().reshape(-1) for p in m2.parameters()])
    assert rel(ga, gb) < (1e-6 if dt == torch.float32 else 2e-2), rel(ga, gb)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    xo = x.double().requires_grad_(True)
    dp64 = [[None if k is None else k.double() for k in pair] for pair in dp]
    o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, dp64)
    ((o_out - tgt) ** 2).mean().backward()
    all_names = [str(s) for s in g[f"{pre}.param_names"]]
    mine = dict(m.named_parameters())
    assert all_names == list(mine)
    names = [n for n in all_names if n != "final_conv.0.bias"]
    flat = torch.cat([mine[n].grad.double().cpu().reshape(-1) for n in names])
    want = torch.cat([params[n].grad.reshape(-1) for n in names])
    e_out, e_g, e_dx = rel(out, o_out), rel(flat, want), rel(xd.grad, xo.grad)
    cos, dxcos = _cosine(flat, want), _cosine(xd.grad, xo.grad)
    worst = max((rel(mine[n].grad, params[n].grad), n) for n in names if float(params[n].grad.norm()) > 0)
    print(f"swin {name} {dt} live DropPath: out {e_out:.2e} grads {e_g:.2e} (cos {cos:.4f}) worst {worst[0]:.2e} "
          f"{worst[1]} dx {e_dx:.2e}")
    # a branch dropped for every sample of the batch leaves its parameters without gradient: exactly zero here too
    for n in names:
        if float(params[n].grad.abs().max()) == 0.0:
            assert float(mine[n].grad.abs().max()) == 0.0, n
    if dt == torch.float32:
        assert e_out < 1e-4 and rel(sub(out, 16384), g[f"{pre}.train_out"]) < 1e-4
        assert abs(loss.item() - float(g[f"{pre}.loss"])) < 1e-4 * float(g[f"{pre}.loss"])
        assert e_g < 2e-2 and e_dx < 2e-2 and worst[0] < 1e-1, (e_g, e_dx, worst)
        keep = np.array([n != "final_conv.0.bias" for n in all_names])
        norms = np.array([mine[n].grad.double().norm().item() for n in all_names])
        assert rel(norms[keep], g[f"{pre}.grad_norms"][keep]) < 2e-2
        assert rel(sub(xd.grad, 4096), g[f"{pre}.grad_input"]) < 2e-2
        msd = m.state_dict()
        stats = np.concatenate([sub(msd[str(k)], 8) for k in g[f"{pre}.stat_names"]])
        assert rel(stats, g[f"{pre}.stats"]) < 1e-4
    else:
        b_out, b_g, b_cos, b_dx, b_dxcos = bf16_bounds(golden("bf16"), pre)
        assert e_out < b_out and e_g < b_g and cos > b_cos, (e_out, b_out, e_g, b_g, cos, b_cos)
        assert e_dx < b_dx and dxcos > b_dxcos, (e_dx, b_dx, dxcos, b_dxcos)
