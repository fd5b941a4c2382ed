"""GPU tests: fused clip_grad_norm_ + AdamW against torch's own and against the frozen reference step."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle
from oracle import weights as W

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def sub(t, n):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().cpu().numpy()


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]


SHAPES = [(48, 4, 3, 3), (48,), (1,), (768, 768, 3, 3), (33000,), (7, 5)]


@pytest.mark.parametrize("max_norm", [None, 1.0, 1e6])
def test_fused_adamw_matches_torch(max_norm):
    from unet_amd.optim import FusedAdamW
    ours, ref = _params(1, SHAPES), _params(1, SHAPES)
    o1 = FusedAdamW(ours, lr=3e-3, weight_decay=0.05, max_grad_norm=max_norm)
    o2 = torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.05)
    for it in range(3):
        g = torch.Generator().manual_seed(100 + it)
        for a, b in zip(ours, ref):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 if it == 1 else 0.01)
            a.grad, b.grad = gr.clone(), gr.clone()
        if max_norm is not None:
            want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        o1.step()
        o2.step()
        if max_norm is not None:
            assert abs(o1.last_grad_norm.item() - want_norm.item()) < 1e-5 * want_norm.item()
            for a, b in zip(ours, ref):      # the clipped gradient is written back, as torch does
                assert rel(a.grad.cpu(), b.grad.cpu()) < 1e-6
        for a, b in zip(ours, ref):
            assert rel(a.detach().cpu(), b.detach().cpu()) < 1e-6
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert set(sd1["state"][0]) == set(sd2["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert rel(sd1["state"][3]["exp_avg_sq"].cpu(), sd2["state"][3]["exp_avg_sq"].cpu()) < 1e-6


def test_clip_grad_norm_matches_torch():
    from unet_amd.optim import clip_grad_norm_
    ours, ref = _params(2, SHAPES), _params(2, SHAPES)
    g = torch.Generator().manual_seed(5)
    for a, b in zip(ours, ref):
        gr = torch.randn(a.shape, generator=g).cuda()
        a.grad, b.grad = gr.clone(), gr.clone()
    n1 = clip_grad_norm_(ours, 1.0)
    n2 = torch.nn.utils.clip_grad_norm_(ref, 1.0)
    assert abs(n1.item() - n2.item()) < 1e-5 * n2.item()
    for a, b in zip(ours, ref):
        assert rel(a.grad.cpu(), b.grad.cpu()) < 1e-6
    with pytest.raises(RuntimeError):
        cpu = torch.nn.Parameter(torch.zeros(3))
        cpu.grad = torch.ones(3)
        clip_grad_norm_([cpu], 1.0)


def test_training_step_matches_frozen_reference(golden):
    """Model, loss and optimizer tail together reproduce the reference trainer's first step."""
    from unet_amd import OptimizedUNet
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    g = golden("step")
    cin, f, B, H, Wd = 4, 8, 2, 32, 32
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
    model = OptimizedUNet(cin, 1, f)
    model.load_state_dict(sd)
    model = model.cuda().train()
    model.dropout_masks_override = [torch.ones(B, c, device="cuda") for c in
                                    (f, 2 * f, 4 * f, 8 * f, 16 * f, 8 * f, 4 * f, 2 * f, f)]
    crit = MultiLossFunction(loss_oracle.preset("all")).cuda()
    opt = FusedAdamW(model.parameters(), lr=5e-5, weight_decay=5e-5, max_grad_norm=1.0)
    loss, parts = crit(model(x.cuda()), y.cuda(), w.cuda())
    opt.zero_grad()
    loss.backward()
    opt.step()
    got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
    assert rel(got, g["parts"]) < 1e-4
    assert abs(opt.last_grad_norm.item() - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    names = [str(s) for s in g["param_names"]]
    now = dict(model.named_parameters())
    delta = np.array([(now[k].detach().cpu() - sd[k]).double().norm().item() for k in names])
    assert rel(delta, g["param_delta_norms"]) < 2e-2       # first Adam step ~ lr*sign(g): flips where g ~ 0
    heads = np.concatenate([sub(now[k], 8) for k in names])
    assert rel(heads, g["param_heads"]) < 1e-5
