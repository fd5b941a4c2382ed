"""GPU parity of unet_amd.OptimizedUNet against the CPU oracle and the golden fixtures.

Bounds are relative L2 errors over a whole tensor.

float32 path: outputs 1e-4 (the stated bound; measured ~1e-5).  Parameter gradients pass through
18 ReLUs, 4 max-pools and 18 batch-statistic BatchNorms, where a sign or arg-max flip of one
element changes a gradient discretely: the torch-CPU fp32 oracle itself differs from its own
float64 run by 1.3e-3 (c1) to 7.5e-3 (f48) on these cases, so gradients are compared with the
float64 oracle at 1e-2 on the wide cases and at 1e-4 on the narrow ones, where no flip occurs.

bfloat16 path: eval outputs 2e-2 against the fp32 oracle (the stated bound).  In training mode
bf16 STORAGE alone (bf16 activations and weights, fp32 accumulation and statistics) puts any
pipeline 3e-2..7e-2 away from fp32 in the outputs and 0.35..0.6 in the gradients on these
random weights and pure-noise targets: batch-statistic BatchNorm re-amplifies every layer's
rounding and the gradient is a heavily cancelling sum.  Roundings are chaotic, so two correct
bf16 pipelines also differ from each other by about that much and no bit-level bf16 model
exists.  The oracle therefore has a mode that performs the same storage roundings on the CPU,
and the GPU path must be NO FURTHER from the fp32 oracle than 1.3x that CPU bf16-storage model
is (plus 2e-3): the kernels add nothing beyond the storage format.  Kernel arithmetic itself is
pinned per op in test_gpu_ops.py (bf16 1e-2 on exactly representable inputs).
"""
import numpy as np
import pytest
import torch

from oracle import unet_oracle
from oracle import weights as W

pytestmark = pytest.mark.gpu

CASES = {"c1": (1, 48, 4, 128, 128), "small": (4, 8, 2, 32, 32), "ragged": (4, 8, 2, 40, 24),
         "odd": (4, 16, 1, 50, 36), "f48": (4, 48, 1, 64, 64)}
OUT_TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
NARROW = ("small", "ragged", "odd")


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def sub(t, n):
    f = torch.as_tensor(t).detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().cpu().numpy()


def build(name, dt):
    from unet_amd import OptimizedUNet
    cin, f, B, H, Wd = CASES[name]
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
    m = OptimizedUNet(cin, 1, f, compute_dtype=dt)
    m.load_state_dict(sd)
    return m.cuda(), sd, x, y


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["small", "ragged", "odd", "f48", "c1"])
def test_eval_forward(golden, name, dt):
    m, sd, x, _ = build(name, dt)
    m.eval()
    out = m(x.cuda())
    ref, _ = unet_oracle.unet_forward(sd, x, False)
    assert out.shape == ref.shape and out.dtype == torch.float32
    e = rel(out, ref)
    print(f"eval {name} {dt}: rel {e:.2e}")
    assert e < OUT_TOL[dt]
    assert rel(sub(out, 16384), golden("unet")[f"{name}.eval_out"]) < OUT_TOL[dt]
    if dt == torch.bfloat16:
        emu, _ = unet_oracle.unet_forward(sd, x, False, None, True)
        assert e < 1.3 * rel(emu, ref) + 2e-3


def _run_train(name, dt):
    m, sd, x, y = build(name, dt)
    m.train()
    m.dropout_masks_override = [None] * 9          # Dropout2d off, as in the golden run
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    loss = ((out - y.cuda()) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    return m, sd, x, y, xd, out, loss


def _grad_errors(m, xd, o_g):
    names = [n for n, _ in m.named_parameters()]
    mine = torch.cat([p.grad.double().cpu().reshape(-1) for _, p in m.named_parameters()])
    want = torch.cat([o_g[n].double().reshape(-1) for n in names])
    worst = max((rel(p.grad, o_g[n]), n) for n, p in m.named_parameters())
    return rel(mine, want), worst, rel(xd.grad, o_g["__input__"])


@pytest.mark.parametrize("name", ["small", "ragged", "odd", "f48", "c1"])
def test_train_forward_backward_fp32(golden, name):
    g = golden("unet")
    dt = torch.float32
    m, sd, x, y, xd, out, loss = _run_train(name, dt)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o_out, o_loss, o_g, o_stats = unet_oracle.unet_fwd_bwd(sd64, x.double(), lambda o: ((o - y.double()) ** 2).mean())
    e_out = rel(out, o_out)
    e_all, worst, e_in = _grad_errors(m, xd, o_g)
    print(f"train {name} fp32: out {e_out:.2e} grads(all) {e_all:.2e} worst {worst[0]:.2e} {worst[1]} dx {e_in:.2e}")
    assert e_out < 1e-4
    assert abs(loss.item() - o_loss.item()) < 1e-4 * abs(o_loss.item())
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-4
    gtol = 1e-4 if name in NARROW else 1e-2
    assert e_all < gtol and e_in < gtol and worst[0] < 3 * gtol, (e_all, e_in, worst)
    norms = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    assert rel(norms, g[f"{name}.grad_norms"]) < gtol
    msd = m.state_dict()
    for k, v in o_stats.items():
        assert rel(msd[k], v) < 1e-5, k
    assert int(msd["encoder1.1.num_batches_tracked"]) == 1
    assert int(msd["decoder1.4.num_batches_tracked"]) == 1


def _cat_grads(g, names):
    return torch.cat([g[n].double().reshape(-1) for n in names])


@pytest.mark.parametrize("name", ["small", "ragged", "odd", "f48", "c1"])
def test_train_forward_backward_bf16(golden, name):
    dt = torch.bfloat16
    m, sd, x, y, xd, out, loss = _run_train(name, dt)
    lf = lambda o: ((o - y) ** 2).mean()
    f_out, f_loss, f_g, f_stats = unet_oracle.unet_fwd_bwd(sd, x, lf)               # fp32 model
    e_out, e_loss, e_g, e_stats = unet_oracle.unet_fwd_bwd(sd, x, lf, None, True)   # bf16-storage model on CPU
    names = [n for n, _ in m.named_parameters()]
    mine_g = torch.cat([p.grad.double().cpu().reshape(-1) for _, p in m.named_parameters()])
    ref_g, emu_g = _cat_grads(f_g, names), _cat_grads(e_g, names)
    out_gpu, out_emu = rel(out, f_out), rel(e_out, f_out)
    g_gpu, g_emu = rel(mine_g, ref_g), rel(emu_g, ref_g)
    dx_gpu, dx_emu = rel(xd.grad, f_g["__input__"]), rel(e_g["__input__"], f_g["__input__"])
    cos = torch.dot(mine_g, ref_g) / (mine_g.norm() * ref_g.norm())
    print(f"train {name} bf16 vs fp32: out gpu {out_gpu:.2e} / cpu-bf16 {out_emu:.2e}; grads gpu {g_gpu:.2e} / "
          f"cpu-bf16 {g_emu:.2e}; dx gpu {dx_gpu:.2e} / cpu-bf16 {dx_emu:.2e}; cos {cos:.3f}")
    assert out_gpu < 1.3 * out_emu + 2e-3 and out_gpu < 1e-1
    assert g_gpu < 1.3 * g_emu + 2e-3
    assert dx_gpu < 1.3 * dx_emu + 2e-3
    assert abs(loss.item() - f_loss.item()) < 5e-2 * abs(f_loss.item())
    msd = m.state_dict()
    for k, v in f_stats.items():
        assert rel(msd[k], v) < 1.3 * rel(e_stats[k], v) + 2e-3, k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_seeded_dropout_masks(golden, dt):
    m, sd, x, _ = build("small", dt)
    m.train()
    torch.manual_seed(1234)
    masks = unet_oracle.draw_dropout_masks(2, 8)
    m.dropout_masks_override = masks
    out = m(x.cuda())
    ref, _ = unet_oracle.unet_forward(sd, x, True, masks)
    if dt == torch.float32:
        assert rel(out, ref) < 1e-4
        assert rel(sub(out, 16384), golden("unet")["small.dropout_seed1234_out"]) < 1e-4
    else:
        emu, _ = unet_oracle.unet_forward(sd, x, True, masks, True)
        assert rel(out, ref) < 1.3 * rel(emu, ref) + 2e-3
    # fresh draws: about 5 % of the channels of every block are dropped, survivors scaled by 1/0.95
    m.dropout_masks_override = None
    ks = m._draw_masks(64, torch.device("cuda"))
    frac = torch.cat([(k == 0).float().reshape(-1) for k in ks]).mean().item()
    assert 0.03 < frac < 0.07
    assert all(torch.all((k == 0) | ((k - 1 / 0.95).abs() < 1e-6)) for k in ks)


def test_cpu_input_raises():
    from unet_amd import OptimizedUNet
    m = OptimizedUNet(4, 1, 8)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 16, 16))


def test_repeatable_bitwise():
    """No atomics anywhere: two runs of the same step give identical bits."""
    m, sd, x, y = build("small", torch.bfloat16)
    m.train()
    m.dropout_masks_override = [None] * 9
    res = []
    for _ in range(2):
        m.load_state_dict(sd)
        m.zero_grad()
        out = m(x.cuda())
        ((out - y.cuda()) ** 2).mean().backward()
        res.append((out.clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
