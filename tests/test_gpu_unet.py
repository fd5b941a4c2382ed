"""GPU parity of unet_amd.OptimizedUNet against the CPU oracle and the golden fixtures.

Bounds are relative L2 errors over a whole tensor.

float32 path: outputs 1e-4 (the stated bound; measured ~1e-5).  Parameter gradients pass through
18 ReLUs, 4 max-pools and 18 batch-statistic BatchNorms, where a sign or arg-max flip of one
element changes a gradient discretely: the torch-CPU fp32 oracle itself differs from its own
float64 run by 1.3e-3 (c1) to 7.5e-3 (f48) on these cases, so gradients are compared with the
float64 oracle at 1e-2 on the wide cases and at 1e-4 on the narrow ones, where no flip occurs.

bfloat16 path: eval outputs 2e-2 against the fp32 oracle (the stated bound).  In training mode the
bound comes from the REFERENCE ITSELF: tests/golden/bf16.npz holds how far the reference modules,
run under torch.autocast('cpu', dtype=torch.bfloat16), land from their own fp32 run on these same
seeded cases (outputs 3.6e-2..7.0e-2, parameter gradients 0.39..0.58 relative L2 with cosine
0.82..0.93: random weights and pure-noise targets make the gradient a heavily cancelling sum that
batch-statistic BatchNorm re-amplifies).  The bf16 GPU path must be no further from fp32 than 1.3x
those reference-run numbers, under absolute caps, with a cosine floor.  The oracle's CPU model of
the storage roundings is printed next to it as a diagnostic only.  Kernel arithmetic itself is pinned
per op in test_gpu_ops.py (bf16 1e-2 on exactly representable inputs), and
test_bf16_loss_curve_tracks_fp32 checks the regime that matters for training: a learnable target.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


def bf16_bounds(gb, prefix):
    """(output, gradient, gradient-cosine, input-gradient, input-gradient-cosine) bounds for a bf16 run
    against fp32, from the reference's own bf16-autocast run of the same case: 1.3x its distance,
    capped; cosine no further below 1 than 1.3x."""
    f = lambda k: float(gb[f"{prefix}.{k}"])
    return (min(1.3 * f("out_rel") + 2e-3, 1e-1), min(1.3 * f("grad_rel"), 0.8),
            max(1.0 - 1.3 * (1.0 - f("grad_cos")), 0.75), min(1.3 * f("dx_rel"), 0.85),
            max(1.0 - 1.3 * (1.0 - f("dx_cos")), 0.70))


def _cosine(a, b):
    a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1).cpu()
    return (torch.dot(a, b) / (a.norm() * b.norm())).item()


@pytest.mark.parametrize("name", ["small", "ragged", "odd", "f48", "c1"])
def test_train_forward_backward_bf16(golden, name):
    dt = torch.bfloat16
    gb = golden("bf16")
    m, sd, x, y, xd, out, loss = _run_train(name, dt)
    lf = lambda o: ((o - y) ** 2).mean()
    f_out, f_loss, f_g, f_stats = unet_oracle.unet_fwd_bwd(sd, x, lf)               # fp32 model
    e_out, e_loss, e_g, e_stats = unet_oracle.unet_fwd_bwd(sd, x, lf, None, unet_oracle.BF16_ALL)   # diagnostic
    names = [n for n, _ in m.named_parameters()]
    mine_g = torch.cat([p.grad.double().cpu().reshape(-1) for _, p in m.named_parameters()])
    ref_g, emu_g = _cat_grads(f_g, names), _cat_grads(e_g, names)
    out_gpu, g_gpu, dx_gpu = rel(out, f_out), rel(mine_g, ref_g), rel(xd.grad, f_g["__input__"])
    cos, dx_cos = _cosine(mine_g, ref_g), _cosine(xd.grad, f_g["__input__"])
    b_out, b_g, b_cos, b_dx, b_dxcos = bf16_bounds(gb, f"unet.{name}")
    print(f"train {name} bf16 vs fp32: out {out_gpu:.3e} (reference autocast {float(gb[f'unet.{name}.out_rel']):.3e}, "
          f"cpu storage model {rel(e_out, f_out):.3e}); grads {g_gpu:.3f} cos {cos:.3f} (reference "
          f"{float(gb[f'unet.{name}.grad_rel']):.3f} / {float(gb[f'unet.{name}.grad_cos']):.3f}, model {rel(emu_g, ref_g):.3f}); "
          f"dx {dx_gpu:.3f} cos {dx_cos:.3f} (reference {float(gb[f'unet.{name}.dx_rel']):.3f})")
    assert out_gpu < b_out, (out_gpu, b_out)
    assert g_gpu < b_g and cos > b_cos, (g_gpu, b_g, cos, b_cos)
    assert dx_gpu < b_dx and dx_cos > b_dxcos, (dx_gpu, b_dx, dx_cos, b_dxcos)
    assert abs(loss.item() - f_loss.item()) < 5e-2 * abs(f_loss.item())
    msd = m.state_dict()
    for k, v in f_stats.items():
        assert rel(msd[k], v) < 2e-2, k


def test_bf16_loss_curve_tracks_fp32():
    """The regime that matters: a learnable target (a fixed smooth function of the input), 40 optimiser
    steps from the same initialisation in fp32 and in bf16 storage, while the loss falls by a factor of ~40.
    The two loss curves must agree within 10 % at every step and within 4 % on average, and both must learn
    (final loss below a tenth of the first).  The learning rate is 3e-4: at 1e-3 the fp32 path's own response
    to a 1e-6 relative perturbation of the input reaches 8 % of the loss in the late steps (20 % at 5e-4 over
    30 steps), so a per-step comparison there measures the optimiser's noise, not the storage precision
    (tools/exp_curve.py prints both); the fp32 self-deviation is measured here too and must stay below the
    bf16 bound, so that the comparison means something."""
    from unet_amd import OptimizedUNet
    from unet_amd.optim import FusedAdamW
    torch.manual_seed(7)
    B, H, Wd = 8, 64, 64
    x = F.avg_pool2d(torch.randn(B, 4, H + 4, Wd + 4), 5, 1)            # smooth inputs
    x = x / x.std()
    y = torch.tanh(0.8 * x[:, :1] - 0.6 * x[:, 1:2] * x[:, 2:3]) + 0.3 * F.avg_pool2d(F.pad(x[:, 3:4], (1, 1, 1, 1)), 3, 1)
    init = OptimizedUNet(4, 1, 16).state_dict()
    noise = torch.randn_like(x)

    def run(dt, eps=0.0):
        m = OptimizedUNet(4, 1, 16, compute_dtype=dt)
        m.load_state_dict(init)
        m = m.cuda().train()
        m.dropout_masks_override = [None] * 9
        opt = FusedAdamW(m.parameters(), lr=3e-4, weight_decay=5e-5, max_grad_norm=1.0)
        xd, yd = (x * (1 + eps * noise)).cuda(), y.cuda()
        losses = []
        for _ in range(40):
            opt.zero_grad()
            loss = ((m(xd) - yd) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return np.array(losses)

    f32, b16, f32_eps = run(torch.float32), run(torch.bfloat16), run(torch.float32, 1e-6)
    dev = np.abs(b16 - f32) / f32
    self_dev = np.abs(f32_eps - f32) / f32
    print("fp32 loss curve", np.round(f32, 4).tolist())
    print("bf16 loss curve", np.round(b16, 4).tolist(), "max deviation", dev.max(), "mean", dev.mean(),
          "; fp32 against itself with 1e-6 input noise: max", self_dev.max())
    assert f32[-1] < 0.1 * f32[0] and b16[-1] < 0.1 * b16[0]
    assert self_dev.max() < 0.10, self_dev
    assert dev.max() < 0.10 and dev.mean() < 0.04, dev


def test_config2_full_size_properties(golden):
    """BASELINE config 2 at full size (batch 64 of 4x256x256, UNet(4,1,48)), where the CPU oracle is too
    slow: (1) eval mode is per-sample, so the first samples of the batch-64 run equal a batch-3 run bit
    for bit (exercises the >=256-workgroup XCD-aware tile order against the small-grid order);
    (2) one bf16 training step against the fp32 GPU step (itself pinned to the oracle at small sizes):
    outputs, gradients and their cosine within the bounds frozen from the reference's bf16-autocast run of
    the same network (case f48); (3) the split-output data gradients at 144/288/576/1152 channels give
    finite, non-zero gradients for every parameter."""
    from unet_amd import OptimizedUNet
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=42)
    x, y, _ = W.synthetic_batch(64, 4, 256, 256, seed=42)
    xd, yd = x.cuda(), y.cuda()
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        m = OptimizedUNet(4, 1, 48, compute_dtype=dt)
        m.load_state_dict(sd)
        m = m.cuda()
        m.eval()
        full = m(xd)
        few = m(xd[:3])
        assert torch.equal(full[:3], few), dt
        m.train()
        m.dropout_masks_override = [None] * 9
        out = m(xd)
        ((out - yd) ** 2).mean().backward()
        torch.cuda.synchronize()
        res[dt] = (out.detach().cpu(), torch.cat([p.grad.double().cpu().reshape(-1) for p in m.parameters()]))
        for n, p in m.named_parameters():
            assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n
        del m, out, full, few
        torch.cuda.empty_cache()
    (o16, g16), (o32, g32) = res[torch.bfloat16], res[torch.float32]
    b_out, b_g, b_cos, _, _ = bf16_bounds(golden("bf16"), "unet.f48")
    e_out, e_g, cos = rel(o16, o32), rel(g16, g32), _cosine(g16, g32)
    print(f"config 2 full size, bf16 vs fp32 on the GPU: out {e_out:.3e} grads {e_g:.3f} cos {cos:.3f} "
          f"(bounds {b_out:.3e} / {b_g:.3f} / {b_cos:.3f})")
    assert e_out < b_out and e_g < b_g and cos > b_cos


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
        assert rel(out, ref) < bf16_bounds(golden("bf16"), "unet.small")[0]
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


def test_backward_fusions_agree_with_the_separate_passes(monkeypatch):
    """The two fusions of the backward pass that are switched per layer - the pool gradient formed inside the
    BatchNorm-backward passes (same bits as the stored tensor) and the BatchNorm-backward sums left by the
    data-gradient conv (another summation order: 1e-5 of the sums) - against the separate passes on the 48-feature
    case, where every layer from 96 channels up takes the second one."""
    from unet_amd import unet as U
    m, sd, x, y = build("f48", torch.bfloat16)
    m.train()
    m.dropout_masks_override = [None] * 9
    res = {}
    for sums, pool in ((True, True), (False, True), (False, False)):
        monkeypatch.setattr(U, "FUSE_BN_SUMS", sums)
        monkeypatch.setattr(U, "FUSE_POOL_BWD", pool)
        m.load_state_dict(sd)
        m.zero_grad()
        out = m(x.cuda())
        ((out - y.cuda()) ** 2).mean().backward()
        res[(sums, pool)] = (out.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    base_out, base = res[(False, False)]
    assert torch.equal(res[(False, True)][0], base_out) and torch.equal(res[(True, True)][0], base_out)
    assert all(torch.equal(res[(False, True)][1][n], g) for n, g in base.items())      # the pool fusion: same bits
    # (bf16 storage: a last-place change of a sum flips a rounding here and there, and the layers below amplify it -
    #  on this case the bf16 step as a whole sits 0.4..0.6 from its fp32 run, see the header)
    errs = sorted(rel(res[(True, True)][1][n], g) for n, g in base.items())
    print("fused-sums vs separate, relative L2 per parameter: median %.2e, worst %.2e" % (errs[len(errs) // 2], errs[-1]))
    assert errs[len(errs) // 2] < 1e-2 and errs[-1] < 5e-2, errs[-5:]


@pytest.mark.parametrize("masked", [False, True])
def test_bn_relu_inside_the_upsampling_gives_the_same_bits(monkeypatch, masked):
    """The last BatchNorm + ReLU + Dropout2d of the bottleneck and of decoder4..2 formed inside the 2x resize that is
    its only reader (no stored activation) against the two separate passes: output and every gradient, bit for bit;
    with all-ones factors and with live Dropout2d masks."""
    from unet_amd import unet as U
    m, sd, x, y = build("f48", torch.bfloat16)
    m.train()
    if masked:
        g = torch.Generator().manual_seed(11)
        B = x.shape[0]
        m.dropout_masks_override = [(torch.rand(B, getattr(m, n)[3].out_channels, generator=g) > 0.3).float() / 0.7
                                    for n in U.BLOCKS]
    else:
        m.dropout_masks_override = [None] * 9
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(U, "FUSE_BN_UPSAMPLE", fused)
        m.load_state_dict(sd)
        m.zero_grad()
        out = m(x.cuda())
        ((out - y.cuda()) ** 2).mean().backward()
        res[fused] = (out.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    assert all(torch.equal(res[True][1][n], g) for n, g in res[False][1].items())


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 9, 13, 64), (3, 16, 24, 96), (2, 5, 3, 32), (1, 1, 1, 32)])
@pytest.mark.parametrize("into_cat", [False, True])
def test_bn_act_upsample_op(shape, into_cat):
    """ops.bn_act_upsample against bn_act_fwd followed by the resize, dense and into a concat buffer whose other
    half it must leave alone; ragged tiles, images smaller than a tile."""
    from unet_amd import ops
    B, h, w, C1 = shape
    g = torch.Generator().manual_seed(3)
    y = torch.randn(B, h, w, C1, generator=g).cuda().to(torch.bfloat16)
    sc = (torch.rand(C1, generator=g) + 0.5).cuda()
    sh = (torch.randn(C1, generator=g) * 0.3).cuda()
    keep = ((torch.rand(B, C1, generator=g) > 0.2).float() / 0.8).cuda()
    assert ops.bn_act_upsample_supported(y)
    for kp in (None, keep):
        a, _ = ops.bn_act_fwd(y, sc, sh, kp, relu=True)
        if into_cat:
            cat = torch.full((B, 2 * h, 2 * w, C1 + 16), 7.0, device="cuda", dtype=torch.bfloat16)
            want = ops.upsample_into(a, cat.clone())
            got = ops.bn_act_upsample(y, sc, sh, kp, cat=cat)
        else:
            want = ops.upsample(a, 2 * h, 2 * w)
            got = ops.bn_act_upsample(y, sc, sh, kp)
        assert torch.equal(got, want)
        # the plain resize takes the tiled kernel too: against the one-thread-per-pixel kernel
        import os
        os.environ["UNETK_UPSAMPLE_NO_TILE"] = "1"
        try:
            old = ops.upsample_into(a, cat.clone()) if into_cat else ops.upsample(a, 2 * h, 2 * w)
        finally:
            del os.environ["UNETK_UPSAMPLE_NO_TILE"]
        assert torch.equal(old, want)
