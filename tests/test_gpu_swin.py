"""GPU parity of the Swin-UNet kernels and of unet_amd.SwinUNet against the CPU oracle and goldens.

float32: outputs 1e-4 relative L2 (the stated bound).  Gradients are compared with the float64
oracle: with batch statistics over a handful of elements in the decoder the fp32 gradients of the
reference itself carry ~5e-3 of rounding noise (see tests/golden/make_golden.py), so the whole-model
gradient bound is 2e-2 while every kernel is pinned on its own at 1e-4.
bfloat16: single kernels 2e-2; the whole model in eval mode 2e-2 (the stated bound; the reference's own
bf16-autocast eval run is 2.03e-2..2.06e-2 from its fp32 run on these cases).  Training mode is bounded
by the reference's bf16-autocast run frozen in tests/golden/bf16.npz (no worse than 1.3x its distance
from fp32, with caps and a cosine floor).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import swin_oracle
from oracle import weights as W

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(shape, seed, dt, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt).float()


def sub(t, n):
    f = torch.as_tensor(t).detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().cpu().numpy()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [96, 384, 1536])
def test_layernorm_fwd_bwd(dt, C):
    from unet_amd import ops
    x = rnd((2, 5, 7, C), 1, dt) * 1.3 + 0.2
    x = x.to(dt).float()
    g, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    dy = rnd((2, 5, 7, C), 2, dt)
    add = rnd((2, 5, 7, C), 3, dt)
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br)
    ref.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x.to("cuda", dt), g.cuda(), b.cuda())
    assert rel(y, ref) < TOL[dt]
    dx, dg, db = ops.layernorm_bwd(dy.to("cuda", dt), x.to("cuda", dt), g.cuda(), mean, rstd, add=add.to("cuda", dt))
    assert rel(dx, xr.grad + add) < TOL[dt]
    assert rel(dg, gr.grad) < TOL[dt] and rel(db, br.grad) < TOL[dt]
    assert rel(ops.channel_sum(dy.to("cuda", dt)), dy.sum((0, 1, 2))) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_gelu_scale_merge_shuffle(dt):
    from unet_amd import ops
    x = rnd((3, 6, 5, 64), 4, dt) * 2
    xr = x.clone().requires_grad_(True)
    ref = F.gelu(xr)
    dy = rnd(tuple(x.shape), 5, dt)
    ref.backward(dy)
    xd = x.to("cuda", dt)
    assert rel(ops.gelu_fwd(xd), ref) < TOL[dt]
    assert rel(ops.gelu_bwd(dy.to("cuda", dt), xd), xr.grad) < TOL[dt]
    keep = torch.tensor([0.0, 1.25, 1.25])
    assert rel(ops.scale_sample(xd, keep.cuda(), xd), x * keep.view(3, 1, 1, 1) + x) < TOL[dt]
    # patch merging gather, odd sizes (zero pad) and its inverse
    s = x
    sp = F.pad(s, (0, 0, 0, 5 % 2, 0, 6 % 2))
    want = torch.cat([sp[:, 0::2, 0::2], sp[:, 1::2, 0::2], sp[:, 0::2, 1::2], sp[:, 1::2, 1::2]], -1)
    got = ops.merge_gather(xd)
    assert torch.equal(got.float().cpu(), want)
    back = ops.merge_scatter(got, 6, 5)
    assert torch.equal(back.float().cpu(), x)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 24, 3, 0), (2, 16, 24, 3, 4), (1, 8, 8, 6, 4), (1, 32, 8, 12, 4),
                                  (20, 64, 64, 3, 4), (5, 32, 64, 6, 0)])
def test_window_attention_fwd_bwd(dt, case):
    """Shifted and unshifted windows, the single-window stage (mask still applied), several heads; the last two
    cases give every persistent wave of the bfloat16 kernels an uneven run of several (window, head) items."""
    from unet_amd import ops
    B, H, Wd, heads, shift = case
    C = heads * 32
    sd = {"qkv.weight": torch.zeros(3 * C, C), "qkv.bias": torch.zeros(3 * C), "proj.weight": torch.eye(C),
          "proj.bias": torch.zeros(C),
          "relative_position_bias_table": rnd((225, heads), 6, torch.float32, 0.5),
          "relative_position_index": swin_oracle.relative_position_index(8)}
    qkv = rnd((B, H, Wd, 3 * C), 7, dt, 0.8)
    dout = rnd((B, H, Wd, C), 8, dt)
    mask = swin_oracle.shift_mask(H, Wd, 8)

    def ref_attn(qkv_t, table):
        # the oracle's attention on pre-computed qkv: identity qkv/proj weights are bypassed by hand
        h = qkv_t
        if shift:
            h = torch.roll(h, (-shift, -shift), (1, 2))
        wins = swin_oracle._partition(h, 8)
        Bn, N, _ = wins.shape
        q, k, v = wins.view(Bn, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
        a = (q * 32 ** -0.5) @ k.transpose(-2, -1)
        bias = table[sd["relative_position_index"].view(-1)].view(N, N, heads).permute(2, 0, 1)
        a = a + bias.unsqueeze(0)
        nW = mask.shape[0]
        a = (a.view(Bn // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N).softmax(-1)
        y = (a @ v).transpose(1, 2).reshape(Bn, N, C)
        y = swin_oracle._unpartition(y, 8, H, Wd)
        if shift:
            y = torch.roll(y, (shift, shift), (1, 2))
        return y

    qr = qkv.double().requires_grad_(True)
    tr_ = sd["relative_position_bias_table"].double().requires_grad_(True)
    ref = ref_attn(qr, tr_)
    ref.backward(dout.double())
    table = sd["relative_position_bias_table"].cuda()
    out = ops.window_attn_fwd(qkv.to("cuda", dt), table, heads, shift, True)
    assert rel(out, ref) < TOL[dt]
    dqkv, dtable = ops.window_attn_bwd(qkv.to("cuda", dt), dout.to("cuda", dt), table, heads, shift, True)
    assert rel(dqkv, qr.grad) < TOL[dt] * (2 if dt == torch.bfloat16 else 1)
    assert rel(dtable, tr_.grad) < TOL[dt] * (2 if dt == torch.bfloat16 else 1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 8, 12, 3, 2, 4, 16), (1, 14, 21, 2, 3, 7, 32), (2, 16, 8, 2, 4, 8, 64),
                                  (1, 24, 12, 3, 6, 12, 16), (3, 8, 8, 4, 0, 4, 32), (1, 16, 16, 1, 8, 16, 32),
                                  (1, 12, 12, 2, 6, 12, 64), (2, 6, 9, 5, 1, 3, 24)])
def test_window_attention_any_geometry(dt, case):
    """Window sides and head dimensions other than 8 / 32 (the general kernels): forward, the gradient of qkv and of
    the relative-position table against the oracle's attention in float64, shifted and unshifted."""
    from unet_amd import ops
    B, H, Wd, heads, shift, ws, hd = case
    C = heads * hd
    t2 = (2 * ws - 1) ** 2
    assert not ops.window_attn_is_mfma(C, heads, ws)
    table0 = rnd((t2, heads), 6, torch.float32, 0.5)
    index = swin_oracle.relative_position_index(ws)
    qkv = rnd((B, H, Wd, 3 * C), 7, dt, 0.8)
    dout = rnd((B, H, Wd, C), 8, dt)
    mask = swin_oracle.shift_mask(H, Wd, ws)

    def ref_attn(h, table):
        if shift:
            h = torch.roll(h, (-shift, -shift), (1, 2))
        wins = swin_oracle._partition(h, ws)
        Bn, N, _ = wins.shape
        q, k, v = wins.view(Bn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = (q * hd ** -0.5) @ k.transpose(-2, -1)
        a = a + table[index.view(-1)].view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
        nW = mask.shape[0]
        a = (a.view(Bn // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N).softmax(-1)
        y = swin_oracle._unpartition((a @ v).transpose(1, 2).reshape(Bn, N, C), ws, H, Wd)
        return torch.roll(y, (shift, shift), (1, 2)) if shift else y

    qr = qkv.double().requires_grad_(True)
    tr_ = table0.double().requires_grad_(True)
    ref = ref_attn(qr, tr_)
    ref.backward(dout.double())
    table = table0.cuda()
    out = ops.window_attn_fwd(qkv.to("cuda", dt), table, heads, shift, True, ws=ws)
    assert rel(out, ref) < TOL[dt]
    dqkv, dtable = ops.window_attn_bwd(qkv.to("cuda", dt), dout.to("cuda", dt), table, heads, shift, True, ws=ws)
    assert dtable.shape == (t2, heads)
    assert rel(dqkv, qr.grad) < TOL[dt] * (2 if dt == torch.bfloat16 else 1)
    assert rel(dtable, tr_.grad) < TOL[dt] * (2 if dt == torch.bfloat16 else 1)


@pytest.mark.parametrize("shape", [(2, 5, 7, 96, 288), (1, 16, 16, 384, 96), (3, 8, 8, 192, 768), (1, 8, 8, 768, 2304),
                                   (2, 9, 4, 64, 96), (1, 24, 24, 1536, 384), (2, 16, 8, 512, 1024), (1, 4, 4, 96, 32)])
def test_token_linear_gemm(shape):
    """The bfloat16 token-linear GEMM kernel against torch on exactly representable inputs: plain, bias,
    GELU with the pre-activation as second output, GELU' multiplier (data gradient of the MLP), residual with a
    per-sample scale, output into a channel slice of a wider buffer, token counts that are not tile multiples."""
    from unet_amd import ops
    B, H, Wd, K, N = shape
    dt = torch.bfloat16
    x = rnd((B, H, Wd, K), 21, dt, 0.7)
    w = rnd((N, K), 22, dt, 1.0 / K ** 0.5)
    b = rnd((N,), 23, torch.float32, 0.3)
    xd, bd = x.to("cuda", dt), b.cuda()
    pw = ops.PackedConv(w.cuda(), dt)
    assert ops.linear_supported(xd, pw)
    ref = F.linear(x.double(), w.double())
    assert rel(ops.linear_fwd(xd, pw), ref) < 1e-2
    refb = ref + b.double()
    pre, act = ops.linear_fwd(xd, pw, bias=bd, act=2, out2=True)
    assert rel(pre, refb) < 1e-2
    assert rel(act, F.gelu(pre.double().cpu())) < 1e-2
    assert rel(ops.linear_fwd(xd, pw, bias=bd, act=2), F.gelu(refb)) < 1e-2
    h = rnd((B, H, Wd, N), 24, dt, 1.5)
    hr = h.double().requires_grad_(True)
    F.gelu(hr).backward(ref)
    assert rel(ops.linear_fwd(xd, pw, gelu_grad_of=h.to("cuda", dt)), hr.grad) < 1e-2
    res = rnd((B, H, Wd, N), 25, dt)
    keep = torch.tensor([0.0, 1.25, 1.0][:B])
    got = ops.linear_fwd(xd, pw, bias=bd, resid=res.to("cuda", dt), resid_keep=keep.cuda())
    assert rel(got, res.double() + keep.view(B, 1, 1, 1).double() * refb) < 1e-2
    wide = torch.full((B, H, Wd, N + 64), 7.0, dtype=dt, device="cuda")
    ops.linear_fwd(xd, pw, out=wide[..., 64:])
    assert rel(wide[..., 64:], ref) < 1e-2 and torch.all(wide[..., :64] == 7.0)
    # the data-gradient form: transposed pack
    pwt = ops.PackedConv(w.cuda(), dt, transpose=True)
    dy = rnd((B, H, Wd, N), 26, dt)
    if ops.linear_supported(dy.to("cuda", dt), pwt):
        assert rel(ops.linear_fwd(dy.to("cuda", dt), pwt), dy.double() @ w.double()) < 1e-2


@pytest.mark.parametrize("shape", [(2, 5, 7, 96, 288), (1, 16, 16, 384, 96), (3, 8, 8, 192, 768), (1, 8, 8, 768, 2304),
                                   (2, 9, 4, 64, 96), (4, 32, 32, 96, 96), (2, 16, 8, 512, 1024), (1, 4, 4, 96, 32),
                                   (8, 64, 64, 96, 384)])
def test_token_weight_gradient_gemm(shape):
    """dw = dy^T x and db = column sums of dy from the bf16 token weight-gradient kernel against float64, on
    exactly representable inputs; token counts that are not stage multiples, channel counts that are not tile
    multiples, strided (channel-slice) operands."""
    from unet_amd import ops
    B, H, Wd, K, N = shape
    dt = torch.bfloat16
    x = rnd((B, H, Wd, K), 31, dt, 0.7)
    dy = rnd((B, H, Wd, N), 32, dt, 0.5)
    xd, dyd = x.to("cuda", dt), dy.to("cuda", dt)
    assert ops.token_wgrad_supported(xd, dyd, K, N)
    dw, db = ops.token_wgrad(xd, dyd, K, N)
    want = dy.double().reshape(-1, N).t() @ x.double().reshape(-1, K)
    assert rel(dw, want) < 1e-5, rel(dw, want)
    assert rel(db, dy.double().reshape(-1, N).sum(0)) < 1e-5
    dw2, none = ops.token_wgrad(xd, dyd, K, N, want_bias=False)
    assert none is None and torch.equal(dw2, dw)
    # operands that are channel slices of wider tensors
    xw = torch.zeros((B, H, Wd, K + 16), dtype=dt, device="cuda")
    xw[..., 8:8 + K] = xd
    dw3, _ = ops.token_wgrad(xw[..., 8:8 + K], dyd, K, N)
    # (the same bits from the same kernel; the narrow layers over many tokens take the whole-row kernel only with
    # dense operands, and the tile kernel sums in another order)
    assert torch.equal(dw3, dw) or (B * H * Wd >= 2048 and rel(dw3, dw) < 1e-6)


def test_token_weight_gradient_every_stage_count():
    """The tile kernel's stage loop has a six-fold steady part, up to five single steady stages and a general tail,
    and two fragment register sets that alternate: every stage count from 1 to 40 (the last stage ragged), and a
    split token axis, against float64."""
    from unet_amd import ops
    dt = torch.bfloat16
    K, N = 384, 512
    g = torch.Generator().manual_seed(77)
    for M in [32 * n - r for n in range(1, 41) for r in ((0,) if n % 3 else (0, 5))] + [32 * 700 + 3]:
        x = (torch.randint(-4, 5, (1, 1, M, K), generator=g).float() * 0.25).to(dt)
        dy = (torch.randint(-4, 5, (1, 1, M, N), generator=g).float() * 0.5).to(dt)
        dw, db = ops.token_wgrad(x.cuda(), dy.cuda(), K, N)
        want = dy.double().reshape(-1, N).t() @ x.double().reshape(-1, K)
        assert rel(dw, want) < 1e-6, (M, rel(dw, want))
        assert rel(db, dy.double().reshape(-1, N).sum(0)) < 1e-6, M


@pytest.mark.parametrize("shape", [(2, 8, 8, 768, 512), (1, 16, 12, 512, 256), (3, 5, 7, 256, 128), (2, 64, 64, 128, 64)])
def test_conv_transpose_gemm_with_shuffled_stores(shape):
    """ConvTranspose2d(k=2, s=2) + bias as one GEMM launch whose epilogue stores each value at its shuffled pixel:
    against GEMM + pixel-shuffle kernel (which rounds to bfloat16 twice: before and after the bias) and
    F.conv_transpose2d, to bf16 accuracy; ragged token counts, the first half of a wider concat buffer."""
    from unet_amd import ops
    B, h, w, K, up = shape
    dt = torch.bfloat16
    x = rnd((B, h, w, K), 41, dt, 0.7).to("cuda", dt)
    wt = rnd((K, up, 2, 2), 42, dt, (1.0 / K) ** 0.5)                       # ConvTranspose2d weight (in, out, kh, kw)
    bias = (torch.randn(up) * 0.1).cuda()
    w2 = wt.cuda().permute(2, 3, 1, 0).reshape(4 * up, K).contiguous()
    pk = ops.PackedConv(w2, dt)
    assert ops.linear_supported(x, pk)
    cat = torch.full((B, 2 * h, 2 * w, 2 * up), 3.0, dtype=dt, device="cuda")
    ops.convT2x2_fwd(x, pk, bias, cat)
    old = torch.full_like(cat, 3.0)
    ops.pixel_shuffle(ops.linear_fwd(x, pk), old, bias, up)
    assert torch.all(cat[..., up:] == 3.0) and rel(cat[..., :up], old[..., :up]) < 5e-3
    ref = F.conv_transpose2d(x.float().cpu().permute(0, 3, 1, 2), wt, bias.cpu(), stride=2).permute(0, 2, 3, 1)
    assert rel(cat[..., :up], ref) < 1e-2


def build(dt):
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    m = SwinUNet(cfg, compute_dtype=dt)
    m.load_state_dict(sd)
    return m.cuda(), sd, cfg


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("name,B,H,Wd", [("s64", 2, 64, 64), ("s128x64", 1, 128, 64), ("s256", 1, 256, 256)])
def test_swin_eval_forward(golden, dt, name, B, H, Wd):
    m, sd, cfg = build(dt)
    m.eval()
    x, _, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    out = m(x.cuda())
    ref, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
    assert out.shape == ref.shape == (B, 1, H // 2, Wd // 2)
    e = rel(out, ref)
    print(f"swin eval {name} {dt}: rel {e:.2e}")
    assert e < TOL[dt]
    if dt == torch.float32:
        assert rel(sub(out, 16384), golden("swin")[f"{name}.eval_out"]) < 1e-4


def test_swin_eval_resized_skip_branch(golden):
    """24-px input: 6x6 / 3x3 / 2x2 / 1x1 token grids (padded windows in every stage, odd-size merging
    pad) and a decoder whose projected skips (3x3, 6x6) are bilinearly resized to the up-sampled size
    (4x4, 8x8), swin_unet_model.py:526-529."""
    x, _, _ = W.synthetic_batch(2, 4, 24, 24, seed=42)
    for dt in DTYPES:
        m, sd, cfg = build(dt)
        m.eval()
        out = m(x.cuda())
        ref, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
        assert tuple(out.shape) == tuple(ref.shape) == tuple(golden("swin2")["s24.out_shape"])
        e = rel(out, ref)
        print(f"swin eval s24 {dt}: rel {e:.2e}")
        assert e < TOL[dt]
        if dt == torch.float32:
            assert rel(sub(out, 16384), golden("swin2")["s24.eval_out"]) < 1e-4


@pytest.mark.parametrize("name,B,H,Wd", [("s64", 2, 64, 64), ("s128x64", 1, 128, 64), ("s256", 1, 256, 256),
                                          ("s24", 2, 24, 24)])
def test_swin_train_forward_backward_fp32(golden, name, B, H, Wd):
    """s256 is BASELINE config 3's geometry (no padding, 64/16/4/1 windows per image); s24 takes the
    padded-window path in every stage and the resized-skip decoder branch."""
    g = golden("swin" if name in ("s64", "s128x64") else "swin2")
    m, sd, cfg = build(torch.float32)
    m.train()
    m.drop_path_override = [[None, None]] * sum(cfg["depths"])
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    if name in ("s64", "s128x64"):
        tgt = F.avg_pool2d(y, 2)
    else:
        tgt = F.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False).float()
    loss = ((out - tgt.cuda()) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-4
    assert abs(loss.item() - float(g[f"{name}.loss"])) < 1e-4 * float(g[f"{name}.loss"])
    # float64 oracle gradients
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    xo = x.double().requires_grad_(True)
    o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
    ((o_out - tgt.double()) ** 2).mean().backward()
    names = [n for n, _ in m.named_parameters() if n != "final_conv.0.bias"]
    mine = torch.cat([dict(m.named_parameters())[n].grad.double().cpu().reshape(-1) for n in names])
    want = torch.cat([params[n].grad.reshape(-1) for n in names])
    worst = max((rel(dict(m.named_parameters())[n].grad, params[n].grad), n) for n in names)
    e_all, e_in = rel(mine, want), rel(xd.grad, xo.grad)
    print(f"swin train {name}: out {rel(out, o_out):.2e} grads(all) {e_all:.2e} worst {worst[0]:.2e} {worst[1]} dx {e_in:.2e}")
    assert rel(out, o_out) < 1e-4
    assert e_all < 2e-2 and e_in < 2e-2 and worst[0] < 1e-1, (e_all, e_in, worst)
    assert float(m.final_conv[0].bias.grad.abs().max()) == 0.0
    norms = np.array([p.grad.double().norm().item() for n, p in m.named_parameters()])
    keep = np.array([n != "final_conv.0.bias" for n, _ in m.named_parameters()])
    assert rel(norms[keep], g[f"{name}.grad_norms"][keep]) < 2e-2
    msd = m.state_dict()
    for k, v in o_stats.items():
        assert rel(msd[k], v) < 1e-4, k


GEOM_CASES = {      # as tests/golden/make_golden.py SWIN_GEOM_CASES
    "w4h16": (dict(window_size=4, embed_dim=48, depths=[2, 2, 2, 2], decoder_channels=[64, 48, 32, 16]), (2, 64, 64)),
    "w7h32": (dict(window_size=7, depths=[2, 2, 2, 2], decoder_channels=[128, 64, 32, 16]), (1, 112, 112)),
    "w8h64": (dict(window_size=8, embed_dim=128, num_heads=[2, 4, 8, 16], depths=[2, 2, 2, 2],
                   decoder_channels=[128, 64, 32, 16]), (1, 64, 64)),
}


@pytest.mark.parametrize("name", sorted(GEOM_CASES))
def test_swin_other_window_and_head_geometries(golden, name):
    """Configurations the reference accepts beyond its default (swin_unet_model.py:545-554): window 4 with 16-wide
    heads (48 channels: the linears leave the GEMM path too), window 7 (token grids of 28 / 14 / 7 and a padded 4),
    window 8 with 64-wide heads.  fp32 eval + train forward / backward against the float64 oracle and the numbers
    frozen from the reference; bf16 eval inside 2e-2."""
    from unet_amd.swin import SwinUNet
    g = golden("swin_geom")
    over, (B, H, Wd) = GEOM_CASES[name]
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    cfg.update(over)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    o_eval, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
    for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
        m = SwinUNet(cfg, compute_dtype=dt)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        with torch.no_grad():
            out = m(x.cuda())
        assert tuple(out.shape) == tuple(g[f"{name}.out_shape"])
        assert rel(out, o_eval) < tol, (dt, rel(out, o_eval))
        if dt == torch.float32:
            assert rel(sub(out, 16384), g[f"{name}.eval_out"]) < 1e-4
    m = SwinUNet(cfg, compute_dtype=torch.float32)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.drop_path_override = [[None, None]] * sum(cfg["depths"])
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    tgt = F.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False)
    loss = ((out - tgt.float().cuda()) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-4
    assert abs(loss.item() - float(g[f"{name}.loss"])) < 1e-4 * float(g[f"{name}.loss"])
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    xo = x.double().requires_grad_(True)
    o_out, _ = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
    ((o_out - tgt) ** 2).mean().backward()
    mine = dict(m.named_parameters())
    assert list(mine) == [str(n) for n in g[f"{name}.param_names"]]
    names = [n for n in mine if n != "final_conv.0.bias"]
    flat = torch.cat([mine[n].grad.double().cpu().reshape(-1) for n in names])
    want = torch.cat([params[n].grad.reshape(-1) for n in names])
    worst = max((rel(mine[n].grad, params[n].grad), n) for n in names)
    e_all, e_in = rel(flat, want), rel(xd.grad, xo.grad)
    print(f"swin {name}: out {rel(out, o_out):.2e} grads {e_all:.2e} worst {worst[0]:.2e} {worst[1]} dx {e_in:.2e}")
    assert rel(out, o_out) < 1e-4
    assert e_all < 2e-2 and e_in < 2e-2 and worst[0] < 1e-1, (e_all, e_in, worst)
    norms = np.array([p.grad.double().norm().item() for n, p in mine.items()])
    keep = np.array([n != "final_conv.0.bias" for n in mine])
    assert rel(norms[keep], g[f"{name}.grad_norms"][keep]) < 2e-2
    assert rel(sub(xd.grad, 4096), g[f"{name}.grad_input"]) < 2e-2


@pytest.mark.parametrize("name,B,H,Wd", [("s64", 2, 64, 64), ("s128x64", 1, 128, 64), ("s256", 1, 256, 256)])
def test_swin_train_forward_backward_bf16(golden, name, B, H, Wd):
    """bf16 storage (the benchmarked path, including the bf16-only fused weight + bias gradient of the
    linear layers) against the fp32 oracle, bounded by the reference's own bf16-autocast run."""
    from tests.test_gpu_unet import _cosine, bf16_bounds
    gb = golden("bf16")
    m, sd, cfg = build(torch.bfloat16)
    m.train()
    m.drop_path_override = [[None, None]] * sum(cfg["depths"])
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    tgt = F.avg_pool2d(y, 2)
    ((out - tgt.cuda()) ** 2).mean().backward()
    torch.cuda.synchronize()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd)
    full.update(params)
    xo = x.clone().requires_grad_(True)
    o_out, _ = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
    ((o_out - tgt) ** 2).mean().backward()
    names = [n for n, _ in m.named_parameters() if n != "final_conv.0.bias"]
    mine = torch.cat([dict(m.named_parameters())[n].grad.double().cpu().reshape(-1) for n in names])
    want = torch.cat([params[n].grad.double().reshape(-1) for n in names])
    b_out, b_g, b_cos, b_dx, b_dxcos = bf16_bounds(gb, f"swin.{name}")
    e_out, e_g, cos = rel(out, o_out), rel(mine, want), _cosine(mine, want)
    e_dx, dxcos = rel(xd.grad, xo.grad), _cosine(xd.grad, xo.grad)
    print(f"swin train {name} bf16 vs fp32: out {e_out:.3e} grads {e_g:.3f} cos {cos:.4f} dx {e_dx:.3f} cos {dxcos:.4f}; "
          f"reference autocast out {float(gb[f'swin.{name}.out_rel']):.3e} grads {float(gb[f'swin.{name}.grad_rel']):.3f} "
          f"cos {float(gb[f'swin.{name}.grad_cos']):.4f} dx {float(gb[f'swin.{name}.dx_rel']):.3f}")
    assert e_out < b_out and e_g < b_g and cos > b_cos, (e_out, b_out, e_g, b_g, cos, b_cos)
    assert e_dx < b_dx and dxcos > b_dxcos, (e_dx, b_dx, dxcos, b_dxcos)
    assert float(m.final_conv[0].bias.grad.abs().max()) == 0.0


def test_swin_training_step_matches_frozen_reference(golden):
    """The Swin trainer's step (swin_unet_model.py:809-823): model, bilinear resize of the half-resolution
    prediction to the label size, MultiLossFunction('tid_gradient'), backward, clip_grad_norm_(1.0),
    AdamW(1e-4, 0.05), all on the HIP kernels, against the step frozen from the reference."""
    from oracle import loss_oracle
    from unet_amd import ops
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    g = golden("step_swin")
    m, sd, cfg = build(torch.float32)
    m.train()
    m.drop_path_override = [[None, None]] * sum(cfg["depths"])
    x, y, w = W.synthetic_batch(2, 4, 64, 64, seed=42)
    crit = MultiLossFunction(loss_oracle.preset("weighted_mse_grad", grad_w=0.3, spec_w=0.0)).cuda()
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
    outputs = m(x.cuda())
    assert outputs.shape[-2:] == (32, 32)
    outputs = ops.interpolate_bilinear(outputs, y.shape[-2:])
    loss, parts = crit(outputs, y.cuda(), w.cuda())
    opt.zero_grad()
    loss.backward()
    names = [str(s) for s in g["param_names"]]
    now = dict(m.named_parameters())
    gnorms = np.array([now[k].grad.double().norm().item() for k in names])
    opt.step()
    got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
    assert rel(got, g["parts"]) < 1e-4
    keep = np.array([k != "final_conv.0.bias" for k in names])
    print("swin step: grad norm", opt.last_grad_norm.item(), "frozen", float(g["grad_norm"]),
          "per-parameter norms rel", rel(gnorms[keep], g["grad_norms"][keep]))
    assert abs(opt.last_grad_norm.item() - float(g["grad_norm"])) < 1e-2 * float(g["grad_norm"])
    assert rel(gnorms[keep], g["grad_norms"][keep]) < 2e-2
    delta = np.array([(now[k].detach().cpu() - sd[k]).double().norm().item() for k in names])
    assert rel(delta[keep], g["param_delta_norms"][keep]) < 2e-2     # first Adam step ~ lr*sign(g): flips where g ~ 0
    heads = np.concatenate([sub(now[k], 8) for k in names])
    assert rel(heads, g["param_heads"]) < 1e-4


def test_interpolate_bilinear_matches_torch():
    from unet_amd import ops
    for (h, w_, H, Wd) in ((32, 32, 64, 64), (7, 5, 16, 9), (12, 12, 12, 12), (16, 16, 8, 8)):
        x = rnd((2, 1, h, w_), 11, torch.float32)
        xr = x.clone().requires_grad_(True)
        ref = F.interpolate(xr, size=(H, Wd), mode="bilinear", align_corners=False)
        dy = rnd((2, 1, H, Wd), 12, torch.float32)
        ref.backward(dy)
        xd = x.cuda().requires_grad_(True)
        out = ops.interpolate_bilinear(xd, (H, Wd))
        out.backward(dy.cuda())
        assert rel(out, ref) < 1e-6 and rel(xd.grad, xr.grad) < 1e-6, (h, w_, H, Wd)


def test_swin_seeded_droppath_and_errors(golden):
    m, sd, cfg = build(torch.float32)
    m.train()
    x, _, _ = W.synthetic_batch(2, 4, 64, 64, seed=42)
    torch.manual_seed(99)
    dp = swin_oracle.draw_drop_path(cfg, 2)
    m.drop_path_override = dp
    out = m(x.cuda())
    assert rel(sub(out, 16384), golden("swin")["s64.droppath_seed99_out"]) < 1e-4
    m.drop_path_override = None
    draws = m._draw_drop_path(512, torch.device("cuda"))
    assert draws[0] == [None, None]
    last = draws[-1][0]
    assert torch.all((last == 0) | ((last - 1 / 0.9).abs() < 1e-5)) and 0.05 < (last == 0).float().mean() < 0.15
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 96, 160, device="cuda"))        # 12x20 tokens at stage 1: not window multiples
    with pytest.raises(RuntimeError):
        m.cpu()(torch.zeros(1, 4, 64, 64))
    from unet_amd.swin import SwinUNet
    SwinUNet(dict(cfg, window_size=7))                      # (other windows: the general attention kernels)
    with pytest.raises(ValueError):
        SwinUNet(dict(cfg, window_size=20))                 # beyond them
    with pytest.raises(ValueError):
        SwinUNet(dict(cfg, embed_dim=384))                  # 128-wide heads
    with pytest.raises(ValueError):
        SwinUNet(dict(cfg, drop_rate=0.1))


@pytest.mark.parametrize("C,ntok", [(96, 4096), (192, 1000), (384, 333), (768, 64)])
def test_layernorm_backward_deferred_sums_same_bits(C, ntok):
    """The two-halves LayerNorm backward (dx + partial tables now, all dgamma / dbeta sums by one launch later)
    against the one-call form: same dx, same dgamma, same dbeta, bit for bit."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(C + ntok)
    dt = torch.bfloat16
    x = torch.randn(1, ntok, C, generator=g).cuda().to(dt)
    dy = torch.randn(1, ntok, C, generator=g).cuda().to(dt)
    add = torch.randn(1, ntok, C, generator=g).cuda().to(dt)
    gamma = torch.randn(C, generator=g).cuda()
    beta = torch.zeros(C).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    want = ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add)
    later = ops.DeferredSums()
    got = [ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add, defer=later) for _ in range(3)]
    assert len(later.jobs) == 3
    # ... and with the DropPath scaling of the next branch as a second output (4 samples, some dropped)
    keep = torch.tensor([0.0, 1 / 0.9, 1 / 0.9, 0.0]).cuda() if ntok % 4 == 0 else None
    if keep is not None:
        dx, dg, db, dxs = ops.layernorm_bwd(dy, x, gamma, mean, rstd, add=add, defer=later, scale_next=(keep, ntok // 4))
        got.append((dx, dg, db))
        assert torch.equal(dxs.view(torch.int16), ops.scale_sample(dx.view(4, ntok // 4, C), keep).view(1, ntok, C).view(torch.int16))
    later.flush()
    for dx, dg, db in got:
        assert torch.equal(dx, want[0]) and torch.equal(dg, want[1]) and torch.equal(db, want[2])


def test_bias_table_gradient_deferred_sum_same_bits():
    """Window-attention backward with the relative-position table gradient summed later, for several layers by one
    launch, against the in-place sum: the same dqkv and the same table gradient, bit for bit."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(11)
    later = ops.DeferredSums()
    want, got = [], []
    for B, H, W, heads, shift in ((2, 16, 16, 3, 0), (1, 8, 8, 12, 4), (3, 16, 8, 6, 4)):
        C = heads * 32
        qkv = torch.randn(B, H, W, 3 * C, generator=g).cuda().to(torch.bfloat16)
        dout = torch.randn(B, H, W, C, generator=g).cuda().to(torch.bfloat16)
        table = (torch.randn(225, heads, generator=g) * 0.02).cuda()
        want.append(ops.window_attn_bwd(qkv, dout, table, heads, shift, True))
        got.append(ops.window_attn_bwd(qkv, dout, table, heads, shift, True, defer=later))
    assert len(later.tables) == 3
    later.flush()
    for (dq0, dt0), (dq1, dt1) in zip(want, got):
        assert torch.equal(dq0, dq1) and torch.equal(dt0, dt1)


@pytest.mark.parametrize("B,h,w,Cin,up,extra", [(2, 8, 8, 768, 384, 384), (1, 16, 12, 128, 64, 0), (3, 5, 7, 256, 128, 128),
                                                (1, 64, 64, 128, 64, 64)])
def test_conv_transpose_data_gradient_gathers_its_operand(B, h, w, Cin, up, extra):
    """The ConvTranspose2d(k=2, s=2) data gradient as one GEMM that reads the gradient image in place, against the
    un-shuffle into a token matrix followed by the plain GEMM (the same products in the same order: same bits) and
    against torch's conv_transpose2d backward."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(B * h + Cin)
    dt = torch.bfloat16
    wt = (torch.randn(Cin, up, 2, 2, generator=g) * 0.05)
    dcat = torch.randn(B, 2 * h, 2 * w, up + extra, generator=g).to(dt)
    w2 = wt.permute(2, 3, 1, 0).reshape(4 * up, Cin).contiguous().cuda()
    pkT = ops.PackedConv(w2, dt, transpose=True)
    dc = dcat.cuda()
    got = ops.convT2x2_bwd_data(dc, up, pkT, h, w)
    assert got is not None
    dtmp = torch.empty((B, h, w, 4 * up), dtype=dt, device="cuda")
    ops.pixel_shuffle(dtmp, dc, None, up, backward=True)
    want = ops.linear_fwd(dtmp, pkT)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    x = torch.zeros(B, Cin, h, w, requires_grad=True)
    y = F.conv_transpose2d(x, wt.to(dt).float(), stride=2)
    y.backward(dcat[..., :up].float().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    assert ((got.float().cpu() - ref).norm() / ref.norm()).item() < 1e-2


def test_new_entry_points_reject_bad_arguments():
    """Host-side checks of the entry points added for the Swin backward: nothing is launched on a bad call."""
    from unet_amd import ops
    from unet_amd._lib import call, ptr, dtype_code, stream
    dt = torch.bfloat16
    x = torch.zeros(1, 8, 8, 96, device="cuda", dtype=dt)
    stat = torch.zeros(64, device="cuda")
    gamma = torch.ones(96, device="cuda")
    part = torch.zeros(64, 2, 96, device="cuda")
    with pytest.raises(ValueError):        # a scaled copy without the per-sample factors
        call("unetk_layernorm_bwd_partials", ptr(x), ptr(x), ptr(gamma), ptr(stat), ptr(stat), None, ptr(x), ptr(part),
             64, 96, dtype_code(dt), None, 64, ptr(x), stream())
    with pytest.raises(ValueError):        # tokens that are not whole samples
        call("unetk_layernorm_bwd_partials", ptr(x), ptr(x), ptr(gamma), ptr(stat), ptr(stat), None, ptr(x), ptr(part),
             64, 96, dtype_code(dt), ptr(stat), 48, ptr(x), stream())
    with pytest.raises(ValueError):
        call("unetk_partial_sums_batched", None, 1, 96, stream())
    with pytest.raises(ValueError):
        call("unetk_dtable_reduce_batched", ptr(stat), 0, 3, stream())
    w2 = torch.zeros(4 * 24, 64, device="cuda")
    pk = ops.PackedConv(w2, dt, transpose=True)
    dout = torch.zeros(1, 16, 16, 24, device="cuda", dtype=dt)
    assert ops.convT2x2_bwd_data(dout, 24, pk, 8, 8) is None          # up = 24: the caller takes the un-shuffle route
    with pytest.raises(ValueError):
        call("unetk_convt2x2_bwd_data", ptr(dout), 24, ptr(pk.data), ptr(x), 64, 1, 8, 8, 24, 64, pk.bn, dtype_code(dt),
             stream())
    with pytest.raises(ValueError):
        ops.pack_many([(torch.zeros(8, 4, 3, 3, device="cuda"), "convt")], dt)


def test_config3_full_size_properties(golden, monkeypatch):
    """BASELINE config 3 at the benchmarked size (batch 64 of 4x256x256), where the CPU oracle is too slow and where
    the grid-size-dependent choices switch (token split of the weight gradients, whole-row weight-gradient kernels
    from 2048 tokens, runs of several windows per persistent attention wave, the second stream):
    (1) eval mode is per-sample: the first samples of the batch-64 run equal a batch-3 run bit for bit;
    (2) one bf16 training step with live DropPath factors against the fp32 GPU step with the same factors (itself
    pinned to the oracle at 1x256x256) inside the bounds frozen from the reference's bf16-autocast run of s256;
    (3) the backward with its weight gradients on a second stream equals the one-stream backward bit for bit;
    (4) every parameter gradient is finite and non-zero."""
    from tests.test_gpu_unet import _cosine, bf16_bounds
    from unet_amd import swin as swin_mod
    x, y, _ = W.synthetic_batch(64, 4, 256, 256, seed=42)
    xd = x.cuda()
    tgt = F.avg_pool2d(y, 2).cuda()
    torch.manual_seed(4)
    dp = swin_oracle.draw_drop_path(swin_oracle.DEFAULT_CONFIG, 64)
    assert sum(int((k == 0).sum()) for pair in dp for k in pair if k is not None) > 20
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        m, sd, cfg = build(dt)
        m.eval()
        full = m(xd)
        few = m(xd[:3])
        assert torch.equal(full[:3], few), dt
        del full, few
        m.train()
        m.drop_path_override = dp
        runs = []
        for overlap in ((True, False) if dt == torch.bfloat16 else (True,)):
            monkeypatch.setattr(swin_mod, "OVERLAP_WGRAD", overlap)
            m.load_state_dict(sd)
            m.zero_grad()
            out = m(xd)
            ((out - tgt) ** 2).mean().backward()
            torch.cuda.synchronize()
            runs.append((out.detach().clone(), [p.grad.clone() for p in m.parameters()]))
        if len(runs) == 2:
            assert torch.equal(runs[0][0], runs[1][0])
            for (n, _), a, b in zip(m.named_parameters(), runs[0][1], runs[1][1]):
                assert torch.equal(a, b), n
        for n, p in m.named_parameters():
            assert torch.isfinite(p.grad).all(), n
            assert n == "final_conv.0.bias" or float(p.grad.abs().max()) > 0, n
        names = [n for n, _ in m.named_parameters() if n != "final_conv.0.bias"]
        pg = dict(m.named_parameters())
        res[dt] = (runs[0][0].cpu(), torch.cat([pg[n].grad.double().cpu().reshape(-1) for n in names]))
        del m, runs, out
        torch.cuda.empty_cache()
    (o16, g16), (o32, g32) = res[torch.bfloat16], res[torch.float32]
    b_out, b_g, b_cos, _, _ = bf16_bounds(golden("bf16"), "swin.s256")
    e_out, e_g, cos = rel(o16, o32), rel(g16, g32), _cosine(g16, g32)
    print(f"config 3 full size, bf16 vs fp32 on the GPU: out {e_out:.3e} grads {e_g:.3f} cos {cos:.4f} "
          f"(bounds {b_out:.3e} / {b_g:.3f} / {b_cos:.4f})")
    assert e_out < b_out and e_g < b_g and cos > b_cos


def test_swin_bf16_loss_curve_tracks_fp32():
    """A learnable target (a fixed smooth function of the input at the model's half resolution), the Swin trainer's
    optimiser (AdamW, weight decay 0.05, clip 1.0) from the same initialisation in fp32 and in bf16 storage, DropPath
    live with the same factors in both runs.  The loss curves must agree within 10 % at every step and 4 % on average
    (measured 3.9 % / 1.0 %) while the loss falls by a factor of ~75, and the fp32 path's own response to a 1e-6 relative
    input perturbation must stay below 3 % (measured 0.7 %), else the comparison would measure the optimiser's noise.
    The learning rate is 5e-5: at the trainer's 1e-4 the loss is below 1 % of its start after 25 steps, the fp32 path
    then moves by up to 8 % under the 1e-6 perturbation and bf16 sits 10-25 % above fp32 (tools/exp_curve_swin.py
    prints the three rates); over the first 20 steps at 1e-4 the deviation is 4.3 % at most."""
    from unet_amd.optim import FusedAdamW
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    torch.manual_seed(7)
    B, H, Wd, steps = 8, 64, 64, 40
    x = F.avg_pool2d(torch.randn(B, 4, H + 4, Wd + 4), 5, 1)
    x = x / x.std()
    yf = torch.tanh(0.8 * x[:, :1] - 0.6 * x[:, 1:2] * x[:, 2:3]) + 0.3 * F.avg_pool2d(F.pad(x[:, 3:4], (1, 1, 1, 1)), 3, 1)
    y = F.avg_pool2d(yf, 2)
    init = SwinUNet(cfg).state_dict()
    noise = torch.randn_like(x)
    draws = [swin_oracle.draw_drop_path(cfg, B) for _ in range(steps)]

    def run(dt, eps=0.0):
        m = SwinUNet(cfg, compute_dtype=dt)
        m.load_state_dict(init)
        m = m.cuda().train()
        opt = FusedAdamW(m.parameters(), lr=5e-5, weight_decay=0.05, max_grad_norm=1.0)
        xd, yd = (x * (1 + eps * noise)).cuda(), y.cuda()
        losses = []
        for s in range(steps):
            m.drop_path_override = draws[s]
            opt.zero_grad()
            loss = ((m(xd) - yd) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return np.array(losses)

    f32, b16, f32_eps = run(torch.float32), run(torch.bfloat16), run(torch.float32, 1e-6)
    dev = np.abs(b16 - f32) / f32
    self_dev = np.abs(f32_eps - f32) / f32
    print("swin fp32 loss curve", np.round(f32, 4).tolist())
    print("swin bf16 loss curve", np.round(b16, 4).tolist(), "max deviation", dev.max(), "mean", dev.mean(),
          "; fp32 against itself with 1e-6 input noise: max", self_dev.max())
    assert f32[-1] < 0.05 * f32[0] and b16[-1] < 0.05 * b16[0]
    assert self_dev.max() < 0.03, self_dev
    assert dev.max() < 0.10 and dev.mean() < 0.04, dev


@pytest.mark.parametrize("C,B,H,Wd", [(96, 2, 16, 24), (96, 3, 9, 7), (192, 2, 8, 16), (192, 1, 5, 3), (96, 4, 64, 64)])
def test_fused_mlp_branch_matches_separate_launches(C, B, H, Wd):
    """The one-launch MLP branch of a Swin block (LayerNorm, fc1, GELU, fc2, residual + DropPath scale) against the
    four separate launches it replaces and against torch: LayerNorm output and statistics bit for bit (the same
    arithmetic lane for lane), pre-activation bit for bit (the same products in the same order), GELU and output to
    bf16 rounding (the stand-alone GELU pass evaluates erf through the library, the fused epilogue through a
    polynomial that is 1.5e-7 away).  Ragged token counts: tiles with rows past the end."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(C + H)
    dt = torch.bfloat16
    x = (torch.randn(B, H, Wd, C, generator=g) * 1.5 + 0.3).to(dt).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).cuda()
    b1 = (torch.randn(4 * C, generator=g) * 0.1).cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).cuda()
    b2 = (torch.randn(C, generator=g) * 0.1).cuda()
    keep = torch.tensor([1 / 0.9, 0.0, 1 / 0.9, 1 / 0.9][:B]).cuda()
    assert ops.swin_mlp_supported(x)
    p1, p2 = ops.pack_many([(w1, "bn64"), (w2, False)], dt)
    assert p1.bn == 64 and p2.bn == 96
    out, ln, mean, rstd, hpre, hact = ops.swin_mlp_fwd(x, gamma, beta, 1e-5, p1, b1, p2, b2, keep=keep)
    # the separate launches
    ln0, mean0, rstd0 = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    pw1, pw2 = ops.PackedConv(w1, dt), ops.PackedConv(w2, dt)
    hpre0 = ops.linear_fwd(ln0, pw1, bias=b1)
    hact0 = ops.gelu_fwd(hpre0)
    out0 = ops.linear_fwd(hact0, pw2, bias=b2, resid=x, resid_keep=keep)
    eq = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert eq(ln, ln0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    assert eq(hpre, hpre0)
    assert rel(hact, hact0) < 2e-3 and rel(out, out0) < 4e-3
    # without the stored GELU: the same pre-activation and output
    out_b, _, _, _, hpre_b, none = ops.swin_mlp_fwd(x, gamma, beta, 1e-5, p1, b1, p2, b2, keep=keep, want_act=False)
    assert none is None and eq(out_b, out) and eq(hpre_b, hpre)
    # torch, fp32 on the bf16-rounded inputs
    xf = x.float().cpu()
    lnr = F.layer_norm(xf, (C,), gamma.cpu(), beta.cpu())
    h = F.linear(lnr, w1.to(dt).float().cpu(), b1.cpu())
    ref = xf + keep.cpu().view(B, 1, 1, 1) * F.linear(F.gelu(h), w2.to(dt).float().cpu(), b2.cpu())
    assert rel(hpre, h) < 1e-2 and rel(out, ref) < 1e-2


@pytest.mark.parametrize("C,B,H,Wd", [(96, 2, 16, 24), (96, 3, 9, 7), (192, 2, 8, 16), (192, 1, 5, 3), (96, 4, 64, 64)])
def test_fused_mlp_backward_matches_separate_launches(C, B, H, Wd):
    """The data-gradient chain of the MLP branch in one launch (fc2's data gradient, GELU', fc1's data gradient,
    LayerNorm backward with the residual gradient added, the DropPath-scaled copy, the LayerNorm parameter-gradient
    partials) against the separate launches and against torch autograd.  The hidden gradient agrees to bf16
    rounding (same products, GELU' through the hardware reciprocal); the LayerNorm backward sums a token's channels
    in another order."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(C + H + 1)
    dt = torch.bfloat16
    x = (torch.randn(B, H, Wd, C, generator=g) * 1.5 + 0.3).to(dt).cuda()
    dy = torch.randn(B, H, Wd, C, generator=g).to(dt).cuda()
    add = torch.randn(B, H, Wd, C, generator=g).to(dt).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).cuda()
    b1 = (torch.randn(4 * C, generator=g) * 0.1).cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).cuda()
    keep = torch.tensor([1 / 0.9, 0.0, 1 / 0.9, 1 / 0.9][:B]).cuda()
    ln, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    hpre = ops.linear_fwd(ln, ops.PackedConv(w1, dt), bias=b1)
    p2t, p1t = ops.pack_many([(w2, "bn64_t"), (w1, True)], dt)
    assert p2t.bn == 64 and p1t.bn == 96
    later = ops.DeferredSums()
    dhp, dx, dg, db, dxs = ops.swin_mlp_bwd(dy, hpre, p2t, p1t, x, mean, rstd, gamma, add, later, scale_next=(keep, H * Wd))
    later.flush()
    # the separate launches
    dhp0 = ops.linear_fwd(dy, ops.PackedConv(w2, dt, transpose=True), gelu_grad_of=hpre)
    dln0 = ops.linear_fwd(dhp0, ops.PackedConv(w1, dt, transpose=True))
    dx0, dg0, db0 = ops.layernorm_bwd(dln0, x, gamma, mean, rstd, add=add)
    assert rel(dhp, dhp0) < 2e-3
    assert rel(dx, dx0) < 4e-3 and rel(dg, dg0) < 4e-3 and rel(db, db0) < 4e-3
    assert torch.equal(dxs.view(torch.int16), ops.scale_sample(dx, keep).view(torch.int16))
    # without the scaled copy, without the residual gradient
    later = ops.DeferredSums()
    dhp_b, dx_b, _, _, none = ops.swin_mlp_bwd(dy, hpre, p2t, p1t, x, mean, rstd, gamma, None, later)
    later.flush()
    assert none is None and torch.equal(dhp_b.view(torch.int16), dhp.view(torch.int16))
    assert rel(dx_b.float() + add.float(), dx0) < 8e-3
    # torch autograd in fp32 on the bf16-rounded operands
    xr = x.float().cpu().requires_grad_(True)
    gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    h = F.linear(F.layer_norm(xr, (C,), gr, br), w1.to(dt).float().cpu(), b1.cpu())
    out = F.linear(F.gelu(h), w2.to(dt).float().cpu())
    h.retain_grad()
    out.backward(dy.float().cpu())
    assert rel(dhp, h.grad) < 1.5e-2
    assert rel(dx, xr.grad + add.float().cpu()) < 1.5e-2 and rel(dg, gr.grad) < 1.5e-2 and rel(db, br.grad) < 1.5e-2
    with pytest.raises(ValueError):
        ops.swin_mlp_bwd(dy, hpre, p1t, p1t, x, mean, rstd, gamma, add, ops.DeferredSums())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 32, 48, 4), (3, 3, 20, 24, 4), (1, 5, 16, 16, 4), (2, 4, 18, 27, 3),
                                   (2, 4, 16, 18, 4), (1, 1, 8, 8, 2)])
def test_patchify_both_kernels(dt, shape):
    """Patch tokens [c][ky][kx] as the strided patch-embedding conv reads them (swin_unet_model.py:452-453), the
    padding channels zero: 4x4 patches of images whose width is a multiple of four take the 16-byte kernel, the
    others the one-element one."""
    from unet_amd import ops
    B, Cin, H, W, ps = shape
    x = torch.randn(B, Cin, H, W, generator=torch.Generator().manual_seed(5))
    tok = ops.patchify(x.cuda(), ps, dt)
    Hp, Wp, K = H // ps, W // ps, Cin * ps * ps
    ref = x[:, :, :Hp * ps, :Wp * ps].reshape(B, Cin, Hp, ps, Wp, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, Hp, Wp, K)
    assert tok.shape[:3] == (B, Hp, Wp) and tok.shape[3] >= K
    assert torch.equal(tok[..., :K].float().cpu(), ref.to(dt).float())
    assert not tok[..., K:].any()
