"""GPU parity tests, one per kernel family, through the C ABI (ctypes) against torch-CPU fp32.

Tolerances (relative L2 over the whole tensor): float32 kernels 1e-4 (the north-star bound;
most land near 1e-6), bfloat16 kernels 1e-2 for a single op (inputs are rounded to bf16 before
both sides compute, so this measures the kernel, not the input rounding).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-4, torch.bfloat16: 1e-2}


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(shape, seed, dt, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(shape, generator=g) * scale
    return t.to(dt).float()     # values exactly representable in the compute dtype


def nhwc(t, dt):
    """(B,C,H,W) fp32 cpu -> (B,H,W,C) compute dtype on the GPU."""
    return t.permute(0, 2, 3, 1).contiguous().to("cuda", dt)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


CONV_SHAPES = [  # B, H, W, Ci, Co, ks
    (2, 20, 33, 8, 16, 3),
    (1, 16, 16, 48, 96, 3),
    (1, 37, 18, 144, 48, 3),
    (2, 8, 8, 96, 192, 3),
    (2, 64, 48, 192, 96, 3),      # long K
    (1, 40, 33, 288, 192, 3),     # ... ragged in both directions, several channel blocks
    (1, 5, 3, 24, 8, 3),
    (3, 9, 11, 64, 64, 1),
    (1, 40, 40, 96, 384, 1),
    (2, 7, 5, 32, 24, 1),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_fwd_dgrad_wgrad(dt, shape):
    from unet_amd import ops
    B, H, W, Ci, Co, ks = shape
    x = rnd((B, Ci, H, W), 1, dt)
    w = rnd((Co, Ci, ks, ks), 2, dt, (2.0 / (Ci * ks * ks)) ** 0.5)
    gy = rnd((B, Co, H, W), 3, dt)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=ks // 2)
    ref.backward(gy)

    xd = nhwc(x, dt)
    pw = ops.PackedConv(w.cuda(), dt)
    y, stats = ops.conv_fwd(xd, pw, want_stats=True)
    torch.cuda.synchronize()
    assert rel(nchw(y), ref) < TOL[dt]
    # batch statistics from the epilogue
    s = stats.double().sum(0).cpu()
    assert rel(s[0], ref.double().sum((0, 2, 3))) < 1e-3 or ref.double().sum((0, 2, 3)).abs().max() < 1e-2
    assert rel(s[1], (ref.double() ** 2).sum((0, 2, 3))) < (1e-4 if dt == torch.float32 else 1e-2)
    # data gradient = same kernel with transposed, flipped weights
    pwt = ops.PackedConv(w.cuda(), dt, transpose=True)
    gd = nhwc(gy, dt)
    dx, _ = ops.conv_fwd(gd, pwt)
    assert rel(nchw(dx), xr.grad) < TOL[dt]
    # weight gradient
    dw = ops.conv_wgrad(xd, gd, Ci, Co, ks)
    assert rel(dw.cpu(), wr.grad) < TOL[dt]
    # accumulate flag
    dw2 = ops.conv_wgrad(xd, gd, Ci, Co, ks, dw=dw.clone(), accumulate=True)
    assert rel(dw2.cpu(), 2 * wr.grad) < TOL[dt]


@pytest.mark.parametrize("shape", [(1, 128, 256, 48, 0, 48), (2, 66, 250, 48, 0, 48), (1, 132, 256, 144, 0, 48),
                                   (1, 128, 256, 96, 48, 48), (4, 47, 161, 96, 48, 48), (1, 128, 256, 96, 0, 96),
                                   (2, 66, 250, 48, 0, 96), (1, 132, 256, 192, 96, 96), (1, 128, 256, 288, 0, 96),
                                   (4, 47, 161, 192, 96, 96)])
def test_wgrad_whole_pixel_kernel(shape):
    """The weight gradient of the full- and half-resolution layers (48 / 96 output channels, dense bf16 operands staged
    as whole pixels, no k-split) against autograd and against the generic kernel; ragged sizes, one and two operands,
    accumulate."""
    import os
    from unet_amd import ops
    dt = torch.bfloat16
    B, H, W, C1, C2, Co = shape
    Ci = C1 + C2
    x = rnd((B, Ci, H, W), 61, dt)
    gy = rnd((B, Co, H, W), 62, dt)
    wr = torch.zeros((Co, Ci, 3, 3), requires_grad=True)
    F.conv2d(x, wr, None, padding=1).backward(gy)
    xd, gd = nhwc(x, dt), nhwc(gy, dt)
    arg = (xd[..., :C1].contiguous(), xd[..., C1:].contiguous()) if C2 else xd
    dw = ops.conv_wgrad(arg, gd, Ci, Co, 3)
    assert rel(dw.cpu(), wr.grad) < TOL[dt]
    os.environ["UNETK_WGRAD_NO_PX"] = "1"
    try:
        generic = ops.conv_wgrad(arg, gd, Ci, Co, 3)
    finally:
        del os.environ["UNETK_WGRAD_NO_PX"]
    assert not torch.equal(dw, generic) and rel(dw.cpu(), generic.cpu()) < 1e-5      # another summation order
    assert torch.equal(dw, ops.conv_wgrad(arg, gd, Ci, Co, 3))                         # and the same bits every run
    dw2 = ops.conv_wgrad(arg, gd, Ci, Co, 3, dw=dw.clone(), accumulate=True)
    assert rel(dw2.cpu(), 2 * wr.grad) < TOL[dt]
    if C2:   # the stored concatenation takes the same kernel with the same unit assignment: the same bits
        assert torch.equal(dw, ops.conv_wgrad(xd, gd, Ci, Co, 3))


@pytest.mark.parametrize("shape", [(16, 128, 128, 64, 64, 3), (16, 128, 128, 64, 32, 3), (12, 100, 130, 40, 24, 3),
                                   (16, 128, 128, 40, 24, 3), (16, 128, 128, 64, 32, 1)])
def test_wgrad_many_slab_reduction(shape):
    """A narrow layer at full resolution: a small dw under many pixel splits, summed by the slab-lane reduction
    (from 32 slabs on) instead of the slab-after-slab one.  Against autograd, the same bits every run, accumulate."""
    from unet_amd import ops
    dt = torch.bfloat16
    B, H, W, Ci, Co, ks = shape
    if shape[:5] == (16, 128, 128, 64, 64):     # this one is known to take the many-slab path: 64x9x64 floats a slab
        ws = ops._lib.load().unetk_conv_wgrad_workspace_bytes(B, H, W, Ci, Co, ks, 1)
        assert ws // (64 * 9 * 64 * 4) >= 32
    x = rnd((B, Ci, H, W), 71, dt)
    gy = rnd((B, Co, H, W), 72, dt)
    wr = torch.zeros((Co, Ci, ks, ks), requires_grad=True)
    F.conv2d(x, wr, None, padding=ks // 2).backward(gy)
    xd, gd = nhwc(x, dt), nhwc(gy, dt)
    dw = ops.conv_wgrad(xd, gd, Ci, Co, ks)
    assert rel(dw.cpu(), wr.grad) < TOL[dt]
    assert torch.equal(dw, ops.conv_wgrad(xd, gd, Ci, Co, ks))
    dw2 = ops.conv_wgrad(xd, gd, Ci, Co, ks, dw=dw.clone(), accumulate=True)
    assert rel(dw2.cpu(), 2 * wr.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(144, 96), (144, 48), (192, 96), (128, 64)])
def test_conv_split_outputs(dt, shape):
    """One conv, two dense results: channels [0,split) and [split,Co) (the data gradient of a
    concat-consuming conv leaves as the gradients of its two inputs), odd image sizes."""
    from unet_amd import ops
    Co, split = shape
    B, H, W, Ci = 2, 21, 19, 16
    x = rnd((B, Ci, H, W), 31, dt)
    w = rnd((Co, Ci, 3, 3), 32, dt, 0.1)
    ref = F.conv2d(x, w, None, padding=1)
    pw = ops.PackedConv(w.cuda(), dt)
    assert split % pw.bn == 0
    (ya, yb), _ = ops.conv_fwd(nhwc(x, dt), pw, split=split)
    assert ya.shape[-1] == split and yb.shape[-1] == Co - split
    whole, _ = ops.conv_fwd(nhwc(x, dt), pw)
    assert torch.equal(torch.cat((ya, yb), -1), whole)
    assert rel(nchw(whole), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_epilogue_affine_relu_residual(dt):
    from unet_amd import ops
    B, H, W, Ci, Co = 2, 19, 21, 16, 32
    x = rnd((B, Ci, H, W), 4, dt)
    w = rnd((Co, Ci, 3, 3), 5, dt, 0.1)
    sc = torch.rand(Co) + 0.5
    sh = torch.randn(Co) * 0.3
    ref = F.relu(F.conv2d(x, w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    pw = ops.PackedConv(w.cuda(), dt)
    y, _ = ops.conv_fwd(nhwc(x, dt), pw, scale=sc.cuda(), shift=sh.cuda(), act=1)
    assert rel(nchw(y), ref) < TOL[dt]
    # residual with per-sample keep scale and GELU, 1x1
    w1 = rnd((Co, Ci, 1, 1), 6, dt, 0.2)
    res = rnd((B, Co, H, W), 7, dt)
    keep = torch.tensor([0.0, 1.25])
    ref2 = res + keep.view(B, 1, 1, 1) * (F.conv2d(x, w1) + sh.view(1, -1, 1, 1))
    pw1 = ops.PackedConv(w1.cuda(), dt)
    y2, _ = ops.conv_fwd(nhwc(x, dt), pw1, shift=sh.cuda(), resid=nhwc(res, dt), resid_keep=keep.cuda())
    assert rel(nchw(y2), ref2) < TOL[dt]
    ref3 = F.gelu(F.conv2d(x, w1) + sh.view(1, -1, 1, 1))
    y3, _ = ops.conv_fwd(nhwc(x, dt), pw1, shift=sh.cuda(), act=2)
    assert rel(nchw(y3), ref3) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (3, 9, 7, 48), (1, 32, 30, 96), (2, 5, 4, 768)])
def test_bn_relu_dropout_pool_fwd_bwd(dt, shape):
    from unet_amd import ops
    B, H, W, C = shape
    y = (rnd((B, C, H, W), 8, dt) * 1.5 + 0.3).to(dt).float()
    gamma = torch.rand(C) + 0.5
    beta = torch.randn(C) * 0.2
    keep = (torch.rand(B, C) > 0.3).float() / 0.7
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    # reference
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    a_ref = F.relu(z) * keep.view(B, C, 1, 1)
    p_ref = F.max_pool2d(a_ref, 2, 2) if H >= 2 and W >= 2 else None
    # kernels: statistics from a 1x1 identity-free path -> use conv epilogue substitute: compute partials by torch
    yd = nhwc(y, dt)
    stats = torch.stack([yd.float().sum((0, 1, 2)), (yd.float() ** 2).sum((0, 1, 2))]).view(1, 2, C)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    mean, invstd, sc, sh = ops.bn_finalize(stats.contiguous(), B * H * W, gamma.cuda(), beta.cuda(), rm_d, rv_d)
    assert rel(rm_d, rm_ref) < 1e-5 and rel(rv_d, rv_ref) < 1e-5
    a, pooled = ops.bn_act_fwd(yd, sc, sh, keep.cuda(), relu=True, pool=p_ref is not None)
    assert rel(nchw(a), a_ref) < TOL[dt]
    if p_ref is not None:
        assert rel(nchw(pooled), p_ref) < TOL[dt]
        assert torch.equal(ops.maxpool2x2(a), pooled)
    # backward
    g = rnd((B, C, H, W), 9, dt)
    a_ref.backward(g)
    dy, dgamma, dbeta = ops.bn_act_bwd(nhwc(g, dt), yd, sc, sh, mean, invstd, keep.cuda(), relu=True)
    tol = TOL[dt] * (3 if dt == torch.bfloat16 else 1)
    assert rel(nchw(dy), yr.grad) < tol
    assert rel(dgamma, gr.grad) < tol and rel(dbeta, br.grad) < tol


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (1, 9, 7, 48), (2, 2, 2, 8)])
def test_pool_bwd_first_max_and_skip(dt, shape):
    from unet_amd import ops
    B, H, W, C = shape
    a = F.relu(rnd((B, C, H, W), 10, dt))          # many exact ties at 0
    ar = a.clone().requires_grad_(True)
    p = F.max_pool2d(ar, 2, 2)
    gp = rnd(tuple(p.shape), 11, dt)
    p.backward(gp)
    skip = rnd((B, C + 8, H, W), 12, dt)            # skip gradient lives in channels [8:] of a wider buffer
    sk = nhwc(skip, dt)
    da = ops.pool_bwd(nhwc(a, dt), nhwc(gp, dt), sk[..., 8:])
    # the same through the forward's argmax map, the activation written into a channel slice
    wide = torch.zeros((B, H, W, C + 8), dtype=dt, device="cuda")
    a2, pooled2, amap = ops.bn_act_fwd(nhwc(a, dt), None, None, None, relu=False, pool=True, out=wide[..., 8:],
                                       want_argmax=True)
    assert torch.equal(wide[..., 8:], nhwc(a, dt)) and torch.equal(wide[..., :8], torch.zeros_like(wide[..., :8]))
    assert torch.equal(ops.pool_bwd(None, nhwc(gp, dt), sk[..., 8:], argmax=amap), da)
    want = ar.grad + skip[:, 8:]
    assert rel(nchw(da), want) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (3, 9, 7, 48), (2, 2, 2, 8), (1, 32, 30, 96), (2, 5, 4, 768),
                                   (3, 70, 50, 48)])
def test_bn_bwd_with_the_pool_gradient_formed_inside(dt, shape):
    """BatchNorm backward whose incoming gradient is pool_bwd(dpooled, dskip, argmax) formed inside both passes
    (unetk_bn_act_bwd_pool): the same bits as the stored tensor gives, odd sizes and a strided skip gradient
    included."""
    from unet_amd import ops
    B, H, W, C = shape
    y = nhwc((rnd((B, C, H, W), 40, dt) * 1.5 + 0.3), dt)
    stats = torch.stack([y.float().sum((0, 1, 2)), (y.float() ** 2).sum((0, 1, 2))]).view(1, 2, C).contiguous()
    gamma, beta = (torch.rand(C) + 0.5).cuda(), (torch.randn(C) * 0.2).cuda()
    mean, invstd, sc, sh = ops.bn_finalize(stats, B * H * W, gamma, beta, None, None)
    keep = ((torch.rand(B, C) > 0.3).float() / 0.7).cuda()
    a, pooled, amap = ops.bn_act_fwd(y, sc, sh, keep, relu=True, pool=True, want_argmax=True)
    gp = nhwc(rnd((B, C, H // 2, W // 2), 41, dt), dt)
    wide = nhwc(rnd((B, C + 8, H, W), 42, dt), dt)
    for dskip in (wide[..., 8:], wide[..., 8:].contiguous()):
        g = ops.pool_bwd(None, gp, dskip, argmax=amap)
        dy0, dg0, db0 = ops.bn_act_bwd(g, y, sc, sh, mean, invstd, keep, relu=True)
        dy1, dg1, db1 = ops.bn_act_bwd(ops.PooledGrad(gp, dskip, amap), y, sc, sh, mean, invstd, keep, relu=True)
        assert torch.equal(dy0, dy1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


@pytest.mark.parametrize("shape", [(2, 32, 32, 96, 96), (3, 19, 23, 64, 128), (1, 16, 16, 192, 96), (2, 7, 40, 96, 64),
                                   (5, 33, 17, 288, 192), (2, 20, 21, 72, 40), (1, 5, 3, 136, 96)])
def test_bn_bwd_sums_from_the_data_gradient_conv(shape):
    """The data-gradient conv that leaves BatchNorm-backward's two sums beside its output
    (unetk_conv_bwd_data_bnsums + unetk_bn_act_bwd_presummed): the gradient itself has the same bits as the plain
    launch, the sums agree with the separate reduction pass (another summation order: 1e-5 of their scale), and so
    does what the BatchNorm backward makes of them.  Ragged image sizes and several channel blocks included."""
    from unet_amd import ops
    dt = torch.bfloat16
    B, H, W, C, C2 = shape              # the block's first conv has C outputs, its second conv C -> C2
    y = nhwc((rnd((B, C, H, W), 50, dt) * 1.5 + 0.3), dt)
    stats = torch.stack([y.float().sum((0, 1, 2)), (y.float() ** 2).sum((0, 1, 2))]).view(1, 2, C).contiguous()
    gamma, beta = (torch.rand(C) + 0.5).cuda(), (torch.randn(C) * 0.2).cuda()
    mean, invstd, sc, sh = ops.bn_finalize(stats, B * H * W, gamma, beta, None, None)
    w = torch.randn(C2, C, 3, 3, generator=torch.Generator().manual_seed(51)) * 0.1
    pwt = ops.PackedConv(w.cuda(), dt, transpose=True)
    assert ops.conv_bwd_data_bnsums_supported(pwt, dt)
    dy2 = nhwc(rnd((B, C2, H, W), 52, dt), dt)
    g0, _ = ops.conv_fwd(dy2, pwt)
    g1, parts = ops.conv_bwd_data_bnsums(dy2, pwt, y, sc, sh, mean, invstd)
    assert torch.equal(g0, g1)
    dz = torch.where(torch.addcmul(sh, y.float(), sc) > 0, g0.float(), torch.zeros((), device="cuda")).double()
    want0 = dz.sum((0, 1, 2))
    want1 = (dz * ((y.float() - mean) * invstd).double()).sum((0, 1, 2))
    got = parts.double().sum(0)
    scale0 = dz.abs().sum((0, 1, 2)).max().item()
    assert (got[0] - want0).abs().max().item() < 1e-5 * scale0
    assert (got[1] - want1).abs().max().item() < 2e-5 * scale0
    dya, dga, dba = ops.bn_act_bwd(g0, y, sc, sh, mean, invstd, None, relu=True)
    dyb, dgb, dbb = ops.bn_act_bwd(g1, y, sc, sh, mean, invstd, None, relu=True, presummed=parts)
    assert (dga - dgb).abs().max().item() < 2e-5 * scale0 and (dba - dbb).abs().max().item() < 1e-5 * scale0
    # the apply pass rounds to bfloat16: a last-place difference in a coefficient may flip a rounding here and there
    diff = (dya.float() - dyb.float()).abs()
    assert diff.max().item() <= 2 ** -7 * dya.float().abs().max().item()
    assert (diff > 0).float().mean().item() < 1e-2
    with pytest.raises(ValueError):
        ops.bn_act_bwd(g1, y, sc, sh, mean, invstd, torch.ones(B, C, device="cuda"), relu=True, presummed=parts)


def test_bn_bwd_sums_entry_point_refuses_what_it_does_not_cover():
    """float32, 32- / 48-channel blocks and a y of another shape are refused with an error, not computed wrongly."""
    from unet_amd import ops
    w = torch.randn(96, 96, 3, 3, generator=torch.Generator().manual_seed(1)) * 0.1
    assert not ops.conv_bwd_data_bnsums_supported(ops.PackedConv(w.cuda(), torch.float32, transpose=True), torch.float32)
    w48 = torch.randn(64, 48, 3, 3, generator=torch.Generator().manual_seed(2)) * 0.1
    p48 = ops.PackedConv(w48.cuda(), torch.bfloat16, transpose=True)       # 48 outputs: the 48-channel block
    assert not ops.conv_bwd_data_bnsums_supported(p48, torch.bfloat16)
    dt = torch.bfloat16
    dy = torch.zeros(1, 8, 8, 64, dtype=dt, device="cuda")
    y = torch.zeros(1, 8, 8, 48, dtype=dt, device="cuda")
    v = torch.ones(48, device="cuda")
    with pytest.raises(ValueError, match="channel block"):
        ops.conv_bwd_data_bnsums(dy, p48, y, v, v, v, v)
    pw = ops.PackedConv(w.cuda(), dt, transpose=True)
    with pytest.raises(ValueError):
        ops.conv_bwd_data_bnsums(torch.zeros(1, 8, 8, 96, dtype=dt, device="cuda"), pw,
                                 torch.zeros(1, 8, 4, 96, dtype=dt, device="cuda"), v, v, v, v)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hw", [((8, 8), (16, 16)), ((6, 9), (12, 18)), ((12, 18), (25, 36)), ((1, 1), (3, 3)),
                                ((3, 5), (7, 11)), ((1, 1), (2, 2)), ((2, 3), (4, 6)), ((1, 5), (2, 10)),
                                ((5, 1), (10, 2)), ((17, 9), (34, 18))])
def test_upsample_concat_fwd_bwd(dt, hw):
    from unet_amd import ops
    (h, w), (H, W) = hw
    B, C1, C2 = 2, 16, 8
    low = rnd((B, C1, h, w), 13, dt)
    skip = rnd((B, C2, H, W), 14, dt)
    lr = low.clone().requires_grad_(True)
    ref = torch.cat((F.interpolate(lr, size=(H, W), mode="bilinear", align_corners=False), skip), 1)
    g = rnd(tuple(ref.shape), 15, dt)
    ref.backward(g)
    out = ops.upsample_concat(nhwc(low, dt), nhwc(skip, dt))
    assert rel(nchw(out), ref) < TOL[dt]
    if dt == torch.float32:          # pointwise, so one wrong border pixel cannot hide in a norm
        assert (nchw(out) - ref).abs().max().item() < 1e-5
    cat = torch.zeros(B, H, W, C1 + C2, device="cuda", dtype=dt)
    ops.upsample_into(nhwc(low, dt), cat)
    assert torch.equal(cat[..., :C1], out[..., :C1]) and not cat[..., C1:].any()
    dlow = ops.upsample_bwd(nhwc(g, dt), C1, h, w)
    assert rel(nchw(dlow), lr.grad) < TOL[dt]


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 16, 24, 96), (3, 9, 13, 64), (2, 32, 32, 192), (1, 40, 8, 32)])
def test_upsample_bwd_exact_2x_tiled(shape):
    """The LDS-tiled backward of the exact 2x bilinear resize (bf16, channel slabs of 32): the same bits as the generic
    gather kernel, and autograd of F.interpolate to bf16 accuracy; ragged tiles, wide strided gradients."""
    import os
    from unet_amd import ops
    dt = torch.bfloat16
    B, h, w, C1 = shape
    lr = rnd((B, C1, h, w), 71, dt).requires_grad_(True)
    g = rnd((B, C1 + 16, 2 * h, 2 * w), 72, dt)
    F.interpolate(lr, size=(2 * h, 2 * w), mode="bilinear", align_corners=False).backward(g[:, :C1])
    gd = nhwc(g, dt)                         # channels [0, C1) of a wider tensor: ldout > C1
    dlow = ops.upsample_bwd(gd, C1, h, w)
    os.environ["UNETK_UPSAMPLE_BWD_GENERIC"] = "1"
    try:
        generic = ops.upsample_bwd(gd, C1, h, w)
    finally:
        del os.environ["UNETK_UPSAMPLE_BWD_GENERIC"]
    assert torch.equal(dlow, generic)
    assert rel(nchw(dlow), lr.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Co", [1, 3])
def test_head1x1_fwd_bwd(dt, Co):
    from unet_amd import ops
    B, H, W, C = 2, 13, 17, 48
    x = rnd((B, C, H, W), 16, dt)
    w = torch.randn(Co, C, 1, 1) * 0.2
    b = torch.randn(Co)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br)
    g = torch.randn(B, Co, H, W)
    ref.backward(g)
    xd = nhwc(x, dt)
    out = ops.head1x1_fwd(xd, w.cuda(), b.cuda())
    assert rel(out, ref) < 1e-5
    dx, dw, db = ops.head1x1_bwd(xd, w.cuda(), g.cuda())
    assert rel(nchw(dx), xr.grad) < TOL[dt]
    assert rel(dw, wr.grad) < 1e-4 and rel(db, br.grad) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [1, 4, 5, 11])      # one 16-byte vector per pixel, and wider
def test_layout_roundtrip(dt, C):
    from unet_amd import ops
    x = rnd((3, C, 7, 9), 17, dt)
    h = ops.nchw_to_nhwc(x.cuda(), dt)
    assert h.shape[-1] % (8 if dt == torch.bfloat16 else 4) == 0
    assert torch.equal(h[..., :C].float().cpu(), x.permute(0, 2, 3, 1))
    assert torch.count_nonzero(h[..., C:]) == 0
    assert torch.equal(ops.nhwc_to_nchw(h, C).cpu(), x)


def test_bad_arguments_raise():
    from unet_amd import ops
    x = torch.zeros((1, 4, 4, 6), device="cuda", dtype=torch.bfloat16)   # 6 channels: not a multiple of 8
    pw = ops.PackedConv(torch.zeros(8, 8, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(ValueError):
        ops.conv_fwd(x, pw, Ci=6, Co=8)
    with pytest.raises(RuntimeError):
        ops.nchw_to_nhwc(torch.zeros(1, 1, 2, 2), torch.float32)   # CPU tensor: no fallback


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 20, 33, 16, 48), (1, 32, 32, 96, 96), (2, 9, 7, 8, 16)])
def test_conv_fused_pool_and_head(dt, shape):
    """Inference-time epilogue fusions: bit-identical max-pool, and the final 1x1 conv."""
    from unet_amd import ops
    B, H, W, Ci, Co = shape
    x = nhwc(rnd((B, Ci, H, W), 21, dt), dt)
    w = rnd((Co, Ci, 3, 3), 22, dt, (2.0 / (Ci * 9)) ** 0.5)
    sc, sh = (torch.rand(Co) + 0.5).cuda(), (torch.randn(Co) * 0.3).cuda()
    pw = ops.PackedConv(w.cuda(), dt)
    y, _ = ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1)
    wide = torch.zeros((B, H, W, Co + 8), dtype=dt, device="cuda")
    y2, pooled, none = ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1, out=wide[..., 8:], pool=True)
    assert none is None and torch.equal(y2, y) and torch.equal(pooled, ops.maxpool2x2(y))
    if Co <= pw.bn:
        hw, hb = torch.randn(2, Co, 1, 1).cuda() * 0.2, torch.randn(2).cuda()
        o, p, head = ops.conv_fwd(x, pw, scale=sc, shift=sh, act=1, head=(hw, hb), write_out=False)
        assert o is None and p is None
        want = ops.head1x1_fwd(y, hw, hb)
        assert rel(head.cpu(), want.cpu()) < 1e-5


@pytest.mark.parametrize("ntok,Ci,Co", [(64 * 64 * 4, 96, 288), (4096, 384, 96), (1000, 192, 192), (8 * 8 * 3, 768, 3072),
                                        (256, 32, 48), (5000, 96, 96), (4133, 96, 384), (9001, 384, 96), (4096, 192, 576),
                                        (4100, 192, 192), (3001, 192, 768), (2100, 768, 192), (32 * 32 * 8, 192, 768)])
def test_linear_wgrad_with_bias(ntok, Ci, Co):
    """nn.Linear backward over tokens: dw = dy^T x and db = sum dy from one pass (bf16), against fp64 sums of the
    same bf16-rounded operands (the narrow shapes over at least 2048-4096 tokens take the whole-row kernel: ragged
    last tiles, every column split)."""
    from unet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(ntok + Ci)
    x = torch.randn((ntok, Ci), device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn((ntok, Co), device="cuda", generator=g).to(torch.bfloat16)
    dw, db = ops.linear_wgrad(x, dy, Ci, Co)
    want_w = dy.double().t() @ x.double()
    want_b = dy.double().sum(0)
    assert dw.shape == (Co, Ci) and db.shape == (Co,)
    assert float((dw.double() - want_w).norm() / want_w.norm()) < 1e-5
    assert float((db.double() - want_b).norm() / want_b.norm()) < 1e-5
    dw2, db2 = ops.linear_wgrad(x, dy, Ci, Co)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed summation order


@pytest.mark.parametrize("shape", [(2, 37, 18, 48, 48), (1, 70, 50, 144, 48), (3, 33, 16, 48, 144), (1, 32, 16, 40, 40),
                                   (2, 5, 3, 8, 48), (1, 96, 64, 48, 48), (5, 64, 64, 96, 48), (40, 64, 64, 48, 48),
                                   (3, 100, 40, 48, 40), (1, 2, 3, 48, 48), (6, 70, 50, 48, 144), (36, 64, 64, 40, 144)])
def test_conv_48_channel_blocks_bf16(shape):
    """48-channel output blocks (bf16, 3x3), among them the persistent whole-pixel kernel that takes the layers with
    33..48 input and up to 48 output channels: ragged images, partial channel counts, one and many tiles per
    workgroup; batch statistics, folded scale/shift + ReLU, split outputs, and the fused pool / 1x1 head epilogues."""
    from unet_amd import ops
    dt = torch.bfloat16
    B, H, W, Ci, Co = shape
    x = rnd((B, Ci, H, W), 41, dt)
    w = rnd((Co, Ci, 3, 3), 42, dt, (2.0 / (Ci * 9)) ** 0.5)
    ref = F.conv2d(x, w, None, padding=1)
    xd = nhwc(x, dt)
    pw = ops.PackedConv(w.cuda(), dt)
    assert pw.bn == 48
    y, stats = ops.conv_fwd(xd, pw, want_stats=True)
    assert rel(nchw(y), ref) < TOL[dt]
    s = stats.double().sum(0).cpu()
    assert rel(s[0], ref.double().sum((0, 2, 3))) < 2e-3 or ref.double().sum((0, 2, 3)).abs().max() < 1e-2
    assert rel(s[1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-2
    y_again, stats_again = ops.conv_fwd(xd, pw, want_stats=True)
    assert torch.equal(y, y_again) and torch.equal(stats, stats_again)
    # input that is a channel slice of a wider buffer, output into a channel slice
    wide_in = torch.randn((B, H, W, Ci + 16), device="cuda").to(dt)
    wide_in[..., 8:8 + Ci] = xd
    wide_out = torch.zeros((B, H, W, Co + 8), dtype=dt, device="cuda")
    y3, _ = ops.conv_fwd(wide_in[..., 8:8 + Ci], pw, out=wide_out[..., 8:])
    y_plain, _ = ops.conv_fwd(xd, pw)        # (with and without statistics different kernels may run: same sums, other order)
    assert rel(nchw(y_plain), ref) < TOL[dt] and rel(y_plain.float().cpu(), y.float().cpu()) < 2e-3
    assert torch.equal(y3, y_plain) and float(wide_out[..., :8].abs().max()) == 0.0
    sc, sh = (torch.rand(Co) + 0.5), (torch.randn(Co) * 0.3)
    ya, _ = ops.conv_fwd(xd, pw, scale=sc.cuda(), shift=sh.cuda(), act=1)
    assert rel(nchw(ya), F.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))) < TOL[dt]
    if Co == 144:
        for split in (48, 96):
            (p, r), _ = ops.conv_fwd(xd, pw, split=split)
            assert torch.equal(torch.cat((p, r), -1), y_plain)
    y2, pooled, _ = ops.conv_fwd(xd, pw, scale=sc.cuda(), shift=sh.cuda(), act=1, pool=True)
    assert rel(y2.float().cpu(), ya.float().cpu()) < 2e-3 and torch.equal(pooled, ops.maxpool2x2(y2))
    if Co <= 48:
        assert torch.equal(y2, ya)
        hw, hb = torch.randn(2, Co, 1, 1).cuda() * 0.2, torch.randn(2).cuda()
        o, p, head = ops.conv_fwd(xd, pw, scale=sc.cuda(), shift=sh.cuda(), act=1, head=(hw, hb), write_out=False)
        assert o is None and p is None
        assert rel(head.cpu(), ops.head1x1_fwd(ya, hw, hb).cpu()) < 1e-5
    # data gradient through the same kernel (transposed, flipped pack)
    gy = rnd((B, Co, H, W), 43, dt)
    xr, wr = x.clone().requires_grad_(True), w.clone()
    F.conv2d(xr, wr, None, padding=1).backward(gy)
    pwt = ops.PackedConv(w.cuda(), dt, transpose=True)
    dx, _ = ops.conv_fwd(nhwc(gy, dt), pwt)
    assert rel(nchw(dx), xr.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 20, 33, 32, 16, 64), (1, 37, 18, 96, 48, 48), (2, 16, 16, 64, 8, 96),
                                   (1, 40, 24, 192, 96, 96), (1, 9, 70, 32, 40, 32)])
def test_conv_of_a_concatenation_read_in_place(dt, shape):
    """conv3x3(cat([x1, x2])) and its weight gradient with the two tensors read as two operands (wide buffers'
    channel slices included): the same bits as on the stored concatenation."""
    from unet_amd import ops
    B, H, W, C1, C2, Co = shape
    x1 = nhwc(rnd((B, C1, H, W), 51, dt), dt)
    wide = torch.randn((B, H, W, C2 + 24), device="cuda").to(dt)
    x2 = wide[..., 8:8 + C2]
    cat = torch.cat((x1, x2), -1).contiguous()
    w = rnd((Co, C1 + C2, 3, 3), 52, dt, (2.0 / ((C1 + C2) * 9)) ** 0.5)
    pw = ops.PackedConv(w.cuda(), dt)
    want, want_stats = ops.conv_fwd(cat, pw, want_stats=True)
    got, got_stats = ops.conv_fwd_cat(x1, x2, pw, want_stats=True)
    assert torch.equal(got, want) and torch.equal(got_stats, want_stats)
    ref = F.conv2d(nchw(cat), w, None, padding=1)
    assert rel(nchw(got), ref) < TOL[dt]
    sc, sh = (torch.rand(Co) + 0.5).cuda(), (torch.randn(Co) * 0.3).cuda()
    assert torch.equal(ops.conv_fwd_cat(x1, x2, pw, scale=sc, shift=sh, act=1)[0],
                       ops.conv_fwd(cat, pw, scale=sc, shift=sh, act=1)[0])
    dy = nhwc(rnd((B, Co, H, W), 53, dt), dt)
    assert torch.equal(ops.conv_wgrad((x1, x2), dy, C1 + C2, Co, 3), ops.conv_wgrad(cat, dy, C1 + C2, Co, 3))
    with pytest.raises(ValueError):
        ops.conv_fwd_cat(x1[..., :C1 - 8], torch.cat((x1[..., C1 - 8:], x2), -1).contiguous(), pw)   # not whole chunks
    up = ops.upsample(x1[:, :H // 2 or 1, :W // 2 or 1].contiguous(), H, W)
    both = torch.zeros((B, H, W, C1 + 8), dtype=dt, device="cuda")
    ops.upsample_into(x1[:, :H // 2 or 1, :W // 2 or 1].contiguous(), both)
    assert torch.equal(up, both[..., :C1])


@pytest.mark.parametrize("dt", DTYPES)
def test_batched_weight_packing_matches_single_packs(dt):
    """One launch packs many weights group by group through LDS; every pack must hold the same bits as the
    one-weight kernel's (forward and data-gradient form, 1x1 and 3x3, ragged channel counts, every block width)."""
    from unet_amd import ops
    shapes = [(48, 8, 3), (48, 48, 3), (96, 144, 3), (100, 40, 3), (24, 200, 3), (384, 1152, 3), (1, 48, 1), (3, 7, 3),
              (288, 96, 1), (96, 384, 1), (50, 70, 1), (768, 192, 1), (64, 64, 3), (33, 31, 1)]
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(co, ci, ks, ks, generator=g).cuda() for co, ci, ks in shapes]
    ws.append(torch.randn(192, 96, generator=g).cuda())               # a Linear weight (two dimensions)
    jobs = [(w, tr) for tr in (False, True) for w in ws]
    many = ops.pack_many(jobs, dt)
    again = ops.pack_many(jobs, dt)                                     # the cached job table
    for (w, tr), pk, pk2 in zip(jobs, many, again):
        one = ops.PackedConv(w if w.dim() == 4 else w.view(*w.shape, 1, 1), dt, transpose=tr)
        assert (pk.bn, pk.Co, pk.Ci, pk.ks) == (one.bn, one.Co, one.Ci, one.ks)
        assert torch.equal(pk.data.view(torch.uint8), one.data.view(torch.uint8)), (tuple(w.shape), tr)
        assert torch.equal(pk2.data.view(torch.uint8), one.data.view(torch.uint8)), (tuple(w.shape), tr)


@pytest.mark.parametrize("dt", DTYPES)
def test_batched_packing_of_conv_transpose_weights(dt):
    """A ConvTranspose2d(k=2, s=2) weight packed by the batched launch as its (4*up, Cin) GEMM matrix, forward and
    data-gradient form, against the pack of the permuted copy: the same bits."""
    from unet_amd import ops
    g = torch.Generator().manual_seed(9)
    for Cin, up in ((768, 384), (128, 64), (40, 24), (96, 8)):
        w = torch.randn(Cin, up, 2, 2, generator=g).cuda()
        lin = torch.randn(up, Cin, generator=g).cuda()                   # an ordinary weight in the same launch
        w2 = w.permute(2, 3, 1, 0).reshape(4 * up, Cin).contiguous()
        got = ops.pack_many([(lin, False), (w, "convt"), (w, "convt_t"), (lin, True)], dt)
        want = [ops.PackedConv(lin, dt), ops.PackedConv(w2, dt), ops.PackedConv(w2, dt, transpose=True),
                ops.PackedConv(lin, dt, transpose=True)]
        for a, b in zip(got, want):
            assert (a.bn, a.Co, a.Ci, a.ks) == (b.bn, b.Co, b.Ci, b.ks)
            assert torch.equal(a.data.view(torch.uint8), b.data.view(torch.uint8)), (Cin, up)
