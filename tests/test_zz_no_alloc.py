"""GPU test of the allocation-free loss entry point, unetk_loss_pixgrad_fwd_static.

Called through the C ABI (ctypes) and compared with unetk_loss_pixgrad_fwd (caller-owned partial
sums) on the same inputs: the two run the same kernels on the same number of workgroups, so the
16-float result record and the Sobel maps must be bit-identical (the loss arithmetic is fp32; the
bar against the oracle is 1e-4 relative and is checked too).  Two calls in a row, the second with a
smaller input, so partial sums left over from the first call must not leak into the second.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import loss_oracle
from oracle import weights as W

pytestmark = pytest.mark.gpu


def _run_pair(lib, p, t, w, method, mae, use_w, scale):
    from unet_amd._lib import ptr, stream
    B, _, H, Wd = p.shape
    dev = p.device
    out = []
    for static in (False, True):
        res = torch.full((16,), -7.0, dtype=torch.float32, device=dev)      # [2], [3] are not this call's to write
        gx = torch.zeros((B, H, Wd), dtype=torch.float32, device=dev)
        gy = torch.zeros((B, H, Wd), dtype=torch.float32, device=dev)
        if static:
            rc = lib.unetk_loss_pixgrad_fwd_static(ptr(p), ptr(t), ptr(w), B, H, Wd, mae, method, use_w, scale,
                                                   ptr(gx), ptr(gy), ptr(res), stream())
        else:
            partials = torch.empty((lib.unetk_loss_num_partials(B, H, Wd), 11), dtype=torch.float32, device=dev)
            rc = lib.unetk_loss_pixgrad_fwd(ptr(p), ptr(t), ptr(w), B, H, Wd, mae, method, use_w, scale,
                                            ptr(gx), ptr(gy), ptr(partials), ptr(res), stream())
        assert rc == 0, lib.unetk_last_error()
        torch.cuda.synchronize()
        out.append((res.cpu().numpy(), gx.cpu().numpy(), gy.cpu().numpy()))
    return out


def test_static_entry_point_matches_the_workspace_one_twice_in_a_row():
    from unet_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    # first call: 8 x 192 x 160 (many workgroups); second: 2 x 24 x 40 (one workgroup)
    for (B, H, Wd), seed in (((8, 192, 160), 5), ((2, 24, 40), 6)):
        x, y, w = W.synthetic_batch(B, 1, H, Wd, seed=seed)
        p = (0.7 * y + 0.3 * x).contiguous()
        pd, yd, wd = p.cuda(), y.cuda(), w.cuda()
        for method, mae, use_w in ((1, 0, 1), (2, 1, 1), (0, 0, 0)):
            (r0, gx0, gy0), (r1, gx1, gy1) = _run_pair(lib, pd, yd, wd, method, mae, use_w, 1.2)
            assert np.array_equal(r0[:12], r1[:12]), (B, H, Wd, method, r0, r1)
            assert np.array_equal(gx0, gx1) and np.array_equal(gy0, gy1)
            # and against the oracle (fp32 bound 1e-4)
            ref_pix = loss_oracle.pixel_loss(p, y, w, "mae" if mae else "mse", bool(use_w), 1.2).item()
            assert abs(r1[0] - ref_pix) <= 1e-4 * abs(ref_pix)
            if method:
                ref_g = loss_oracle.gradient_loss(p, y, w, "sobel" if method == 1 else "diff").item()
                assert abs(r1[1] - ref_g) <= 1e-4 * abs(ref_g)


def test_static_entry_point_rejects_oversized_and_null_inputs():
    from unet_amd import _lib
    from unet_amd._lib import ptr, stream
    lib = _lib.load()
    torch.cuda.set_device(0)
    t = torch.zeros(16, device="cuda")
    assert lib.unetk_loss_static_max_pixels() == 1 << 30
    # 2^15 x 2^8 x 2^8 pixels = 2^31 > 2^30: refused before any launch, the pointers are never read
    rc = lib.unetk_loss_pixgrad_fwd_static(ptr(t), ptr(t), None, 1 << 15, 256, 256, 0, 0, 0, 1.0, None, None,
                                           ptr(t), stream())
    assert rc == _lib.EINVAL and b"exceed" in lib.unetk_last_error()
    rc = lib.unetk_loss_pixgrad_fwd_static(None, ptr(t), None, 1, 4, 4, 0, 0, 0, 1.0, None, None, ptr(t), stream())
    assert rc == _lib.EINVAL
