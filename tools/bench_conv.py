#!/usr/bin/env python
"""Per-layer timing of the MFMA conv kernels on the UNet shapes of the headline step (b64, 256^2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import ops, _lib
if os.environ.get("UNETK_AB_LIB"):      # A/B a kernel variant built to another file
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16
f = 48
layers = [("enc1.0", 256, 8, f), ("enc1.3", 256, f, f), ("enc2.0", 128, f, 2*f), ("enc2.3", 128, 2*f, 2*f),
          ("enc3.0", 64, 2*f, 4*f), ("enc3.3", 64, 4*f, 4*f), ("enc4.0", 32, 4*f, 8*f), ("enc4.3", 32, 8*f, 8*f),
          ("bott.0", 16, 8*f, 16*f), ("bott.3", 16, 16*f, 16*f), ("dec4.0", 32, 24*f, 8*f), ("dec3.0", 64, 12*f, 4*f),
          ("dec2.0", 128, 6*f, 2*f), ("dec1.0", 256, 3*f, f)]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

if os.environ.get("SHAPES") == "swin":   # the 3x3 convs of the Swin-UNet decoder at batch 64
    layers = [("d0.0", 16, 768, 512), ("d0.1", 16, 512, 512), ("d1.0", 32, 512, 256), ("d1.1", 32, 256, 256),
              ("d2.0", 64, 256, 128), ("d2.1", 64, 128, 128), ("d3.0", 128, 64, 64), ("d3.1", 128, 64, 64), ("final", 128, 64, 32)]
only = os.environ.get("LAYERS")          # e.g. LAYERS=dec3.0,enc2.3 to profile a few shapes
if only:
    layers = [l for l in layers if l[0] in only.split(",")]
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
for name, hw, ci, co in layers:
    x = torch.randn(B, hw, hw, ci, device="cuda").to(dt)
    dy = torch.randn(B, hw, hw, co, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    if os.environ.get("ZERO"):              # all-zero operands: same instruction stream, less switching power
        x.zero_(); dy.zero_(); w.zero_()
    pw, pwt = ops.PackedConv(w, dt), ops.PackedConv(w, dt, transpose=True)
    fl = 2.0 * B * hw * hw * ci * co * 9
    t_f = timeit(lambda: ops.conv_fwd(x, pw, want_stats=True))
    t_d = timeit(lambda: ops.conv_fwd(dy, pwt))
    ws = torch.empty(ops._lib.load().unetk_conv_wgrad_workspace_bytes(B, hw, hw, ci, co, 3, 1), dtype=torch.uint8, device="cuda")
    t_w = timeit(lambda: ops.conv_wgrad(x, dy, ci, co, 3, workspace=ws))
    tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
    if os.environ.get("BNSUMS") and ci == co and ops.conv_bwd_data_bnsums_supported(pwt, dt):
        # the data gradient that also leaves the next BatchNorm backward's sums, and that pass with / without them
        y = torch.randn(B, hw, hw, ci, device="cuda").to(dt)
        C = ci
        st = torch.stack([y.float().sum((0, 1, 2)), (y.float() ** 2).sum((0, 1, 2))]).view(1, 2, C).contiguous()
        mean, invstd, sc, sh = ops.bn_finalize(st, B * hw * hw, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), None, None)
        g, parts = ops.conv_bwd_data_bnsums(dy, pwt, y, sc, sh, mean, invstd)
        t_d2 = timeit(lambda: ops.conv_bwd_data_bnsums(dy, pwt, y, sc, sh, mean, invstd))
        t_b = timeit(lambda: ops.bn_act_bwd(g, y, sc, sh, mean, invstd))
        t_b2 = timeit(lambda: ops.bn_act_bwd(g, y, sc, sh, mean, invstd, presummed=parts))
        tot["bnsums"] = tot.get("bnsums", 0) + (t_d2 - t_d) - (t_b - t_b2)
        print(f"         dgrad+sums {t_d2:7.3f} ms (+{t_d2 - t_d:6.3f})   BN backward {t_b:7.3f} -> {t_b2:7.3f} ms ({t_b2 - t_b:+6.3f})", flush=True)
    print(f"{name:8s} {hw:4d}^2 {ci:5d}->{co:4d}  fwd {t_f:7.3f} ms {fl/t_f/1e9:7.0f} TF/s | dgrad {t_d:7.3f} ms {fl/t_d/1e9:7.0f} | wgrad {t_w:7.3f} ms {fl/t_w/1e9:7.0f}", flush=True)
print(tot)
