#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the imported reference.

Run ONLY where /root/reference exists (the build container):

    python tests/golden/make_golden.py

It (1) imports the reference modules with a stub ``xarray`` from a scratch working
directory (the reference opens a log file in its CWD at import time), (2) asserts
that the CPU oracle in ``oracle/`` agrees with the reference on every case, and
(3) freezes the reference's outputs as small ``.npz`` files next to this script.
The fixtures hold data only: seeds, shapes, expected numbers.  Weights and inputs
are regenerated from seeds by ``oracle.weights`` wherever the fixtures are used.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import filter_oracle, loss_oracle, patch_oracle, step_oracle, swin_oracle, tile_oracle, unet_oracle  # noqa: E402
from oracle import weights as W  # noqa: E402


def import_reference():
    import importlib.machinery
    xr = types.ModuleType("xarray")
    xr.__spec__ = importlib.machinery.ModuleSpec("xarray", None)
    xr.Dataset = type("Dataset", (), {})
    xr.DataArray = type("DataArray", (), {})
    xr.open_dataset = lambda *a, **k: None
    xr.open_dataarray = lambda *a, **k: None
    sys.modules["xarray"] = xr
    os.chdir(tempfile.mkdtemp(prefix="refscratch_"))
    sys.path.insert(0, REF)
    import enhanced_training_multi_loss as etml
    import large_area_predictor as lap
    import swin_unet_model as swin
    import training as tr
    return tr, etml, swin, lap


def close(a, b, tol, what):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).norm() / max(b.norm().item(), 1e-30)
    assert err <= tol, f"oracle != reference for {what}: rel {err:.3e}"
    return float(err)


def sub(t, n=4096):
    """Deterministic subsample of a tensor (flat stride)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


UNET_CASES = [  # name, in_ch, features, batch, H, W
    ("c1", 1, 48, 4, 128, 128),      # BASELINE config 1
    ("small", 4, 8, 2, 32, 32),
    ("ragged", 4, 8, 2, 40, 24),
    ("odd", 4, 16, 1, 50, 36),       # 50->25->12->6->3: resize is not an exact 2x
    ("f48", 4, 48, 1, 64, 64),
]


def gen_unet(tr):
    out = {}
    for name, cin, f, B, H, Wd in UNET_CASES:
        sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
        x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
        ref = tr.OptimizedUNet(cin, 1, f)
        ref.load_state_dict(sd)
        # eval forward
        ref.eval()
        with torch.no_grad():
            r_eval = ref(x)
        o_eval, _ = unet_oracle.unet_forward(sd, x, False)
        close(o_eval, r_eval, 1e-6, f"unet {name} eval")
        # train forward+backward, dropout p=0, MSE loss against y
        ref.train()
        for m in ref.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        ref.zero_grad()
        xr_ = x.clone().requires_grad_(True)
        r_out = ref(xr_)
        r_loss = ((r_out - y) ** 2).mean()
        r_loss.backward()
        o_out, o_loss, o_g, o_stats = unet_oracle.unet_fwd_bwd(sd, x, lambda o: ((o - y) ** 2).mean())
        close(o_out, r_out, 1e-6, f"unet {name} train out")
        close(o_loss, r_loss, 1e-6, f"unet {name} loss")
        rsd = ref.state_dict()
        for k, p in ref.named_parameters():
            close(o_g[k], p.grad, 2e-5, f"unet {name} grad {k}")
        close(o_g["__input__"], xr_.grad, 2e-5, f"unet {name} grad input")
        for k, v in o_stats.items():
            close(v, rsd[k], 1e-6, f"unet {name} stat {k}")
        names = [k for k, _ in ref.named_parameters()]
        out[f"{name}.eval_out"] = sub(r_eval, 16384)
        out[f"{name}.train_out"] = sub(r_out, 16384)
        out[f"{name}.loss"] = np.float64(r_loss.item())
        out[f"{name}.grad_norms"] = np.array([p.grad.double().norm().item() for _, p in ref.named_parameters()])
        out[f"{name}.grad_heads"] = np.concatenate([sub(p.grad, 16) for _, p in ref.named_parameters()])
        out[f"{name}.grad_input"] = sub(xr_.grad, 4096)
        out[f"{name}.param_names"] = np.array(names)
        stat_names = sorted(o_stats)
        out[f"{name}.stat_names"] = np.array(stat_names)
        out[f"{name}.stats"] = np.concatenate([sub(rsd[k], 8) for k in stat_names])
    # seeded dropout: the oracle draws masks exactly as nn.Dropout2d does
    name, cin, f, B, H, Wd = UNET_CASES[1]
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
    ref = tr.OptimizedUNet(cin, 1, f)
    ref.load_state_dict(sd)
    ref.train()
    torch.manual_seed(1234)
    r = ref(x)
    torch.manual_seed(1234)
    masks = unet_oracle.draw_dropout_masks(B, f)
    o, _ = unet_oracle.unet_forward(sd, x, True, masks)
    close(o, r, 1e-6, "unet seeded dropout")
    out["small.dropout_seed1234_out"] = sub(r, 16384)
    np.savez_compressed(os.path.join(HERE, "unet.npz"), **out)
    print("unet.npz", len(out), "entries")


LOSS_CFGS = {
    "baseline": dict(comb="baseline", use_tid=False, grad_w=0.0, spec_w=0.0),
    "weighted_mse": dict(comb="weighted_mse", grad_w=0.0, spec_w=0.0),
    "weighted_mse_grad": dict(comb="weighted_mse_grad", grad_w=0.3, spec_w=0.0),
    "weighted_mse_fft": dict(comb="weighted_mse_fft", grad_w=0.0, spec_w=0.2),
    "all": dict(comb="all", grad_w=0.3, spec_w=0.2),
    "all_aggr": dict(comb="all", grad_w=0.5, spec_w=0.3, scale=1.2, factor=2.5),
    "all_s05_f15": dict(comb="all", grad_w=0.3, spec_w=0.2, scale=0.5, factor=1.5),
    "all_s2_f3": dict(comb="all", grad_w=0.3, spec_w=0.2, scale=2.0, factor=3.0),
    "mae_all": dict(comb="all", grad_w=0.3, spec_w=0.2, pixel_type="mae"),
    "diff_grad": dict(comb="weighted_mse_grad", grad_w=0.3, spec_w=0.0, method="diff"),
}


def gen_loss(etml):
    out = {}
    shapes = [(3, 32, 32), (2, 24, 40), (2, 128, 128)]
    for si, (B, H, Wd) in enumerate(shapes):
        _, y, w = W.synthetic_batch(B, 1, H, Wd, seed=7 + si)
        p = W.synthetic_batch(B, 1, H, Wd, seed=100 + si)[1] * 0.7 + 0.3 * y
        for wname, wt in (("none", None), ("rand", w), ("zero", torch.zeros_like(w))):
            for cname, kw in LOSS_CFGS.items():
                cfg = loss_oracle.preset(**kw)
                ref = etml.MultiLossFunction(cfg)
                pr = p.clone().requires_grad_(True)
                r_total, r_parts = ref(pr, y, wt)
                r_total.backward()
                po = p.clone().requires_grad_(True)
                o_total, o_parts = loss_oracle.multi_loss(cfg, po, y, wt)
                o_total.backward()
                close(o_total, r_total, 1e-6, f"loss {cname}/{wname}/{si}")
                for k in ("pixel", "gradient", "spectral", "total"):
                    assert abs(o_parts[k] - r_parts[k]) <= 1e-6 * max(1.0, abs(r_parts[k])), (cname, k)
                close(po.grad, pr.grad, 1e-5, f"loss grad {cname}/{wname}/{si}")
                key = f"s{si}.{wname}.{cname}"
                out[key + ".parts"] = np.array([r_parts[k] for k in ("pixel", "gradient", "spectral", "total")])
                out[key + ".grad"] = sub(pr.grad, 2048)
    # frequency weight known answers (DC-heavy quirk)
    fw = loss_oracle.spectral_weight(128, 128, 2.0)
    assert abs(fw[0, 0].item() - 2.0) < 1e-6 and abs(fw[64, 64].item() - 1.0) < 1e-6
    out["fw_128_2.0"] = fw.numpy()
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)
    print("loss.npz", len(out), "entries")


class _Var:
    def __init__(self, a):
        self.a = a

    def isel(self, lat, lon):
        return types.SimpleNamespace(values=self.a[lat, lon])


class _DS(dict):
    """Duck-typed stand-in for the xarray.Dataset predict_large_area reads."""
    def __init__(self, arrays):
        super().__init__({k: _Var(v) for k, v in arrays.items()})
        any_a = next(iter(arrays.values()))
        self.lat = np.arange(any_a.shape[0])
        self.lon = np.arange(any_a.shape[1])


class _ToyNet(torch.nn.Module):
    """Cheap deterministic stand-in model: a fixed 3x3 conv over the 4 channels, then x/(1+|x|)."""
    def __init__(self):
        super().__init__()
        g = np.random.Generator(np.random.PCG64(5))
        self.w = torch.from_numpy(g.normal(0, 0.3, (1, 4, 3, 3)).astype(np.float32))

    def forward(self, x):
        # explicit shifted multiply-adds in a fixed order and IEEE-exact elementwise ops only, so
        # the bits of a sample's output do not depend on batch size, threading or vector tails
        xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
        H, Wd = x.shape[2], x.shape[3]
        acc = torch.zeros((x.shape[0], H, Wd), dtype=x.dtype)
        for c in range(4):
            for ky in range(3):
                for kx in range(3):
                    acc = acc + xp[:, c, ky:ky + H, kx:kx + Wd] * self.w[0, c, ky, kx]
        return (acc / (1.0 + acc.abs())).unsqueeze(1)


def toy_features(lat, lon, seed, nan_block=None, nan_speckle=0.0):
    g = np.random.Generator(np.random.PCG64(seed))
    arrays = {}
    for v in tile_oracle.FEATURE_VARS:
        a = g.normal(0, 1, (lat, lon)).astype(np.float32)
        if nan_speckle:
            a[g.random((lat, lon)) < nan_speckle] = np.nan
        arrays[v] = a
    if nan_block is not None:
        i0, i1, j0, j1 = nan_block
        arrays["DOV_NS"][i0:i1, j0:j1] = np.nan
    return arrays


TILE_CASES = [  # name, lat, lon, P, S, nan_block, speckle
    ("g200x168_p64_s32", 200, 168, 64, 32, (0, 70, 90, 168), 0.01),
    ("g160_p32_s2", 160, 160, 32, 2, None, 0.0),
    ("g512_p128_s64", 512, 512, 128, 64, (128, 300, 0, 200), 0.0),
    ("g300x260_p128_s8", 300, 260, 128, 8, None, 0.002),
]


def gen_tile(lap):
    out = {}
    for P in (8, 64, 128, 256):
        cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": P // 2, "BATCH_SIZE": 16})
        k_ref = lap.LargeAreaPredictor(cfg).create_gaussian_weight_kernel(P)
        k_or = tile_oracle.gaussian_weight_kernel(P)
        assert np.array_equal(k_ref, k_or), f"gaussian kernel P={P} not bit-exact"
        out[f"kernel_{P}"] = k_ref
    net = _ToyNet().eval()
    lap.device = torch.device("cpu")
    for name, lat, lon, P, S, nb, sp in TILE_CASES:
        feats = toy_features(lat, lon, 11, nb, sp)
        cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "BATCH_SIZE": 16})
        pred = lap.LargeAreaPredictor(cfg)
        r = pred.predict_large_area(net, _DS({k: v.copy() for k, v in feats.items()}))
        o = tile_oracle.predict_large_area(lambda b: net(torch.from_numpy(b)).numpy(),
                                           {k: v.copy() for k, v in feats.items()}, P, S, 16)
        assert np.array_equal(r, o), f"tiled predict {name} not bit-exact"
        pos, _ = tile_oracle.collect_tiles(feats, P, S)
        out[name + ".pos"] = np.array(pos, dtype=np.int32)
        out[name + ".pred"] = r.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "tile.npz"), **out)
    print("tile.npz", len(out), "entries")


SWIN_CASES = [("s256", 1, 256, 256), ("s64", 2, 64, 64), ("s128x64", 1, 128, 64)]


def gen_swin(swin):
    out = {}
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    ref = swin.SwinUNet(dict(cfg, use_checkpoint=False))
    template = {k: v for k, v in ref.state_dict().items()}
    assert list(template) == list(swin_oracle.swin_template(cfg)), "state_dict key order differs"
    for k, v in swin_oracle.swin_template(cfg).items():
        assert tuple(v.shape) == tuple(template[k].shape), k
        if k.endswith("relative_position_index"):
            assert torch.equal(v, template[k])
    sd = W.fill_state_dict(template, seed=42)
    ref.load_state_dict(sd)
    for name, B, H, Wd in SWIN_CASES:
        x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
        ref.eval()
        with torch.no_grad():
            r_eval = ref(x)
        o_eval, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
        close(o_eval, r_eval, 1e-5, f"swin {name} eval")
        out[f"{name}.eval_out"] = sub(r_eval, 16384)
        if name == "s256":
            continue
        # train mode, DropPath rate forced to 0, MSE against the half-resolution label.
        # Done in float64: with batch statistics over a handful of elements the fp32
        # gradients of either implementation carry ~5e-3 of rounding noise, so the
        # frozen numbers are the float64 ones (oracle == reference to 1e-12 there).
        ref64 = swin.SwinUNet(dict(cfg, use_checkpoint=False)).double()
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ref64.load_state_dict(sd64)
        ref64.train()
        for m in ref64.modules():
            if isinstance(m, swin.DropPath):
                m.drop_prob = 0.0
        x64, y64 = x.double(), y.double()
        xr_ = x64.clone().requires_grad_(True)
        r_out = ref64(xr_)
        tgt = torch.nn.functional.avg_pool2d(y64, 2)
        r_loss = ((r_out - tgt) ** 2).mean()
        r_loss.backward()
        params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
                  if v.is_floating_point() and "running_" not in k}
        full = dict(sd64)
        full.update(params)
        xo = x64.clone().requires_grad_(True)
        o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
        o_loss = ((o_out - tgt) ** 2).mean()
        o_loss.backward()
        close(o_out, r_out, 1e-12, f"swin {name} train out")
        rsd = ref64.state_dict()
        for k, p in ref64.named_parameters():
            if k == "final_conv.0.bias":      # a bias in front of BatchNorm: gradient is exactly 0
                assert p.grad.abs().max() < 1e-12 and params[k].grad.abs().max() < 1e-12
                continue
            close(params[k].grad, p.grad, 1e-9, f"swin {name} grad {k}")
        close(xo.grad, xr_.grad, 1e-9, f"swin {name} grad input")
        for k, v in o_stats.items():
            close(v, rsd[k], 1e-12, f"swin {name} stat {k}")
        out[f"{name}.train_out"] = sub(r_out, 16384)
        out[f"{name}.loss"] = np.float64(r_loss.item())
        out[f"{name}.grad_norms"] = np.array([p.grad.norm().item() for _, p in ref64.named_parameters()])
        out[f"{name}.param_names"] = np.array([k for k, _ in ref64.named_parameters()])
        out[f"{name}.grad_input"] = sub(xr_.grad, 4096)
        stat_names = sorted(o_stats)
        out[f"{name}.stat_names"] = np.array(stat_names)
        out[f"{name}.stats"] = np.concatenate([sub(rsd[k], 8) for k in stat_names])
    # seeded DropPath: oracle draws as drop_path() does
    name, B, H, Wd = SWIN_CASES[1]
    x, _, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    ref.load_state_dict(sd)
    ref.train()
    torch.manual_seed(99)
    r = ref(x)
    torch.manual_seed(99)
    dp = swin_oracle.draw_drop_path(cfg, B)
    o, _ = swin_oracle.swin_unet_forward(sd, x, cfg, True, dp)
    close(o, r, 1e-5, "swin seeded droppath")
    out["s64.droppath_seed99_out"] = sub(r, 16384)
    # per-stage shift masks at 256 px input (64,32,16,8 tokens a side) and a pad case
    for hw in (64, 32, 16, 8, 4):
        out[f"mask_{hw}"] = swin_oracle.shift_mask(hw, hw, 8).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "swin.npz"), **out)
    print("swin.npz", len(out), "entries")

def _flat(gs, names):
    return torch.cat([gs[n].double().reshape(-1) for n in names])


def _rel(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _cos(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))


def _ref_train_pass(model, x, loss_of, autocast):
    """One train-mode forward + backward of a reference module, optionally under CPU bf16 autocast.
    Returns (output fp32, loss, {name: grad}, input grad)."""
    import contextlib
    model.zero_grad()
    xin = x.clone().requires_grad_(True)
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
    with ctx:
        out = model(xin)
    out32 = out.float()
    loss = loss_of(out32)
    loss.backward()
    return out32.detach(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, \
        xin.grad.detach().clone()


def _bf16_record(out, prefix, model, x, loss_of):
    """Distance of the reference's own bf16-autocast run from its fp32 run on one case."""
    o32, l32, g32, dx32 = _ref_train_pass(model, x, loss_of, False)
    o16, l16, g16, dx16 = _ref_train_pass(model, x, loss_of, True)
    names = [k for k, _ in model.named_parameters()]
    a, b = _flat(g16, names), _flat(g32, names)
    # parameters whose gradient is pure rounding noise (a conv bias in front of BatchNorm) carry no signal
    floor = 1e-6 * float(b.norm())
    per = [(_rel(g16[k], g32[k]) if float(g32[k].double().norm()) > floor else 0.0) for k in names]
    rec = dict(out_rel=_rel(o16, o32), loss_rel=abs(l16 - l32) / abs(l32), grad_rel=_rel(a, b), grad_cos=_cos(a, b),
               dx_rel=_rel(dx16, dx32), dx_cos=_cos(dx16, dx32), grad_rel_worst=max(per))
    for k, v in rec.items():
        out[f"{prefix}.{k}"] = np.float64(v)
    out[f"{prefix}.param_names"] = np.array(names)
    out[f"{prefix}.grad_rel_per_param"] = np.array(per)
    print(prefix, {k: round(v, 4) for k, v in rec.items()})


def gen_bf16(tr, swin):
    """What bfloat16 costs the REFERENCE: its modules run train-mode forward + backward under
    torch.autocast('cpu', dtype=torch.bfloat16) and are compared with their own fp32 run on the same
    seeded cases as unet.npz / swin.npz.  These numbers bound the bf16 GPU path (no worse than 1.3x)."""
    out = {}
    for name, cin, f, B, H, Wd in UNET_CASES:
        sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
        x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
        ref = tr.OptimizedUNet(cin, 1, f)
        ref.load_state_dict(sd)
        ref.train()
        for m in ref.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        _bf16_record(out, f"unet.{name}", ref, x, lambda o, y=y: ((o - y) ** 2).mean())
        # eval forward under autocast against fp32 eval
        ref.eval()
        with torch.no_grad():
            e32 = ref(x)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                e16 = ref(x).float()
        out[f"unet.{name}.eval_rel"] = np.float64(_rel(e16, e32))
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    ref = swin.SwinUNet(dict(cfg, use_checkpoint=False))
    sd = W.fill_state_dict({k: v for k, v in ref.state_dict().items()}, seed=42)
    for name, B, H, Wd in SWIN_BF16_CASES:
        x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
        ref.load_state_dict(sd)
        ref.train()
        for m in ref.modules():
            if isinstance(m, swin.DropPath):
                m.drop_prob = 0.0
        tgt = torch.nn.functional.avg_pool2d(y, 2)
        _bf16_record(out, f"swin.{name}", ref, x, lambda o, tgt=tgt: ((o - tgt) ** 2).mean())
        ref.eval()
        with torch.no_grad():
            e32 = ref(x)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                e16 = ref(x).float()
        out[f"swin.{name}.eval_rel"] = np.float64(_rel(e16, e32))
    np.savez_compressed(os.path.join(HERE, "bf16.npz"), **out)
    print("bf16.npz", len(out), "entries")


SWIN_BF16_CASES = [("s64", 2, 64, 64), ("s128x64", 1, 128, 64), ("s256", 1, 256, 256)]

SWIN2_CASES = [("s256", 1, 256, 256), ("s24", 2, 24, 24)]


def gen_swin2(swin):
    """Two more Swin cases, train mode in float64 as in gen_swin: the full-size no-pad path (256 px:
    64/32/16/8 tokens a side, 64/16/4/1 windows) and a 24-px input, whose 6x6 / 3x3 / 2x2 / 1x1 token
    grids take the padded-window path in every stage, the odd-size PatchMerging pad, and the decoder's
    bilinear resize of the projected skip (swin_unet_model.py:526-529: 3x3 -> 4x4 and 6x6 -> 8x8)."""
    out = {}
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    ref32 = swin.SwinUNet(dict(cfg, use_checkpoint=False))
    sd = W.fill_state_dict({k: v for k, v in ref32.state_dict().items()}, seed=42)
    ref32.load_state_dict(sd)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for name, B, H, Wd in SWIN2_CASES:
        x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
        if name != "s256":       # s256.eval_out is in swin.npz already
            ref32.eval()
            with torch.no_grad():
                r_eval = ref32(x)
            o_eval, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
            close(o_eval, r_eval, 1e-5, f"swin {name} eval")
            out[f"{name}.eval_out"] = sub(r_eval, 16384)
        ref64 = swin.SwinUNet(dict(cfg, use_checkpoint=False)).double()
        ref64.load_state_dict(sd64)
        ref64.train()
        for m in ref64.modules():
            if isinstance(m, swin.DropPath):
                m.drop_prob = 0.0
        x64 = x.double()
        xr_ = x64.clone().requires_grad_(True)
        r_out = ref64(xr_)
        # the reference trainer resizes the prediction to the label's size (:812-814); here the label is
        # resized to the prediction instead so that only the model is under test
        tgt = torch.nn.functional.interpolate(y.double(), size=r_out.shape[-2:], mode="bilinear", align_corners=False)
        r_loss = ((r_out - tgt) ** 2).mean()
        r_loss.backward()
        params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
                  if v.is_floating_point() and "running_" not in k}
        full = dict(sd64)
        full.update(params)
        xo = x64.clone().requires_grad_(True)
        o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
        ((o_out - tgt) ** 2).mean().backward()
        close(o_out, r_out, 1e-12, f"swin {name} train out")
        rsd = ref64.state_dict()
        for k, p in ref64.named_parameters():
            if k == "final_conv.0.bias":
                continue
            close(params[k].grad, p.grad, 1e-9, f"swin {name} grad {k}")
        close(xo.grad, xr_.grad, 1e-9, f"swin {name} grad input")
        for k, v in o_stats.items():
            close(v, rsd[k], 1e-12, f"swin {name} stat {k}")
        out[f"{name}.out_shape"] = np.array(r_out.shape)
        out[f"{name}.train_out"] = sub(r_out, 16384)
        out[f"{name}.loss"] = np.float64(r_loss.item())
        out[f"{name}.grad_norms"] = np.array([p.grad.norm().item() for _, p in ref64.named_parameters()])
        out[f"{name}.param_names"] = np.array([k for k, _ in ref64.named_parameters()])
        out[f"{name}.grad_input"] = sub(xr_.grad, 4096)
        stat_names = sorted(o_stats)
        out[f"{name}.stat_names"] = np.array(stat_names)
        out[f"{name}.stats"] = np.concatenate([sub(rsd[k], 8) for k in stat_names])
    np.savez_compressed(os.path.join(HERE, "swin2.npz"), **out)
    print("swin2.npz", len(out), "entries")


# Configurations whose windows / heads are NOT 8 / 32 (the general attention kernels): name, overrides of the default
# configuration, (B, H, W) of the input
SWIN_GEOM_CASES = [
    ("w4h16", dict(window_size=4, embed_dim=48, depths=[2, 2, 2, 2], decoder_channels=[64, 48, 32, 16]), (2, 64, 64)),
    ("w7h32", dict(window_size=7, depths=[2, 2, 2, 2], decoder_channels=[128, 64, 32, 16]), (1, 112, 112)),
    ("w8h64", dict(window_size=8, embed_dim=128, num_heads=[2, 4, 8, 16], depths=[2, 2, 2, 2],
                   decoder_channels=[128, 64, 32, 16]), (1, 64, 64)),
]


def gen_swin_geom(swin):
    """The reference with other window sides / head dimensions (swin_unet_model.py:545-554 takes any): eval output in
    float32, train-mode forward + backward in float64 with DropPath off, oracle == reference asserted first."""
    out = {}
    for name, over, (B, H, Wd) in SWIN_GEOM_CASES:
        cfg = dict(swin_oracle.DEFAULT_CONFIG)
        cfg.update(over)
        ref32 = swin.SwinUNet(dict(cfg, use_checkpoint=False))
        sd = W.fill_state_dict({k: v for k, v in ref32.state_dict().items()}, seed=42)
        ref32.load_state_dict(sd)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
        ref32.eval()
        with torch.no_grad():
            r_eval = ref32(x)
        o_eval, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
        close(o_eval, r_eval, 1e-5, f"swin {name} eval")
        out[f"{name}.eval_out"] = sub(r_eval, 16384)
        ref64 = swin.SwinUNet(dict(cfg, use_checkpoint=False)).double()
        ref64.load_state_dict(sd64)
        ref64.train()
        for m in ref64.modules():
            if isinstance(m, swin.DropPath):
                m.drop_prob = 0.0
        xr_ = x.double().clone().requires_grad_(True)
        r_out = ref64(xr_)
        tgt = torch.nn.functional.interpolate(y.double(), size=r_out.shape[-2:], mode="bilinear", align_corners=False)
        r_loss = ((r_out - tgt) ** 2).mean()
        r_loss.backward()
        params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
                  if v.is_floating_point() and "running_" not in k}
        full = dict(sd64)
        full.update(params)
        xo = x.double().clone().requires_grad_(True)
        o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, None)
        ((o_out - tgt) ** 2).mean().backward()
        close(o_out, r_out, 1e-12, f"swin {name} train out")
        for k, p in ref64.named_parameters():
            if k == "final_conv.0.bias":
                continue
            close(params[k].grad, p.grad, 1e-9, f"swin {name} grad {k}")
        close(xo.grad, xr_.grad, 1e-9, f"swin {name} grad input")
        out[f"{name}.out_shape"] = np.array(r_out.shape)
        out[f"{name}.train_out"] = sub(r_out, 16384)
        out[f"{name}.loss"] = np.float64(r_loss.item())
        out[f"{name}.grad_norms"] = np.array([p.grad.norm().item() for _, p in ref64.named_parameters()])
        out[f"{name}.param_names"] = np.array([k for k, _ in ref64.named_parameters()])
        out[f"{name}.grad_input"] = sub(xr_.grad, 4096)
    np.savez_compressed(os.path.join(HERE, "swin_geom.npz"), **out)
    print("swin_geom.npz", len(out), "entries")


MASKED_UNET = [("small", 1234), ("ragged", 77), ("f48", 5)]      # case of UNET_CASES, torch seed of the draw
MASKED_SWIN = [("s64", 2, 64, 64, 16), ("s256", 1, 256, 256, 17), ("s24", 2, 24, 24, 3)]


def gen_masked(tr, swin):
    """Train-mode forward + BACKWARD with live Dropout2d (training.py:530) / DropPath (swin_unet_model.py:139-158,
    371-372).  The keep factors are the ones a seeded fp32 run of the reference draws (checked: the seeded fp32
    reference run and the same modules with those factors patched in give identical bits); the frozen numbers come
    from the reference in float64 with the factors patched into its Dropout2d / DropPath modules, for the reason
    given in gen_swin (fp32 gradients of either implementation carry rounding noise of their own)."""
    out = {}
    for name, seed in MASKED_UNET:
        _, cin, f, B, H, Wd = next(c for c in UNET_CASES if c[0] == name)
        sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
        x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
        torch.manual_seed(seed)
        masks = unet_oracle.draw_dropout_masks(B, f)
        assert any(bool((m == 0).any()) for m in masks), f"{name}: seed {seed} drops nothing"

        def patched(dtype):
            ref = tr.OptimizedUNet(cin, 1, f).to(dtype)
            ref.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()})
            ref.train()
            for blk, m in zip(unet_oracle.BLOCKS, masks):
                drop = getattr(ref, blk)[6]
                assert isinstance(drop, torch.nn.Dropout2d)
                drop.forward = (lambda t, k=m.to(dtype): t * k.view(k.shape[0], k.shape[1], 1, 1))
            return ref
        seeded = tr.OptimizedUNet(cin, 1, f)
        seeded.load_state_dict(sd)
        seeded.train()
        torch.manual_seed(seed)
        assert torch.equal(seeded(x), patched(torch.float32)(x)), f"{name}: patched masks are not the seeded draw"
        ref64 = patched(torch.float64)
        x64, y64 = x.double(), y.double()
        xr_ = x64.clone().requires_grad_(True)
        r_out = ref64(xr_)
        r_loss = ((r_out - y64) ** 2).mean()
        r_loss.backward()
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        o_out, o_loss, o_g, o_stats = unet_oracle.unet_fwd_bwd(sd64, x64, lambda o: ((o - y64) ** 2).mean(),
                                                                [m.double() for m in masks])
        close(o_out, r_out, 1e-12, f"masked unet {name} out")
        rsd = ref64.state_dict()
        for k, p in ref64.named_parameters():
            close(o_g[k], p.grad, 1e-9, f"masked unet {name} grad {k}")
        close(o_g["__input__"], xr_.grad, 1e-9, f"masked unet {name} grad input")
        for k, v in o_stats.items():
            close(v, rsd[k], 1e-12, f"masked unet {name} stat {k}")
        pre = f"unet.{name}"
        out[f"{pre}.seed"] = np.int64(seed)
        out[f"{pre}.dropped"] = np.array([int((m == 0).sum()) for m in masks])
        out[f"{pre}.train_out"] = sub(r_out, 16384)
        out[f"{pre}.loss"] = np.float64(r_loss.item())
        out[f"{pre}.param_names"] = np.array([k for k, _ in ref64.named_parameters()])
        out[f"{pre}.grad_norms"] = np.array([p.grad.norm().item() for _, p in ref64.named_parameters()])
        out[f"{pre}.grad_heads"] = np.concatenate([sub(p.grad, 16) for _, p in ref64.named_parameters()])
        out[f"{pre}.grad_input"] = sub(xr_.grad, 4096)
        stat_names = sorted(o_stats)
        out[f"{pre}.stat_names"] = np.array(stat_names)
        out[f"{pre}.stats"] = np.concatenate([sub(rsd[k], 8) for k in stat_names])
        print(pre, "dropped channels per block", out[f"{pre}.dropped"].tolist())
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    ref32 = swin.SwinUNet(dict(cfg, use_checkpoint=False))
    sd = W.fill_state_dict({k: v for k, v in ref32.state_dict().items()}, seed=42)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for name, B, H, Wd, seed in MASKED_SWIN:
        x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
        torch.manual_seed(seed)
        dp = swin_oracle.draw_drop_path(cfg, B)
        assert any(k is not None and bool((k == 0).any()) for pair in dp for k in pair), f"{name}: nothing dropped"

        def patch(model, dtype):
            blocks = [blk for layer in model.layers for blk in layer.blocks]
            assert len(blocks) == len(dp)
            for blk, pair in zip(blocks, dp):
                if pair[0] is None:
                    assert not isinstance(blk.drop_path, swin.DropPath)
                    continue
                assert isinstance(blk.drop_path, swin.DropPath)
                state = {"n": 0}

                def fwd(t, pair=pair, state=state):       # called for the attention branch, then for the MLP branch
                    k = pair[state["n"] % 2].to(dtype)
                    state["n"] += 1
                    return t * k.view(-1, 1, 1)
                blk.drop_path.forward = fwd
            return model
        ref32.load_state_dict(sd)
        ref32.train()
        torch.manual_seed(seed)
        r_seeded = ref32(x)
        p32 = swin.SwinUNet(dict(cfg, use_checkpoint=False))
        p32.load_state_dict(sd)
        p32.train()
        # x.div(keep) * floor(...) against x * (floor(...)/keep): the same number to an ulp per branch
        close(patch(p32, torch.float32)(x), r_seeded, 1e-5, f"masked swin {name}: patched factors vs the seeded draw")
        ref64 = swin.SwinUNet(dict(cfg, use_checkpoint=False)).double()
        ref64.load_state_dict(sd64)
        ref64.train()
        patch(ref64, torch.float64)
        x64 = x.double()
        xr_ = x64.clone().requires_grad_(True)
        r_out = ref64(xr_)
        tgt = torch.nn.functional.interpolate(y.double(), size=r_out.shape[-2:], mode="bilinear", align_corners=False)
        r_loss = ((r_out - tgt) ** 2).mean()
        r_loss.backward()
        params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
                  if v.is_floating_point() and "running_" not in k}
        full = dict(sd64)
        full.update(params)
        xo = x64.clone().requires_grad_(True)
        dp64 = [[None if k is None else k.double() for k in pair] for pair in dp]
        o_out, o_stats = swin_oracle.swin_unet_forward(full, xo, cfg, True, dp64)
        ((o_out - tgt) ** 2).mean().backward()
        close(o_out, r_out, 1e-12, f"masked swin {name} out")
        rsd = ref64.state_dict()
        for k, p in ref64.named_parameters():
            if k == "final_conv.0.bias":
                continue
            close(params[k].grad, p.grad, 1e-9, f"masked swin {name} grad {k}")
        close(xo.grad, xr_.grad, 1e-9, f"masked swin {name} grad input")
        pre = f"swin.{name}"
        out[f"{pre}.seed"] = np.int64(seed)
        out[f"{pre}.dropped"] = np.array([0 if k is None else int((k == 0).sum()) for pair in dp for k in pair])
        out[f"{pre}.train_out"] = sub(r_out, 16384)
        out[f"{pre}.loss"] = np.float64(r_loss.item())
        out[f"{pre}.param_names"] = np.array([k for k, _ in ref64.named_parameters()])
        out[f"{pre}.grad_norms"] = np.array([p.grad.norm().item() for _, p in ref64.named_parameters()])
        out[f"{pre}.grad_heads"] = np.concatenate([sub(p.grad, 16) for _, p in ref64.named_parameters()])
        out[f"{pre}.grad_input"] = sub(xr_.grad, 4096)
        stat_names = sorted(o_stats)
        out[f"{pre}.stat_names"] = np.array(stat_names)
        out[f"{pre}.stats"] = np.concatenate([sub(rsd[k], 8) for k in stat_names])
        print(pre, "dropped samples per branch", out[f"{pre}.dropped"].tolist())
    np.savez_compressed(os.path.join(HERE, "masked.npz"), **out)
    print("masked.npz", len(out), "entries")


SWIN_STEP_LOSS = dict(comb="weighted_mse_grad", grad_w=0.3, spec_w=0.0)      # SWIN_LOSS_CONFIGS['tid_gradient']


def gen_step_swin(swin, etml):
    """One optimiser step as the Swin trainer does it (swin_unet_model.py:809-823): model, bilinear resize of
    the half-resolution prediction to the label size, MultiLossFunction('tid_gradient'), zero_grad, backward,
    clip_grad_norm_(1.0), AdamW(lr 1e-4, weight decay 0.05).  DropPath off."""
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    B, H, Wd = 2, 64, 64
    ref = swin.SwinUNet(dict(cfg, use_checkpoint=False))
    sd = W.fill_state_dict({k: v for k, v in ref.state_dict().items()}, seed=42)
    ref.load_state_dict(sd)
    ref.train()
    for m in ref.modules():
        if isinstance(m, swin.DropPath):
            m.drop_prob = 0.0
    x, y, w = W.synthetic_batch(B, 4, H, Wd, seed=42)
    lcfg = loss_oracle.preset(**SWIN_STEP_LOSS)
    assert {k: lcfg[k] for k in lcfg} == {k: swin.SWIN_LOSS_CONFIGS["tid_gradient"][k] for k in lcfg}
    crit = etml.MultiLossFunction(lcfg)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=0.05)
    outputs = ref(x)
    assert outputs.shape[-2:] == (H // 2, Wd // 2)
    outputs = torch.nn.functional.interpolate(outputs, size=y.shape[-2:], mode="bilinear", align_corners=False)
    loss, parts = crit(outputs, y, w)
    opt.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
    opt.step()
    o_parts, o_norm, o_new, o_grads = step_oracle.swin_training_step(sd, x, y, w, cfg, lcfg)
    for k in ("pixel", "gradient", "spectral", "total"):
        assert abs(o_parts[k] - parts[k]) <= 1e-5 * max(1.0, abs(parts[k])), (k, o_parts, parts)
    assert abs(o_norm - gn.item()) <= 1e-3 * gn.item(), (o_norm, gn.item())
    names = [k for k, _ in ref.named_parameters()]
    out = {"parts": np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")]),
           "grad_norm": np.float64(gn.item()),
           "param_names": np.array(names),
           "grad_norms": np.array([grads[k].double().norm().item() for k in names]),
           "param_delta_norms": np.array([(p.detach() - sd[k]).double().norm().item()
                                          for k, p in ref.named_parameters()]),
           "param_heads": np.concatenate([sub(p, 8) for _, p in ref.named_parameters()])}
    np.savez_compressed(os.path.join(HERE, "step_swin.npz"), **out)
    print("step_swin.npz", len(out), "entries; grad norm", gn.item(), parts)


def gen_step(etml):
    """One optimiser step as the trainer does it (enhanced_training_multi_loss.py:819-829)."""
    cin, f, B, H, Wd = 4, 8, 2, 32, 32
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
    ref = etml.OptimizedUNet(cin, 1, f)
    ref.load_state_dict(sd)
    ref.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    crit = etml.MultiLossFunction(loss_oracle.preset("all"))
    opt = torch.optim.AdamW(ref.parameters(), lr=5e-5, weight_decay=5e-5, betas=(0.9, 0.999))
    loss, parts = crit(ref(x), y, w)
    opt.zero_grad()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
    opt.step()
    out = {"parts": np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")]),
           "grad_norm": np.float64(gn.item()),
           "param_names": np.array([k for k, _ in ref.named_parameters()]),
           "param_delta_norms": np.array([(p.detach() - sd[k]).double().norm().item()
                                          for k, p in ref.named_parameters()]),
           "param_heads": np.concatenate([sub(p, 8) for _, p in ref.named_parameters()])}
    np.savez_compressed(os.path.join(HERE, "step.npz"), **out)
    print("step.npz", len(out), "entries")


PATCH_CASES = [("p16", 96, 112, 16, 8, 0.25, 21), ("p32", 160, 128, 32, 16, 0.25, 22), ("p8", 40, 33, 8, 3, 0.5, 23)]


class _DA(_Var):
    """Duck-typed xarray.DataArray: .lat, .lon, .isel(lat=, lon=).values"""
    def __init__(self, a):
        super().__init__(a)
        self.lat, self.lon = np.arange(a.shape[0]), np.arange(a.shape[1])


def gen_patch(etml):
    """Training-patch extraction and sample preparation (enhanced_training_multi_loss.py:633-720, 470-513)."""
    out = {}
    for name, lat, lon, P, S, mv, seed in PATCH_CASES:
        feats, labels, weights = patch_oracle.synthetic_fields(lat, lon, seed)
        cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "MIN_VALID_RATIO": mv,
                             "USE_DATA_AUGMENTATION": False, "AUGMENTATION_PROB": 0.0})
        ref = etml.EnhancedDataPreparation(cfg).extract_patches(_DS(feats), _DA(labels), _DA(weights))
        mine = patch_oracle.extract_patches(feats, labels, weights, P, S, mv)
        assert len(ref) == len(mine) and len(ref) > 0, (name, len(ref), len(mine))
        ncand = len(patch_oracle.candidate_starts(lat, P, S)) * len(patch_oracle.candidate_starts(lon, P, S))
        assert len(ref) < ncand, f"{name}: every candidate passed, the rules are not exercised"
        ds = etml.EnhancedBathymetryDataset(ref, cfg, is_training=False)
        sums = []
        for k, (a, b) in enumerate(zip(ref, mine)):
            assert (a["lat_start"], a["lon_start"]) == (b["lat_start"], b["lon_start"])
            for key in ("features", "label", "tid_weights"):
                assert np.array_equal(a[key], b[key], equal_nan=True), (name, k, key)
            f, l, w = ds[k]
            fo, lo, wo = patch_oracle.dataset_item(b)
            assert np.array_equal(f.numpy(), fo) and np.array_equal(l.numpy(), lo) and np.array_equal(w.numpy(), wo)
            assert np.array_equal(torch.flip(f, dims=[2]).numpy(), patch_oracle.dataset_item(b, "h")[0])
            assert np.array_equal(torch.flip(l, dims=[1]).numpy(), patch_oracle.dataset_item(b, "v")[1])
            sums.append([fo.astype(np.float64).sum(), lo.astype(np.float64).sum(), wo.astype(np.float64).sum()])
        out[name + ".pos"] = np.array([(p["lat_start"], p["lon_start"]) for p in ref], dtype=np.int32)
        out[name + ".sums"] = np.array(sums)
        out[name + ".ncand"] = np.int64(ncand)
    np.savez_compressed(os.path.join(HERE, "patch.npz"), **out)
    print("patch.npz", len(out), "entries")


FILTER_CASES = [  # name, lat, lon, sigma_lat, sigma_lon, NaN fraction, seed
    ("small", 40, 56, 2.5, 3.25, 0.05, 3),
    ("wide_sigma", 48, 300, 1.0, 20.0, 0.0, 4),       # radius 80 > lat extent is not hit; lon reflects deep
    ("short_axis", 7, 90, 6.0, 2.0, 0.1, 5),          # radius 24 > 7 rows: the reflection wraps more than once
]


def gen_filter():
    """Gaussian low-pass of the data preparation (data_preparation.py:392-414), run through the reference's own
    method with a minimal stand-in for the xarray container it wraps the result in."""
    import data_preparation as dp

    class DA:
        def __init__(self, values, coords=None, dims=None, name=None):
            self.values, self.coords, self.dims, self.name, self.attrs = values, coords, dims, name, {}
    dp.xr.DataArray = DA
    out = {}
    for name, lat, lon, s_lat, s_lon, nan_frac, seed in FILTER_CASES:
        x = filter_oracle.synthetic_field(lat, lon, nan_frac, seed)
        ref = dp.EnhancedBathymetryProcessor.apply_gaussian_filter(None, DA(x, {}, ("lat", "lon"), "bathy"), s_lat, s_lon).values
        mine = filter_oracle.apply_gaussian_filter(x, s_lat, s_lon)
        assert np.array_equal(np.isnan(ref), np.isnan(mine)), name
        err = np.nanmax(np.abs(ref - mine)) / np.nanmax(np.abs(ref))
        assert err < 1e-13, (name, err)
        out[name] = ref
    np.savez_compressed(os.path.join(HERE, "filter.npz"), **out)
    print("filter.npz", len(out), "entries")


if __name__ == "__main__":
    torch.set_num_threads(8)
    tr, etml, swin, lap = import_reference()
    which = sys.argv[1:] or ["unet", "loss", "tile", "swin", "step", "patch", "filter", "bf16", "swin2", "step_swin", "masked", "swin_geom"]
    if "unet" in which:
        gen_unet(tr)
    if "loss" in which:
        gen_loss(etml)
    if "tile" in which:
        gen_tile(lap)
    if "swin" in which:
        gen_swin(swin)
    if "step" in which:
        gen_step(etml)
    if "bf16" in which:
        gen_bf16(tr, swin)
    if "swin2" in which:
        gen_swin2(swin)
    if "step_swin" in which:
        gen_step_swin(swin, etml)
    if "masked" in which:
        gen_masked(tr, swin)
    if "swin_geom" in which:
        gen_swin_geom(swin)
    if "patch" in which:
        gen_patch(etml)
    if "filter" in which:
        gen_filter()
