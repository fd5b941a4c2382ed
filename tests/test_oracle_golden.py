"""CPU tests: the oracle reproduces the fixtures frozen from the imported reference."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle, patch_oracle, swin_oracle, tile_oracle, unet_oracle
from oracle import weights as W


def sub(t, n=4096):
    f = torch.as_tensor(t).detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


UNET_CASES = {"c1": (1, 48, 4, 128, 128), "small": (4, 8, 2, 32, 32), "ragged": (4, 8, 2, 40, 24),
              "odd": (4, 16, 1, 50, 36), "f48": (4, 48, 1, 64, 64)}


@pytest.mark.parametrize("name", ["small", "ragged", "odd", "f48", "c1"])
def test_unet_oracle_matches_golden(golden, name):
    g = golden("unet")
    cin, f, B, H, Wd = UNET_CASES[name]
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
    o_eval, _ = unet_oracle.unet_forward(sd, x, False)
    assert rel(sub(o_eval, 16384), g[f"{name}.eval_out"]) < 1e-6
    out, loss, grads, stats = unet_oracle.unet_fwd_bwd(sd, x, lambda o: ((o - y) ** 2).mean())
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-6
    assert abs(loss.item() - g[f"{name}.loss"]) < 1e-6 * max(1, abs(g[f"{name}.loss"]))
    names = [str(s) for s in g[f"{name}.param_names"]]
    norms = np.array([grads[k].double().norm().item() for k in names])
    assert rel(norms, g[f"{name}.grad_norms"]) < 1e-4
    assert rel(sub(grads["__input__"], 4096), g[f"{name}.grad_input"]) < 1e-4
    st = np.concatenate([sub(stats[str(k)], 8) for k in g[f"{name}.stat_names"]])
    assert rel(st, g[f"{name}.stats"]) < 1e-6


def test_unet_param_count_known_answers():
    # reference training log :19 and the 1-channel variant
    n4 = sum(v.numel() for k, v in W.unet_template(4, 1, 48).items()
             if v.is_floating_point() and "running_" not in k)
    n1 = sum(v.numel() for k, v in W.unet_template(1, 1, 48).items()
             if v.is_floating_point() and "running_" not in k)
    assert n4 == 17_656_945 and n1 == 17_655_649
    assert len(W.unet_template()) == 110


def test_unet_seeded_dropout(golden):
    g = golden("unet")
    cin, f, B, H, Wd = UNET_CASES["small"]
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, _, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
    torch.manual_seed(1234)
    masks = unet_oracle.draw_dropout_masks(B, f)
    o, _ = unet_oracle.unet_forward(sd, x, True, masks)
    assert rel(sub(o, 16384), g["small.dropout_seed1234_out"]) < 1e-6


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
LOSS_SHAPES = [(3, 32, 32), (2, 24, 40), (2, 128, 128)]


def loss_inputs(si):
    B, H, Wd = LOSS_SHAPES[si]
    _, y, w = W.synthetic_batch(B, 1, H, Wd, seed=7 + si)
    p = W.synthetic_batch(B, 1, H, Wd, seed=100 + si)[1] * 0.7 + 0.3 * y
    return p, y, w


@pytest.mark.parametrize("si", [0, 1, 2])
def test_loss_oracle_matches_golden(golden, si):
    g = golden("loss")
    p, y, w = loss_inputs(si)
    for wname, wt in (("none", None), ("rand", w), ("zero", torch.zeros_like(w))):
        for cname, kw in LOSS_CFGS.items():
            cfg = loss_oracle.preset(**kw)
            po = p.clone().requires_grad_(True)
            total, parts = loss_oracle.multi_loss(cfg, po, y, wt)
            total.backward()
            key = f"s{si}.{wname}.{cname}"
            got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
            assert np.allclose(got, g[key + ".parts"], rtol=1e-6, atol=1e-9), key
            assert rel(sub(po.grad, 2048), g[key + ".grad"]) < 1e-5, key


def test_spectral_weight_is_dc_heavy(golden):
    fw = loss_oracle.spectral_weight(128, 128, 2.0)
    assert np.array_equal(fw.numpy(), golden("loss")["fw_128_2.0"])
    assert abs(fw[0, 0].item() - 2.0) < 1e-6 and abs(fw[64, 64].item() - 1.0) < 1e-6


def test_spectral_without_high_freq_focus_raises():
    p, y, w = loss_inputs(0)
    with pytest.raises(NameError):
        loss_oracle.spectral_loss(p, y, w, focus_high_freq=False)


@pytest.mark.parametrize("P", [8, 64, 128, 256])
def test_gaussian_kernel_bit_exact(golden, P):
    k = tile_oracle.gaussian_weight_kernel(P)
    assert k.dtype == np.float32 and np.array_equal(k, golden("tile")[f"kernel_{P}"])
    if P == 256:   # known answers
        assert abs(float(k.min()) - 0.018895887) < 1e-9 and abs(float(k.max()) - 0.999938965) < 1e-9


def test_tile_count_known_answer():
    assert len(tile_oracle.tile_starts(8192, 256, 128)) ** 2 == 3969


def test_swin_known_answers(golden):
    tpl = swin_oracle.swin_template()
    assert len(tpl) == 249
    n = sum(v.numel() for k, v in tpl.items() if v.is_floating_point() and "running_" not in k)
    assert n == 37_810_795
    g = golden("swin")
    for hw in (64, 32, 16, 8, 4):
        assert np.array_equal(swin_oracle.shift_mask(hw, hw, 8).numpy(), g[f"mask_{hw}"])
    # the mask is applied even when the map is one window: 75 % of entries are -100
    assert abs((g["mask_8"] == -100).mean() - 0.75) < 1e-12


@pytest.mark.parametrize("name,B,H,Wd", [("s64", 2, 64, 64), ("s128x64", 1, 128, 64)])
def test_swin_oracle_matches_golden(golden, name, B, H, Wd):
    g = golden("swin")
    cfg = swin_oracle.DEFAULT_CONFIG
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    o, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
    assert rel(sub(o, 16384), g[f"{name}.eval_out"]) < 1e-5
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
              if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    out, stats = swin_oracle.swin_unet_forward(full, x.double(), cfg, True, None)
    loss = ((out - torch.nn.functional.avg_pool2d(y.double(), 2)) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - g[f"{name}.loss"]) < 1e-9
    names = [str(s) for s in g[f"{name}.param_names"]]
    norms = np.array([params[k].grad.norm().item() for k in names])
    keep = np.array([k != "final_conv.0.bias" for k in names])
    assert rel(norms[keep], g[f"{name}.grad_norms"][keep]) < 1e-8


def test_training_step_oracle_matches_golden(golden):
    """loss -> backward -> clip_grad_norm_(1.0) -> AdamW as the reference trainer does it."""
    from oracle import step_oracle
    g = golden("step")
    cin, f, B, H, Wd = 4, 8, 2, 32, 32
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    x, y, w = W.synthetic_batch(B, cin, H, Wd, seed=42)
    parts, norm, new = step_oracle.training_step(sd, x, y, w, loss_oracle.preset("all"))
    want = g["parts"]
    got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
    assert rel(got, want) < 1e-5
    assert abs(norm - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    names = [str(s) for s in g["param_names"]]
    delta = np.array([(new[k] - sd[k]).double().norm().item() for k in names])
    assert rel(delta, g["param_delta_norms"]) < 1e-3
    heads = np.concatenate([sub(new[k], 8) for k in names])
    assert rel(heads, g["param_heads"]) < 1e-6


@pytest.mark.parametrize("name,B,H,Wd", [("s256", 1, 256, 256), ("s24", 2, 24, 24)])
def test_swin_oracle_matches_golden_full_size_and_resized_skip(golden, name, B, H, Wd):
    """swin2.npz: the no-padding 256-px geometry of BASELINE config 3, and a 24-px input whose decoder resizes
    the projected skips (swin_unet_model.py:526-529); float64 as in the generator."""
    g = golden("swin2")
    cfg = swin_oracle.DEFAULT_CONFIG
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    if name == "s24":
        o, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
        assert tuple(o.shape) == tuple(g["s24.out_shape"]) == (2, 1, 16, 16)
        assert rel(sub(o, 16384), g["s24.eval_out"]) < 1e-5
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
              if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    out, stats = swin_oracle.swin_unet_forward(full, x.double(), cfg, True, None)
    tgt = torch.nn.functional.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False)
    loss = ((out - tgt) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - g[f"{name}.loss"]) < 1e-9
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-10
    names = [str(s) for s in g[f"{name}.param_names"]]
    norms = np.array([params[k].grad.norm().item() for k in names])
    keep = np.array([k != "final_conv.0.bias" for k in names])
    assert rel(norms[keep], g[f"{name}.grad_norms"][keep]) < 1e-8


@pytest.mark.parametrize("name", ["w4h16", "w7h32", "w8h64"])
def test_swin_oracle_other_geometries_match_golden(golden, name):
    """swin_geom.npz: the reference with window sides / head dimensions other than 8 / 32 (frozen by
    make_golden.py gen_swin_geom: eval in float32, train forward + backward in float64)."""
    from tests.test_gpu_swin import GEOM_CASES
    g = golden("swin_geom")
    over, (B, H, Wd) = GEOM_CASES[name]
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    cfg.update(over)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    o, _ = swin_oracle.swin_unet_forward(sd, x, cfg, False)
    assert tuple(o.shape) == tuple(g[f"{name}.out_shape"])
    assert rel(sub(o, 16384), g[f"{name}.eval_out"]) < 1e-5
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
              if v.is_floating_point() and "running_" not in k}
    full = dict(sd64)
    full.update(params)
    out, _ = swin_oracle.swin_unet_forward(full, x.double(), cfg, True, None)
    tgt = torch.nn.functional.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False)
    loss = ((out - tgt) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - g[f"{name}.loss"]) < 1e-9
    assert rel(sub(out, 16384), g[f"{name}.train_out"]) < 1e-10
    names = [str(s) for s in g[f"{name}.param_names"]]
    norms = np.array([params[k].grad.norm().item() for k in names])
    keep = np.array([k != "final_conv.0.bias" for k in names])
    assert rel(norms[keep], g[f"{name}.grad_norms"][keep]) < 1e-8


def test_swin_training_step_oracle_matches_golden(golden):
    """The Swin trainer's step (swin_unet_model.py:809-823) restated: resize to the label, 'tid_gradient' loss,
    clip, AdamW(1e-4, 0.05), against the step frozen from the reference's own modules and torch.optim.AdamW."""
    from oracle import step_oracle
    g = golden("step_swin")
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, w = W.synthetic_batch(2, 4, 64, 64, seed=42)
    lcfg = loss_oracle.preset("weighted_mse_grad", grad_w=0.3, spec_w=0.0)
    parts, norm, new, grads = step_oracle.swin_training_step(sd, x, y, w, cfg, lcfg)
    got = np.array([parts[k] for k in ("pixel", "gradient", "spectral", "total")])
    assert rel(got, g["parts"]) < 1e-5
    assert abs(norm - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    names = [str(s) for s in g["param_names"]]
    keep = np.array([k != "final_conv.0.bias" for k in names])
    gn = np.array([grads[k].double().norm().item() for k in names])
    assert rel(gn[keep], g["grad_norms"][keep]) < 5e-3
    delta = np.array([(new[k] - sd[k]).double().norm().item() for k in names])
    assert rel(delta[keep], g["param_delta_norms"][keep]) < 5e-3
    heads = np.concatenate([sub(new[k], 8) for k in names])
    assert rel(heads, g["param_heads"]) < 1e-4      # includes final_conv.0.bias: zero gradient, its Adam step is sign(noise)


def test_bf16_fixture_is_the_reference_autocast_run(golden):
    """bf16.npz holds the distance of the reference's bf16-autocast run from its fp32 run; the oracle's CPU model
    of bf16 storage (every stored tensor class rounded) must land in the same place, which is what lets the
    GPU tests bound the bf16 kernels by these numbers."""
    from oracle import unet_oracle
    g = golden("bf16")
    for name, (cin, f, B, H, Wd) in {"small": (4, 8, 2, 32, 32), "odd": (4, 16, 1, 50, 36)}.items():
        sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
        x, y, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
        lf = lambda o: ((o - y) ** 2).mean()
        o32, _, g32, _ = unet_oracle.unet_fwd_bwd(sd, x, lf)
        o16, _, g16, _ = unet_oracle.unet_fwd_bwd(sd, x, lf, None, unet_oracle.BF16_ALL)
        names = [k for k in g32 if k != "__input__"]
        a = torch.cat([g16[k].double().reshape(-1) for k in names])
        b = torch.cat([g32[k].double().reshape(-1) for k in names])
        out_rel, grad_rel = rel(o16, o32), rel(a, b)
        ref_out, ref_grad = float(g[f"unet.{name}.out_rel"]), float(g[f"unet.{name}.grad_rel"])
        assert 0.5 * ref_out < out_rel < 1.5 * ref_out, (name, out_rel, ref_out)
        assert 0.5 * ref_grad < grad_rel < 1.5 * ref_grad, (name, grad_rel, ref_grad)
    assert 0.015 < float(g["swin.s256.eval_rel"]) < 0.025 and float(g["swin.s64.grad_cos"]) > 0.97


PATCH_CASES = [("p16", 96, 112, 16, 8, 0.25, 21), ("p32", 160, 128, 32, 16, 0.25, 22), ("p8", 40, 33, 8, 3, 0.5, 23)]


class _DA:
    """Duck-typed xarray.DataArray: .lat, .lon, .isel(lat=, lon=).values"""
    def __init__(self, a):
        self.a = a
        self.lat, self.lon = np.arange(a.shape[0]), np.arange(a.shape[1])

    def isel(self, lat, lon):
        import types
        return types.SimpleNamespace(values=self.a[lat, lon])


class _DSet(dict):
    def __init__(self, arrays):
        super().__init__({k: _DA(v) for k, v in arrays.items()})


@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
def test_patch_oracle_matches_golden(golden, case):
    name, lat, lon, P, S, mv, seed = case
    g = golden("patch")
    feats, labels, weights = patch_oracle.synthetic_fields(lat, lon, seed)
    got = patch_oracle.extract_patches(feats, labels, weights, P, S, mv)
    assert np.array_equal(np.array([(p["lat_start"], p["lon_start"]) for p in got], np.int32), g[name + ".pos"])
    assert 0 < len(got) < int(g[name + ".ncand"])
    sums = np.array([[a.astype(np.float64).sum() for a in patch_oracle.dataset_item(p)] for p in got])
    assert np.array_equal(sums, g[name + ".sums"])


FILTER_CASES = [("small", 40, 56, 2.5, 3.25, 0.05, 3), ("wide_sigma", 48, 300, 1.0, 20.0, 0.0, 4),
                ("short_axis", 7, 90, 6.0, 2.0, 0.1, 5)]


@pytest.mark.parametrize("case", FILTER_CASES, ids=[c[0] for c in FILTER_CASES])
def test_filter_oracle_matches_golden_and_scipy(golden, case):
    """Gaussian low-pass (data_preparation.py:392-414): the restatement against the fixture frozen from the
    reference's method, and against the reference's dependency (scipy.ndimage) called directly."""
    from oracle import filter_oracle
    from scipy.ndimage import gaussian_filter
    name, lat, lon, s_lat, s_lon, nan_frac, seed = case
    x = filter_oracle.synthetic_field(lat, lon, nan_frac, seed)
    got = filter_oracle.apply_gaussian_filter(x, s_lat, s_lon)
    gold = golden("filter")[name]
    assert np.array_equal(np.isnan(got), np.isnan(gold))
    scale = np.nanmax(np.abs(gold))
    assert np.nanmax(np.abs(got - gold)) <= 1e-13 * scale
    direct = gaussian_filter(np.nan_to_num(x, nan=0.0), sigma=[s_lat, s_lon], mode="reflect")
    m = ~np.isnan(x)
    assert np.max(np.abs(direct[m] - got[m])) <= 1e-13 * scale


@pytest.mark.parametrize("name,seed", [("small", 1234), ("ragged", 77), ("f48", 5)])
def test_unet_oracle_backward_with_seeded_dropout_matches_golden(golden, name, seed):
    """masked.npz: the reference in float64, train mode, with the Dropout2d factors of a seeded run."""
    g = golden("masked")
    pre = f"unet.{name}"
    cin, f, B, H, Wd = UNET_CASES[name]
    sd = W.fill_state_dict(W.unet_template(cin, 1, f), seed=42)
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    x, y, _ = W.synthetic_batch(B, cin, H, Wd, seed=42)
    torch.manual_seed(seed)
    masks = [m.double() for m in unet_oracle.draw_dropout_masks(B, f)]
    assert [int((m == 0).sum()) for m in masks] == g[f"{pre}.dropped"].tolist()
    out, loss, grads, stats = unet_oracle.unet_fwd_bwd(sd, x.double(), lambda o: ((o - y.double()) ** 2).mean(), masks)
    assert rel(sub(out, 16384), g[f"{pre}.train_out"]) < 1e-12
    assert abs(loss.item() - g[f"{pre}.loss"]) < 1e-12
    names = [str(s) for s in g[f"{pre}.param_names"]]
    assert rel(np.array([grads[k].norm().item() for k in names]), g[f"{pre}.grad_norms"]) < 1e-9
    assert rel(np.concatenate([sub(grads[k], 16) for k in names]), g[f"{pre}.grad_heads"]) < 1e-9
    assert rel(sub(grads["__input__"], 4096), g[f"{pre}.grad_input"]) < 1e-9


@pytest.mark.parametrize("name,B,H,Wd,seed", [("s64", 2, 64, 64, 16), ("s24", 2, 24, 24, 3)])
def test_swin_oracle_backward_with_seeded_droppath_matches_golden(golden, name, B, H, Wd, seed):
    g = golden("masked")
    pre = f"swin.{name}"
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    x, y, _ = W.synthetic_batch(B, 4, H, Wd, seed=42)
    torch.manual_seed(seed)
    dp = [[None if k is None else k.double() for k in pair] for pair in swin_oracle.draw_drop_path(cfg, B)]
    assert [0 if k is None else int((k == 0).sum()) for pair in dp for k in pair] == g[f"{pre}.dropped"].tolist()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd)
    full.update(params)
    xo = x.double().requires_grad_(True)
    out, _ = swin_oracle.swin_unet_forward(full, xo, cfg, True, dp)
    tgt = torch.nn.functional.interpolate(y.double(), size=out.shape[-2:], mode="bilinear", align_corners=False)
    loss = ((out - tgt) ** 2).mean()
    loss.backward()
    assert rel(sub(out, 16384), g[f"{pre}.train_out"]) < 1e-12
    assert abs(loss.item() - g[f"{pre}.loss"]) < 1e-12
    names = [str(s) for s in g[f"{pre}.param_names"]]
    keep = np.array([k != "final_conv.0.bias" for k in names])
    norms = np.array([params[k].grad.norm().item() for k in names])
    assert rel(norms[keep], g[f"{pre}.grad_norms"][keep]) < 1e-9
    assert rel(sub(xo.grad, 4096), g[f"{pre}.grad_input"]) < 1e-9
