"""GPU parity of the tiled large-area predictor: bit-exact tile indexing and blending, and the
whole pipeline against the oracle (fp32 1e-4, bf16 2e-2 relative L2)."""
import tempfile
import types

import numpy as np
import pytest
import torch

from oracle import tile_oracle, unet_oracle
from oracle import weights as W
from tests._cpu_tile_backend import toy_features

pytestmark = pytest.mark.gpu

CASES = [  # name, lat, lon, P, S, nan_block, speckle
    ("g200x168_p64_s32", 200, 168, 64, 32, (0, 70, 90, 168), 0.01),
    ("g160_p32_s2", 160, 160, 32, 2, None, 0.0),
    ("g300x260_p128_s8", 300, 260, 128, 8, None, 0.002),
]


class _Var:
    def __init__(self, a):
        self.a = a

    def isel(self, lat, lon):
        return types.SimpleNamespace(values=self.a[lat, lon])


class _DS(dict):
    """The duck type the reference reads: .lat, .lon, `var in ds`, ds[var].isel(...).values"""
    def __init__(self, arrays):
        super().__init__({k: _Var(v) for k, v in arrays.items()})
        a = next(iter(arrays.values()))
        self.lat, self.lon = np.arange(a.shape[0]), np.arange(a.shape[1])


def make_model(dt, f=8):
    from unet_amd import OptimizedUNet
    sd = W.fill_state_dict(W.unet_template(4, 1, f), seed=42)
    m = OptimizedUNet(4, 1, f, compute_dtype=dt)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


def cfg_for(P, S):
    return type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "BATCH_SIZE": 16,
                          "DEVICE_BATCH": 37})


@pytest.mark.parametrize("P", [8, 64, 128, 256])
def test_gaussian_kernel_bit_exact(golden, P):
    from unet_amd.predictor import LargeAreaPredictor
    k = LargeAreaPredictor(cfg_for(P, P // 2)).create_gaussian_weight_kernel(P)
    assert k.dtype == np.float32 and np.array_equal(k, golden("tile")[f"kernel_{P}"])


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_tile_indexing_and_blend_bit_exact(golden, case):
    from unet_amd.predictor import LargeAreaPredictor
    name, lat, lon, P, S, nb, sp = case
    feats = toy_features(lat, lon, 11, nb, sp)
    model, sd = make_model(torch.float32)
    pred = LargeAreaPredictor(cfg_for(P, S))
    pred.keep_tile_predictions = True
    out = pred.predict_large_area(model, _DS(feats))
    pos, tiles = pred.last_tile_predictions
    # tile positions, order and the all-NaN skip rule: identical to the reference's list
    assert np.array_equal(pos, golden("tile")[name + ".pos"])
    # given the same per-tile predictions, the blended map is bit-identical to numpy's sequential +=
    want = tile_oracle.blend_only(tiles, [tuple(p) for p in pos], lat, lon, P)
    assert out.dtype == np.float32 and np.array_equal(out, want)
    # strips no tile covers stay exactly 0
    cov_lat = (lat - P) // S * S + P
    assert not out[cov_lat:].any()


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_predict_large_area_vs_oracle(dt, tol):
    from unet_amd.predictor import LargeAreaPredictor
    name, lat, lon, P, S, nb, sp = CASES[0]
    feats = toy_features(lat, lon, 11, nb, sp)
    model, sd = make_model(dt)
    out = LargeAreaPredictor(cfg_for(P, S)).predict_large_area(model, _DS(feats))
    fn = lambda b: unet_oracle.unet_forward(sd, torch.from_numpy(b), False)[0].numpy()
    want = tile_oracle.predict_large_area(fn, feats, P, S, 16)
    err = np.linalg.norm(out.astype(np.float64) - want) / np.linalg.norm(want)
    assert err < tol, err


@pytest.mark.parametrize("block", [1, 3])
@pytest.mark.parametrize("case", CASES[:2], ids=[c[0] for c in CASES[:2]])
def test_host_path_blocks_match_device_path_bitwise(golden, case, block):
    """predict_large_area (host arrays in, host array out: band rows uploaded in blocks of `block` tile rows
    through pinned memory on a copy stream, overlapped with the network passes) against predict_device on the
    resident grid and the frozen reference output positions: the same bits whatever the block size."""
    from unet_amd.predictor import LargeAreaPredictor
    name, lat, lon, P, S, nb, sp = case
    feats = toy_features(lat, lon, 11, nb, sp)
    model, sd = make_model(torch.float32)
    cfg = cfg_for(P, S)
    cfg.HOST_BLOCK_TILE_ROWS = block
    pred = LargeAreaPredictor(cfg)
    pred.keep_tile_predictions = True
    out = pred.predict_large_area(model, _DS(feats))
    pos, _ = pred.last_tile_predictions
    assert np.array_equal(pos, golden("tile")[name + ".pos"])
    whole = LargeAreaPredictor(cfg_for(P, S))
    f = torch.from_numpy(np.stack([feats[v] for v in tile_oracle.FEATURE_VARS])).cuda()
    want = whole.predict_device(model, f).cpu().numpy()
    assert np.array_equal(out, want)
    again = pred.predict_large_area(model, _DS(feats))          # the staging buffers are re-used
    assert np.array_equal(again, want)


def test_missing_variable_and_cpu_model():
    from unet_amd.predictor import LargeAreaPredictor
    feats = toy_features(80, 80, 3)
    model, _ = make_model(torch.float32)
    p = LargeAreaPredictor(cfg_for(32, 16))
    del feats["VGG"]
    assert not p.predict_large_area(model, _DS(feats)).any()     # reference: no tile is valid
    with pytest.raises(RuntimeError):
        p.predict_large_area(model.cpu(), _DS(toy_features(80, 80, 3)))


def test_full_size_properties():
    """BASELINE config 4 geometry (8192^2, P=256, S=128) is too big for the CPU oracle; check
    size-independent properties on a 2048^2 grid with the real network: 225 tiles, every pixel covered,
    and for a constant-prediction the blend returns that constant (weights cancel)."""
    from unet_amd.predictor import HipTileBackend, LargeAreaPredictor
    G, P, S = 2048, 256, 128
    nti = (G - P) // S + 1
    assert nti * nti == 225 and len(tile_oracle.tile_starts(8192, 256, 128)) ** 2 == 3969
    be = HipTileBackend()
    wk = torch.from_numpy(tile_oracle.gaussian_weight_kernel(P)).cuda()
    slot = torch.arange(nti * nti, dtype=torch.int32, device="cuda")
    acc = torch.zeros((G, G), device="cuda")
    cnt = torch.zeros((G, G), device="cuda")
    ones = torch.full((nti * nti, 1, P, P), 3.25, device="cuda")
    be.accumulate(ones, wk, slot, 0, nti * nti, nti, nti, P, S, acc, cnt, 0, G)
    out = be.normalize(acc, cnt)
    assert float((out - 3.25).abs().max()) < 1e-5 and float(cnt.min()) > 0


def test_load_model_and_save_prediction(tmp_path):
    """large_area_predictor.py:83-106 and :300-311: both checkpoint layouts the reference accepts load into the
    same weights; a file that is not a checkpoint yields None; the NetCDF file reads back what was written."""
    from scipy.io import netcdf_file
    from unet_amd.predictor import LargeAreaPredictor
    cfg = cfg_for(32, 16)
    cfg.IN_CHANNELS, cfg.OUT_CHANNELS, cfg.INIT_FEATURES = 4, 1, 8
    p = LargeAreaPredictor(cfg)
    sd = W.fill_state_dict(W.unet_template(4, 1, 8), seed=42)
    torch.save(sd, tmp_path / "bare.pth")
    torch.save({"model_state_dict": sd, "epoch": 3, "val_loss": 0.5}, tmp_path / "wrapped.pth")
    (tmp_path / "junk.pth").write_bytes(b"not a checkpoint")
    a, b = p.load_model(str(tmp_path / "bare.pth")), p.load_model(str(tmp_path / "wrapped.pth"))
    assert a is not None and b is not None and not a.training
    assert next(a.parameters()).is_cuda
    for k, v in sd.items():
        assert torch.equal(a.state_dict()[k].cpu(), v) and torch.equal(b.state_dict()[k].cpu(), v)
    assert p.load_model(str(tmp_path / "junk.pth")) is None
    assert p.load_model(str(tmp_path / "absent.pth")) is None
    feats = toy_features(80, 72, 3)
    ds = _DS(feats)
    out = p.predict_large_area(a, ds)
    path = p.save_prediction(out, ds, "baseline")
    assert path.endswith("large_area_prediction_baseline.nc")
    with netcdf_file(path, "r", mmap=False) as nc:
        assert np.array_equal(nc.variables["prediction_baseline"][:], out)
        assert np.array_equal(nc.variables["lat"][:], np.arange(80)) and np.array_equal(nc.variables["lon"][:], np.arange(72))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("lat,lon,P,nvar", [(70, 96, 32, 4), (64, 100, 16, 4), (50, 61, 8, 4), (40, 48, 12, 3), (33, 64, 16, 5)])
def test_tile_gather_every_alignment(dt, lat, lon, P, nvar):
    """The batched gather (large_area_predictor.py:141-156: slice, NaN -> 0, stack) against torch indexing, bit for bit:
    tile columns that are and are not multiples of four floats, row pitches that are not, NaNs and infinities."""
    from unet_amd import ops
    from unet_amd._lib import call, ptr, dtype_code, stream
    g = torch.Generator().manual_seed(lat * lon + P)
    feats = torch.randn(nvar, lat, lon, generator=g)
    feats[torch.rand(nvar, lat, lon, generator=g) < 0.05] = float("nan")
    feats[0, 3, 5] = float("inf"); feats[nvar - 1, 7, 2] = float("-inf")
    pos = [(0, 0), (lat - P, lon - P), (1, 4), (5, 3), (2, 1), (lat - P, 8), (3, lon - P)]
    pos = [(i, j) for i, j in pos if 0 <= i <= lat - P and 0 <= j <= lon - P]
    ld = ops.pad_channels(nvar, dt)
    fd = feats.cuda()
    pd = torch.tensor(pos, dtype=torch.int32).reshape(-1).cuda()
    out = torch.full((len(pos), P, P, ld), 7.0, dtype=dt, device="cuda")
    call("unetk_tile_gather", ptr(fd), nvar, lat, lon, ptr(pd), len(pos), P, ptr(out), ld, dtype_code(dt), stream())
    want = torch.zeros((len(pos), P, P, ld), dtype=dt)
    for t, (i, j) in enumerate(pos):
        want[t, :, :, :nvar] = torch.nan_to_num(feats[:, i:i + P, j:j + P]).permute(1, 2, 0).to(dt)
    assert torch.equal(out.cpu().view(torch.uint8), want.view(torch.uint8))


def test_blend_four_pixels_per_thread_same_bits(monkeypatch):
    """The blend kernel that moves four pixels per thread against the one-pixel kernel (UNETK_BLEND_SCALAR) and against
    numpy's sequential `acc[win] += pred * w` (large_area_predictor.py:196-203): product and sum round separately."""
    from unet_amd._lib import call, ptr, stream
    P, S, lat, lon = 64, 32, 200, 168
    nti, ntj = (lat - P) // S + 1, (lon - P) // S + 1
    n = nti * ntj
    g = torch.Generator().manual_seed(1)
    pred = torch.randn(n, P, P, generator=g)
    pred[3, 5, 7] = float("nan")
    wk = torch.rand(P, P, generator=g)
    slot = torch.arange(n, dtype=torch.int32).cuda()
    pd, wd = pred.cuda(), wk.cuda()
    outs = []
    for scalar in (True, False):
        if scalar:
            monkeypatch.setenv("UNETK_BLEND_SCALAR", "1")
        else:
            monkeypatch.delenv("UNETK_BLEND_SCALAR")
        acc = torch.zeros(lat, lon, device="cuda")
        cnt = torch.zeros(lat, lon, device="cuda")
        call("unetk_blend_accumulate", ptr(pd), ptr(wd), ptr(slot), 0, n, nti, ntj, P, S, ptr(acc), ptr(cnt), lat, lon,
             0, lat, stream())
        outs.append((acc.cpu().numpy(), cnt.cpu().numpy()))
    want_a = np.zeros((lat, lon), np.float32)
    want_c = np.zeros((lat, lon), np.float32)
    pn, wn = np.nan_to_num(pred.numpy()), wk.numpy()
    for t in range(n):
        i, j = (t // ntj) * S, (t % ntj) * S
        want_a[i:i + P, j:j + P] += pn[t] * wn
        want_c[i:i + P, j:j + P] += wn
    for a, c in outs:
        assert np.array_equal(a, want_a) and np.array_equal(c, want_c)
