"""Sliding-window large-area prediction with Gaussian-weighted blending, on the GPU.

Drop-in for the reference ``LargeAreaPredictor`` hot path (large_area_predictor.py:57-81, 133-219):
``create_gaussian_weight_kernel(P)`` and ``predict_large_area(model, features)``.  The features are
uploaded once; tile validity, the batched tile gather, the network and the blend all run on the
device.  Tile order, NaN handling, the fp32 accumulation order and the final division reproduce
the reference bit for bit given the same per-tile predictions.

With ``torch.distributed`` initialised the tile rows are split into contiguous bands, one per
rank; a rank receives the partially accumulated overlap strip from the rank above it, adds its own
tiles in order and passes its own bottom strip on, which keeps the single-process order (and bits).
"""
import logging
import os
import pickle
import re
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import call, ptr, stream, dtype_code

FEATURE_VARS = ("GA", "DOV_EW", "DOV_NS", "VGG")    # large_area_predictor.py:161
logger = logging.getLogger("unet_amd.predictor")


# ------------------------------------------------------------------ checkpoints without execution
class InertObject:
    """Stands in for an instance of a class a checkpoint pickled (the trainer's config object,
    enhanced_training_multi_loss.py:964).  The unpickler creates it without calling anything and
    stores the pickled attributes; using the class as a function (a REDUCE of an arbitrary callable)
    is refused, which fails the load."""

    def __init__(self, *args, **kwargs):
        raise pickle.UnpicklingError(f"checkpoint calls the global {type(self).__module__}."
                                     f"{type(self).__qualname__}: refused (only class instances are tolerated)")

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["state"] = state


class InertValue:
    """Stands in for the numpy reconstructors behind a pickled numpy scalar, dtype or array
    (the training history holds np.float64 correlations, :889, :925): calling them yields an inert
    placeholder instead of running numpy code on file contents."""

    def __init__(self, *args, **kwargs):
        self.args = args

    def __setstate__(self, state):
        self.state = state


_NUMPY_VALUE_GLOBALS = (
    "numpy.core.multiarray.scalar", "numpy._core.multiarray.scalar", "numpy.dtype",
    "numpy.core.multiarray._reconstruct", "numpy._core.multiarray._reconstruct", "numpy.ndarray",
)
_UNSUPPORTED_GLOBAL = re.compile(r"Unsupported global: GLOBAL (\S+) was not an allowed global")


def _inert_stub(qualified: str):
    module, _, name = qualified.rpartition(".")
    base = InertValue if qualified in _NUMPY_VALUE_GLOBALS else InertObject
    return type(name, (base,), {"__module__": module, "__qualname__": name})


def load_checkpoint(path: str, max_globals: int = 64):
    """``torch.load(path, map_location='cpu', weights_only=True)`` that also reads checkpoints holding
    instances of classes this process cannot (and must not) import.  Every global the weights-only
    unpickler rejects is registered as an inert stand-in under its pickled qualified name and the load
    is retried: class instances become attribute bags, numpy scalars placeholders, and a pickle that
    CALLS an unknown global (os.system and the like) is refused.  No code from the file or from the
    modules it names is imported or run."""
    stubs = {}
    while True:
        try:
            with torch.serialization.safe_globals(list(stubs.values())):
                return torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            m = _UNSUPPORTED_GLOBAL.search(str(e))
            if m is None or m.group(1) in stubs or len(stubs) >= max_globals:
                raise
            stubs[m.group(1)] = _inert_stub(m.group(1))


class LargeAreaPredictionConfig:
    """Prediction parameters (large_area_predictor.py:26-52); paths are the caller's business."""
    PATCH_SIZE = 128
    STRIDE = PATCH_SIZE // 16
    BATCH_SIZE = 16
    IN_CHANNELS = 4
    OUT_CHANNELS = 1
    INIT_FEATURES = 48
    OUTPUT_DIR = "./large_area_predictions"


def band_partition(nti: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous tile-row bands [lo, hi) per rank, sizes differing by at most one row."""
    base, extra = divmod(nti, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _feature_array(features, var, lat, lon):
    da = features[var]
    if hasattr(da, "values") and not callable(getattr(da, "values")):
        return np.asarray(da.values, dtype=np.float32)
    return np.asarray(da.isel(lat=slice(0, lat), lon=slice(0, lon)).values, dtype=np.float32)


class HipTileBackend:
    """The device side of the tiled predictor: four libunetk kernels and the UNet eval engine."""

    def valid_device(self, feats, P, S, nti, ntj):
        """int32 flags on the device, queued on the current stream (no host synchronisation)."""
        nvar, lat, lon = feats.shape
        v = torch.empty(nti * ntj, dtype=torch.int32, device=feats.device)
        call("unetk_tile_valid", ptr(feats), nvar, lat, lon, P, S, nti, ntj, ptr(v), stream())
        return v

    def valid(self, feats, P, S, nti, ntj):
        return self.valid_device(feats, P, S, nti, ntj).cpu().numpy().astype(bool)

    def make_engine(self, model):
        from .unet import EvalEngine
        return EvalEngine(model), model.compute_dtype

    def gather(self, feats, pos_d, first, n, P, dt):
        nvar, lat, lon = feats.shape
        ld = ops.pad_channels(nvar, dt)
        batch = torch.empty((n, P, P, ld), dtype=dt, device=feats.device)
        call("unetk_tile_gather", ptr(feats), nvar, lat, lon, ptr(pos_d[first:]), n, P, ptr(batch), ld,
             dtype_code(dt), stream())
        return batch

    def accumulate(self, out, wk, slot, t0, t1, nti, ntj, P, S, acc, cnt, r0, r1):
        lat, lon = acc.shape
        call("unetk_blend_accumulate", ptr(out), ptr(wk), ptr(slot), t0, t1, nti, ntj, P, S, ptr(acc), ptr(cnt),
             lat, lon, r0, r1, stream())

    def normalize(self, acc, cnt):
        out = torch.empty_like(acc)
        call("unetk_blend_normalize", ptr(acc), ptr(cnt), ptr(out), acc.numel(), stream())
        return out


class LargeAreaPredictor:
    def __init__(self, config):
        self.config = config
        os.makedirs(config.OUTPUT_DIR, exist_ok=True)      # large_area_predictor.py:59
        # tiles per network launch; any value gives identical results (eval mode is per-sample)
        self.device_batch = max(int(getattr(config, "DEVICE_BATCH", 64)), 1)
        self.backend = HipTileBackend()
        # debugging / test hook: when True the per-tile predictions of the last call are kept as
        # (positions (n,2) int32, predictions (n,P,P) fp32 on the host)
        self.keep_tile_predictions = False
        self.last_tile_predictions = None

    def create_gaussian_weight_kernel(self, patch_size: int) -> np.ndarray:
        """float64 math on the host, then float32, exactly as large_area_predictor.py:71-81."""
        y, x = np.ogrid[:patch_size, :patch_size]
        center = patch_size / 2.0
        sigma = center * 0.5
        if sigma <= 0:
            sigma = 1.0
        k = np.exp(-0.5 * ((x - (center - 0.5)) ** 2 + (y - (center - 0.5)) ** 2) / sigma ** 2)
        return k.astype(np.float32)

    # ------------------------------------------------------------------ checkpoints and results
    def load_model(self, model_path: str, device: Optional[torch.device] = None, compute_dtype=torch.bfloat16):
        """large_area_predictor.py:83-106: build ``OptimizedUNet(IN_CHANNELS, OUT_CHANNELS, INIT_FEATURES)``,
        load ``checkpoint['model_state_dict']`` (or the bare state dict), eval mode; ``None`` on any failure,
        as the reference (the reason is logged).  The file is read by ``load_checkpoint``: nothing in it is
        executed, and the trainer's own checkpoints, which pickle a config class instance and numpy scalars
        next to the tensors (enhanced_training_multi_loss.py:956-971), load."""
        from .unet import OptimizedUNet
        try:
            ckpt = load_checkpoint(model_path)
            model = OptimizedUNet(in_channels=self.config.IN_CHANNELS, out_channels=self.config.OUT_CHANNELS,
                                  init_features=self.config.INIT_FEATURES, compute_dtype=compute_dtype)
            model.load_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt)
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            model.to(device)
            model.eval()
            return model
        except Exception as e:      # the reference swallows every failure and returns None (:104-106)
            logger.error("load_model(%s) failed: %s: %s", model_path, type(e).__name__, e)
            return None

    def save_prediction(self, prediction: np.ndarray, features, model_name: str) -> str:
        """large_area_predictor.py:300-311: ``large_area_prediction_<model_name>.nc`` in OUTPUT_DIR with the
        variable ``prediction_<model_name>`` on the (lat, lon) coordinates of ``features``.  Written as
        NetCDF-3 classic through scipy.io (xarray.open_dataarray reads it back)."""
        from scipy.io import netcdf_file
        path = os.path.join(self.config.OUTPUT_DIR, f"large_area_prediction_{model_name}.nc")
        lat = np.asarray(getattr(features.lat, "values", features.lat), dtype=np.float64)
        lon = np.asarray(getattr(features.lon, "values", features.lon), dtype=np.float64)
        pred = np.asarray(prediction, dtype=np.float32)
        if pred.shape != (len(lat), len(lon)):
            raise ValueError(f"prediction {pred.shape} does not match the coordinates ({len(lat)}, {len(lon)})")
        with netcdf_file(path, "w") as nc:
            nc.createDimension("lat", len(lat))
            nc.createDimension("lon", len(lon))
            vlat = nc.createVariable("lat", "f8", ("lat",))
            vlon = nc.createVariable("lon", "f8", ("lon",))
            vlat[:], vlon[:] = lat, lon
            v = nc.createVariable(f"prediction_{model_name}", "f4", ("lat", "lon"))
            v[:] = pred
        return path

    # ------------------------------------------------------------------ core
    def _plan(self, lat, lon):
        import torch.distributed as dist
        P, S = int(self.config.PATCH_SIZE), int(self.config.STRIDE)
        nti, ntj = (lat - P) // S + 1, (lon - P) // S + 1
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        if world > nti:
            raise ValueError(f"{world} ranks for {nti} tile rows: use at most one rank per tile row")
        return P, S, nti, ntj, world, rank, band_partition(nti, world)

    def predict_large_area(self, model, features, device: Optional[torch.device] = None) -> np.ndarray:
        """large_area_predictor.py:133-219 from host arrays to a host array.  Every rank uploads only the grid rows
        of its own band of tile rows (+ the P - S rows its last tiles reach into), in blocks of a few tile rows
        staged through pinned memory on a copy stream, so that the upload of block k+1 runs under the network
        passes of block k; the blended map comes back through pinned memory."""
        P = int(self.config.PATCH_SIZE)
        lat, lon = len(features.lat), len(features.lon)
        if device is None:
            device = next(model.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("LargeAreaPredictor runs on the GPU only (no CPU fallback)")
        if any(v not in features for v in FEATURE_VARS) or lat < P or lon < P:
            return np.zeros((lat, lon), np.float32)         # no tile is ever valid (:173-175)
        arrays = [_feature_array(features, v, lat, lon) for v in FEATURE_VARS]
        P, S, nti, ntj, world, rank, bands = self._plan(lat, lon)
        be = self.backend
        model.eval()
        engine, dt = be.make_engine(model)
        wk = torch.from_numpy(self.create_gaussian_weight_kernel(P)).to(device)
        lo_row, hi_row = bands[rank]
        g0, g1 = lo_row * S, (hi_row - 1) * S + P            # grid rows this rank reads
        acc = torch.zeros((g1 - g0, lon), dtype=torch.float32, device=device)
        cnt = torch.zeros((g1 - g0, lon), dtype=torch.float32, device=device)
        self._recv_strip(acc, cnt, bands, rank, S, P, lat, g0)
        step = max(1, int(getattr(self.config, "HOST_BLOCK_TILE_ROWS", 8)))
        blocks = [(a, min(a + step, hi_row)) for a in range(lo_row, hi_row, step)]
        main = torch.cuda.current_stream(device)
        copy = _copy_stream(device)
        nvar = len(arrays)
        max_rows = max((b - 1) * S + P - a * S for a, b in blocks)
        stage = self._staging(nvar, max_rows, lon, device)     # two pinned blocks + two device blocks, re-used across calls
        kept = []
        pending = None
        free_events = [None, None]                # the network passes that last read device block i
        for k, blk in enumerate(blocks + [None]):          # block k is uploaded while block k-1 runs
            nxt = None
            if blk is not None:
                a, b = blk
                r0, r1 = a * S, (b - 1) * S + P
                host, devb = stage[k % 2]
                # (dense (nvar, rows, lon) views of the flat blocks: the kernels take one plane pitch of rows * lon)
                hv = host.view(-1)[:nvar * (r1 - r0) * lon].view(nvar, r1 - r0, lon)
                if free_events[k % 2] is not None:
                    free_events[k % 2].synchronize()        # pinned + device block k%2 were last used two blocks ago
                for v in range(nvar):                        # (multi-threaded copy into the pinned block)
                    hv[v].copy_(torch.from_numpy(arrays[v][r0:r1]))
                dv = devb.view(-1)[:nvar * (r1 - r0) * lon].view(nvar, r1 - r0, lon)
                with torch.cuda.stream(copy):
                    dv.copy_(hv, non_blocking=True)
                    valid_d = be.valid_device(dv, P, S, b - a, ntj)
                    valid_p = torch.empty(valid_d.shape, dtype=valid_d.dtype, pin_memory=True)
                    valid_p.copy_(valid_d, non_blocking=True)
                    ready = torch.cuda.Event()
                    ready.record(copy)
                nxt = (k, a, b, dv, valid_p, ready)
            if pending is not None:
                pk, pa, pb, pdv, pvalid, pready = pending
                pready.synchronize()                          # waits for the copy stream only
                main.wait_event(pready)
                off = pa * S - g0
                rows = (pb - 1) * S + P - pa * S
                got = self._run_rows(engine, dt, wk, pdv, acc[off:off + rows], cnt[off:off + rows],
                                     pvalid.numpy().astype(bool), pa * S)
                if got is not None:
                    kept.append(got)
                done = torch.cuda.Event()
                done.record(main)
                free_events[pk % 2] = done
            pending = nxt
        self._send_strip(acc, cnt, bands, rank, S, P, lat, g0)
        if self.keep_tile_predictions:
            self.last_tile_predictions = (np.concatenate([k[0] for k in kept]) if kept else np.zeros((0, 2), np.int32),
                                          np.concatenate([k[1] for k in kept]) if kept else np.zeros((0, P, P), np.float32))
        return self._assemble(be.normalize(acc, cnt), bands, rank, world, S, lat, lon, g0, to_host=True)

    def predict_device(self, model, feats: torch.Tensor) -> torch.Tensor:
        """feats: (4, lat, lon) fp32 on the device -> blended (lat, lon) fp32 map on every rank.  With several
        ranks each one copies its band's rows out of the grid and works on that copy alone."""
        be = self.backend
        nvar, lat, lon = feats.shape
        dev = feats.device
        P, S, nti, ntj, world, rank, bands = self._plan(lat, lon)
        model.eval()
        engine, dt = be.make_engine(model)
        wk = torch.from_numpy(self.create_gaussian_weight_kernel(P)).to(dev)
        lo_row, hi_row = bands[rank]
        g0, g1 = lo_row * S, (hi_row - 1) * S + P
        if world == 1:
            band, g0, g1 = feats, 0, lat                     # (the rows below the last tile stay zero)
        else:
            band = feats[:, g0:g1].contiguous()
        acc = torch.zeros((g1 - g0, lon), dtype=torch.float32, device=dev)
        cnt = torch.zeros((g1 - g0, lon), dtype=torch.float32, device=dev)
        self._recv_strip(acc, cnt, bands, rank, S, P, lat, g0)
        kept = self._run_rows(engine, dt, wk, band, acc, cnt, None, g0)
        self._send_strip(acc, cnt, bands, rank, S, P, lat, g0)
        if self.keep_tile_predictions:
            self.last_tile_predictions = kept
        return self._assemble(be.normalize(acc, cnt), bands, rank, world, S, lat, lon, g0, to_host=False)

    def _run_rows(self, engine, dt, wk, feats, acc, cnt, valid_h, grid_row0):
        """The reference's loop on one block of whole tile rows: feats (nvar, rows, lon) on the device; every valid
        tile of the block is gathered, run through the network and ADDED to acc / cnt (rows, lon) in the reference's
        order (lat-major).  acc / cnt may already hold the partial sums of the tiles above the block.  Returns the
        kept tile predictions (grid positions, predictions) when asked for."""
        be = self.backend
        P, S = int(self.config.PATCH_SIZE), int(self.config.STRIDE)
        nvar, rows, lon = feats.shape
        nti, ntj = (rows - P) // S + 1, (lon - P) // S + 1
        if valid_h is None:
            valid_h = be.valid(feats, P, S, nti, ntj)
        dev = feats.device
        slot_h = np.full(nti * ntj, -1, np.int32)
        slot_h[valid_h] = np.arange(int(valid_h.sum()), dtype=np.int32)
        tiles = np.flatnonzero(valid_h)                      # lat-major, as the reference loops
        if len(tiles) == 0:
            return (np.zeros((0, 2), np.int32), np.zeros((0, P, P), np.float32)) if self.keep_tile_predictions else None
        pos_all = np.stack([(tiles // ntj) * S, (tiles % ntj) * S], 1).astype(np.int32)
        both = torch.from_numpy(np.concatenate([slot_h, pos_all.reshape(-1)]))
        if dev.type == "cuda":
            both = both.pin_memory().to(dev, non_blocking=True)
        slot, pos_d = both[:nti * ntj], both[nti * ntj:].view(-1, 2)
        # the network on the block's tiles: the expensive part, independent of the blend order.  Equal-sized
        # launches (at most device_batch tiles each): no short last batch that would run the whole network for a
        # handful of tiles
        preds = []
        nlaunch = max(1, -(-len(tiles) // self.device_batch))
        per_launch = max(1, -(-len(tiles) // nlaunch))
        for s in range(0, len(tiles), per_launch):
            n = min(per_launch, len(tiles) - s)
            out = engine(be.gather(feats, pos_d, s, n, P, dt))
            if out.shape[1] != 1:
                out = out[:, :1].contiguous()
            preds.append((s, s + n, out))
        for t0, t1, out in preds:
            r0 = int(pos_all[t0, 0])
            r1 = min(int(pos_all[t1 - 1, 0]) + P, rows)
            be.accumulate(out, wk, slot, t0, t1, nti, ntj, P, S, acc, cnt, r0, r1)
        if self.keep_tile_predictions:
            gpos = pos_all.copy()
            gpos[:, 0] += grid_row0
            return gpos, np.concatenate([o[:, 0].float().cpu().numpy() for _, _, o in preds])
        return None

    @staticmethod
    def _recv_strip(acc, cnt, bands, rank, S, P, lat, g0):
        """Rows at the top of this band already hold tiles of the rank above: take its partial sums first."""
        import torch.distributed as dist
        if rank == 0:
            return
        strip_end = min((bands[rank - 1][1] - 1) * S + P, lat)
        if strip_end > g0:
            buf = torch.empty((2, strip_end - g0, acc.shape[1]), dtype=torch.float32, device=acc.device)
            dist.recv(buf, src=rank - 1)
            acc[:strip_end - g0] = buf[0]
            cnt[:strip_end - g0] = buf[1]

    @staticmethod
    def _send_strip(acc, cnt, bands, rank, S, P, lat, g0):
        import torch.distributed as dist
        if rank >= len(bands) - 1:
            return
        hi_row = bands[rank][1]
        nxt_row0 = hi_row * S
        send_end = min((hi_row - 1) * S + P, lat)
        if send_end > nxt_row0:
            dist.send(torch.stack([acc[nxt_row0 - g0:send_end - g0], cnt[nxt_row0 - g0:send_end - g0]]).contiguous(),
                      dst=rank + 1)

    def _assemble(self, band_map, bands, rank, world, S, lat, lon, g0, to_host):
        """band_map: the normalised rows [g0, ...) of this rank.  Rank r owns the rows from its band's first row to
        the next band's first row (the last rank: to the end of the grid, rows no tile covers are zero); the map
        is assembled by an all-gather of the owned rows (padded to the tallest band)."""
        import torch.distributed as dist
        dev = band_map.device
        if world == 1:
            if band_map.shape[0] == lat:
                full = band_map
            else:
                full = torch.zeros((lat, lon), dtype=torch.float32, device=dev)
                full[g0:g0 + band_map.shape[0]] = band_map
        else:
            owned = [(bands[r][0] * S if r > 0 else 0, bands[r + 1][0] * S if r < world - 1 else lat) for r in range(world)]
            tall = max(b - a for a, b in owned)
            own0, own1 = owned[rank]
            mine = torch.zeros((tall, lon), dtype=torch.float32, device=dev)
            n = min(own1, g0 + band_map.shape[0]) - own0
            mine[:n] = band_map[own0 - g0:own0 - g0 + n]
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            full = torch.empty((lat, lon), dtype=torch.float32, device=dev)
            for (a, b), part in zip(owned, parts):
                full[a:b] = part[:b - a]
        if not to_host:
            return full
        out = torch.empty((lat, lon), dtype=torch.float32, pin_memory=True)
        out.copy_(full, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return out.numpy()

    def _staging(self, nvar, rows, lon, device):
        key = (nvar, rows, lon, str(device))
        hit = getattr(self, "_stage", None)
        if hit is None or hit[0] != key:
            bufs = [(torch.empty((nvar, rows, lon), dtype=torch.float32, pin_memory=True),
                     torch.empty((nvar, rows, lon), dtype=torch.float32, device=device)) for _ in range(2)]
            hit = self._stage = (key, bufs)
        return hit[1]


_COPY_STREAMS = {}


def _copy_stream(device):
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = _COPY_STREAMS[key] = torch.cuda.Stream(device=key)
    return st
