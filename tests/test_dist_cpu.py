"""CPU tests of the multi-rank paths with torch.distributed (gloo, world_size 2 and 3)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tile_oracle
from tests._cpu_tile_backend import NumpyTileBackend, ToyNet, toy_features

TILE_CASES = [  # lat, lon, P, S, nan_block, speckle  (same recipes as the golden fixtures)
    ("g200x168_p64_s32", 200, 168, 64, 32, (0, 70, 90, 168), 0.01),
    ("g160_p32_s2", 160, 160, 32, 2, None, 0.0),
    ("g512_p128_s64", 512, 512, 128, 64, (128, 300, 0, 200), 0.0),
]


def _predict_worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.predictor import LargeAreaPredictor
    name, lat, lon, P, S, nb, sp = case
    cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "BATCH_SIZE": 16,
                         "DEVICE_BATCH": 7})
    pred = LargeAreaPredictor(cfg)
    pred.backend = NumpyTileBackend()
    feats = toy_features(lat, lon, 11, nb, sp)
    f = torch.from_numpy(np.stack([feats[v] for v in tile_oracle.FEATURE_VARS]))
    out = pred.predict_device(ToyNet(), f).numpy()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", TILE_CASES, ids=[c[0] for c in TILE_CASES])
def test_band_sharded_predict_is_bit_exact(golden, case, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world * 7 + len(case[0])) % 2000
    procs = [ctx.Process(target=_predict_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = golden("tile")[case[0] + ".pred"]      # frozen output of the reference's own predict_large_area
    assert np.array_equal(out, want)


def test_single_process_numpy_backend_matches_golden(golden):
    """The band/strip code with world_size 1 and the CPU test double reproduces the reference."""
    from unet_amd.predictor import LargeAreaPredictor
    name, lat, lon, P, S, nb, sp = TILE_CASES[0]
    cfg = type("C", (), {"OUTPUT_DIR": tempfile.mkdtemp(), "PATCH_SIZE": P, "STRIDE": S, "BATCH_SIZE": 16})
    pred = LargeAreaPredictor(cfg)
    pred.backend = NumpyTileBackend()
    pred.keep_tile_predictions = True
    feats = toy_features(lat, lon, 11, nb, sp)
    f = torch.from_numpy(np.stack([feats[v] for v in tile_oracle.FEATURE_VARS]))
    out = pred.predict_device(ToyNet(), f).numpy()
    g = golden("tile")
    assert np.array_equal(out, g[name + ".pred"])
    assert np.array_equal(pred.last_tile_predictions[0], g[name + ".pos"])      # bit-exact tile indexing


def test_band_partition():
    from unet_amd.predictor import band_partition
    b = band_partition(63, 8)
    assert b[0] == (0, 8) and b[-1] == (56, 63) and all(b[i][1] == b[i + 1][0] for i in range(7))
    assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _reduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dist import GradReducer, broadcast_parameters
    torch.manual_seed(rank)
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    broadcast_parameters(m)
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    red = GradReducer(m, bucket_bytes=4096)      # several buckets
    assert len(red.buckets) > 1
    red.allreduce()
    if rank == 0:
        q.put(([p.grad.clone() for p in m.parameters()], [p.detach().clone() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_averages_per_shard_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, params = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    for a, b in zip(params, m.parameters()):
        assert torch.equal(a, b.detach())           # broadcast from rank 0
    shapes = [p.shape for p in m.parameters()]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    per_rank = [[torch.randn(s, generator=g) for s in shapes] for g in gens]
    for i in range(len(shapes)):
        want = (per_rank[0][i] + per_rank[1][i]) / world
        assert torch.allclose(grads[i], want, rtol=1e-6, atol=1e-7)
