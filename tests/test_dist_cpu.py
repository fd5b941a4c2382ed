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
    red = GradReducer(m, bucket_bytes=4096)      # several buckets
    assert len(red.buckets) > 1
    for p in m.parameters():                     # gradients set by hand: nothing was launched during a backward
        p.grad = torch.randn(p.shape, generator=g)
    red.allreduce()
    if rank == 0:
        # numpy arrays travel by value (tensors travel as file descriptors of a process that may be gone)
        q.put(([p.grad.numpy().copy() for p in m.parameters()], [p.detach().numpy().copy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, port, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_grad_reducer_averages_per_shard_gradients():
    world = 2
    grads, params = _spawn(_reduce_worker, world, 31500 + os.getpid() % 2000)
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    for a, b in zip(params, m.parameters()):
        assert torch.equal(torch.from_numpy(a), b.detach())           # broadcast from rank 0
    shapes = [p.shape for p in m.parameters()]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    per_rank = [[torch.randn(s, generator=g) for s in shapes] for g in gens]
    for i in range(len(shapes)):
        want = (per_rank[0][i] + per_rank[1][i]) / world
        assert torch.allclose(torch.from_numpy(grads[i]), want, rtol=1e-6, atol=1e-7)


class _StandIn(torch.nn.Module):
    """BatchNorm-bearing stand-in with the UNet's ingredients (conv, batch-statistic BatchNorm, ReLU, pooling,
    a skip concat) small enough for the CPU; plain torch autograd, so gradients reach the reducer through
    its per-parameter hooks while the backward pass is still running."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        self.enc = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.mid = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU())
        self.dec = nn.Sequential(nn.Conv2d(24, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.head = nn.Conv2d(8, 1, 1)

    def forward(self, x):
        e = self.enc(x)
        m = self.mid(torch.nn.functional.max_pool2d(e, 2))
        u = torch.nn.functional.interpolate(m, size=e.shape[-2:], mode="bilinear", align_corners=False)
        return self.head(self.dec(torch.cat((u, e), 1)))


def _standin_data(rank):
    g = torch.Generator().manual_seed(500 + rank)
    x = torch.randn((3, 3, 12, 10), generator=g) * (1.0 + rank)        # shards with different statistics
    y = torch.randn((3, 1, 12, 10), generator=g)
    w = (torch.rand((3, 1, 12, 10), generator=g) > 0.3 + 0.2 * rank).float()   # count(w > 0) differs per shard
    return x, y, w


def _shard_step(m, rank):
    from oracle import loss_oracle
    x, y, w = _standin_data(rank)
    loss, _ = loss_oracle.multi_loss(loss_oracle.preset("weighted_mse_grad"), m(x), y, w)
    loss.backward()


def _dp_contract_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dist import GradReducer, broadcast_parameters
    torch.manual_seed(1234 + rank)
    m = _StandIn().train()
    broadcast_parameters(m)
    red = GradReducer(m, bucket_bytes=2048)
    assert len(red.buckets) > 2
    launched = []
    for step in range(2):                       # the buckets are persistent: a second step re-uses them
        m.zero_grad(set_to_none=True)
        _shard_step(m, rank)
        launched.append(sum(b.work is not None for b in red.buckets))      # collectives started inside backward
        red.finish()
    out = ([p.grad.numpy().copy() for p in m.parameters()],
           {k: v.numpy().copy() for k, v in m.state_dict().items() if "running" in k}, launched)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_dp_step_equals_mean_of_per_shard_reference_gradients(world):
    """The data-parallel contract on a BatchNorm-bearing model: after an N-rank step every rank holds the MEAN
    of the N per-shard single-process gradients; BatchNorm statistics and the weighted-loss normalisation stay
    per shard (so this differs from one big-batch step)."""
    gathered = _spawn(_dp_contract_worker, world, 33500 + (os.getpid() + world) % 2000)
    # single-process references: the same initial weights (rank 0's), each shard alone, two steps
    torch.manual_seed(1234)
    init = _StandIn().state_dict()
    per_shard, stats = [], []
    for r in range(world):
        m = _StandIn().train()
        m.load_state_dict(init)
        for step in range(2):
            m.zero_grad(set_to_none=True)
            _shard_step(m, r)
        per_shard.append([p.grad.clone() for p in m.parameters()])
        stats.append({k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    want = [sum(gs) / world for gs in zip(*per_shard)]
    for r, (grads, run, launched) in enumerate(gathered):
        for a, b in zip(grads, want):
            assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-7), r
        for k in stats[r]:
            assert torch.equal(torch.from_numpy(run[k]), stats[r][k]), (r, k)       # un-synced BatchNorm: every rank keeps its own
        assert all(n >= len(want) // 4 for n in launched), launched   # most buckets went out during backward
    # and it is NOT the big-batch gradient (BatchNorm couples the samples of a batch)
    m = _StandIn().train()
    m.load_state_dict(init)
    from oracle import loss_oracle
    xs, ys, ws = zip(*[_standin_data(r) for r in range(world)])
    for step in range(2):
        m.zero_grad(set_to_none=True)
        loss, _ = loss_oracle.multi_loss(loss_oracle.preset("weighted_mse_grad"), m(torch.cat(xs)), torch.cat(ys),
                                         torch.cat(ws))
        loss.backward()
    big = [p.grad for p in m.parameters()]
    assert any(not torch.allclose(a, b, rtol=1e-3, atol=1e-6) for a, b in zip(big, want))


class _SinkNet(torch.nn.Module):
    """A model of the unet_amd kind on the CPU: its whole backward is ONE autograd node that hands every parameter
    gradient to ``grad_sink(name, grad)`` the moment it exists (here: two linear layers and a ReLU by hand)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(23, 31)
        self.b = torch.nn.Linear(31, 5)
        self.grad_sink = None

    def forward(self, x):
        return _SinkFn.apply(x, self, *[p for _, p in self.named_parameters()])


class _SinkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        with torch.no_grad():
            h = torch.relu(model.a(x))
            out = model.b(h)
        ctx.model, ctx.saved = model, (x, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        from unet_amd.ops import GradOut
        m = ctx.model
        x, h = ctx.saved
        grads = GradOut(m.grad_sink)
        grads["b.weight"], grads["b.bias"] = dout.t() @ h, dout.sum(0)          # last layer first, as a backward runs
        dh = (dout @ m.b.weight) * (h > 0)
        grads["a.weight"], grads["a.bias"] = dh.t() @ x, dh.sum(0)
        return (None, None) + tuple(grads[n] for n, _ in m.named_parameters())


def _sink_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dist import GradReducer, broadcast_parameters
    torch.manual_seed(77 + rank)
    m = _SinkNet()
    broadcast_parameters(m)
    red = GradReducer(m, bucket_bytes=512)           # default collective: asynchronous dist.all_reduce work handles
    assert m.grad_sink is not None and len(red.buckets) >= 3
    g = torch.Generator().manual_seed(900 + rank)
    xs = [torch.randn(6, 23, generator=g) for _ in range(3)]
    res, launched = [], []
    for step, x in enumerate(xs):
        # step 0: set_to_none=True; step 1: set_to_none=False (the .grad tensors are last step's bucket views);
        # step 2: no zero_grad at all - a .grad that aliases its bucket view is replaced, never added to itself
        if step == 0:
            m.zero_grad(set_to_none=True)
        elif step == 1:
            m.zero_grad(set_to_none=False)
        m(x).square().sum().backward()
        launched.append(sum(b.work is not None for b in red.buckets))
        red.finish()
        res.append([p.grad.numpy().copy() for p in m.parameters()])
    # a second backward before finish() is refused (gradient accumulation is not supported by the reducer)
    m.zero_grad(set_to_none=True)
    m(xs[0]).sum().backward()
    try:
        m(xs[0]).sum().backward()
        second = "no error"
    except RuntimeError as e:
        second = str(e)
    red.finish()
    gathered = [None] * world
    dist.all_gather_object(gathered, (res, launched, second))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sink_path_with_default_collective():
    """The sink path (one autograd node per model, gradients handed over inside the backward) with the DEFAULT
    collective - asynchronous ``dist.all_reduce`` work handles launched from the autograd thread, ``finish()``
    waits - over several buckets and three steps with the three zero_grad habits."""
    world = 2
    gathered = _spawn(_sink_worker, world, 35500 + os.getpid() % 2000)
    torch.manual_seed(77)
    init = _SinkNet().state_dict()
    want = []
    for step in range(3):
        per = []
        for r in range(world):
            m = _SinkNet()
            m.load_state_dict(init)
            g = torch.Generator().manual_seed(900 + r)
            xs = [torch.randn(6, 23, generator=g) for _ in range(3)]
            ref = torch.nn.Sequential(m.a, torch.nn.ReLU(), m.b)
            ref(xs[step]).square().sum().backward()
            per.append([p.grad.clone() for p in m.parameters()])
        want.append([sum(gs) / world for gs in zip(*per)])
    for r, (res, launched, second) in enumerate(gathered):
        for step in range(3):
            for a, b in zip(res[step], want[step]):
                assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-6), (r, step)
        assert all(n >= 2 for n in launched), launched       # every full bucket went out inside the backward
        assert "second gradient" in second, second


def test_bench_launcher_starts_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two rank processes itself (rehearsed on the CPU:
    gloo rendezvous, one reduction) and rank 0 prints the one JSON line with n_gpus 2."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["UNETK_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 3.0
    sys.path.insert(0, repo)
    import bench
    envs = bench.rank_environments(4, 29999, base={})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["LOCAL_RANK"] == e["RANK"] for e in envs)
