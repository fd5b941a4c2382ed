"""GPU test of the data-parallel path with the real HIP models: two ranks (both on the one GPU of the box,
gloo as the collective backend) run the UNet and Swin-UNet training step on different shards with
GradReducer installed as the models' gradient sink, so every bucket's all-reduce is launched from inside the
one-node backward pass.  Contract: afterwards every rank holds the MEAN of the per-shard single-process
gradients (BatchNorm statistics per shard), and a FusedAdamW step from those gradients gives identical
parameters on both ranks."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import swin_oracle
from oracle import weights as W

pytestmark = pytest.mark.gpu


def _models():
    from unet_amd import OptimizedUNet
    from unet_amd.swin import SwinUNet
    sd = W.fill_state_dict(W.unet_template(4, 1, 8), seed=42)
    u = OptimizedUNet(4, 1, 8, compute_dtype=torch.float32)
    u.load_state_dict(sd)
    u.dropout_masks_override = [None] * 9
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    s = SwinUNet(cfg, compute_dtype=torch.float32)
    s.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
    s.drop_path_override = [[None, None]] * sum(cfg["depths"])
    return {"unet": u.cuda().train(), "swin": s.cuda().train()}


def _shard(name, rank):
    x, y, _ = W.synthetic_batch(2, 4, 32 if name == "unet" else 64, 32 if name == "unet" else 64, seed=70 + rank)
    return x.cuda() * (1.0 + 0.5 * rank), y.cuda()


def _step(model, name, rank):
    x, y = _shard(name, rank)
    out = model(x)
    tgt = y if out.shape == y.shape else torch.nn.functional.avg_pool2d(y, 2)
    ((out - tgt) ** 2).mean().backward()


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:           # the parent must hear about it instead of waiting for the queue
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


def _digest(t):
    """Small stand-in for a tensor that still pins every element: head, float64 sum, norm and a byte hash."""
    import hashlib
    a = t.detach().cpu().numpy()
    return (a.reshape(-1)[:32].copy(), float(a.astype(np.float64).sum()), float(np.linalg.norm(a.astype(np.float64))),
            hashlib.sha1(a.tobytes()).hexdigest())


def _worker_body(rank, world, port, q):
    import faulthandler
    logdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    log = open(os.path.join(logdir, f"dist_rank{rank}.txt"), "w") if os.path.isdir(logdir) else None
    if log is not None:
        faulthandler.dump_traceback_later(100, exit=True, file=log)      # a stuck rank leaves its stack behind
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dist import GradReducer
    from unet_amd.optim import FusedAdamW
    out = {}
    for name, model in _models().items():
        # (gloo's CUDA-tensor collectives stall when issued from the autograd thread with both ranks on one
        # card; the sum is staged through the host here, synchronously, still from inside the backward pass.
        # On RCCL the default asynchronous all-reduce is used.)
        def host_sum(flat):
            cpu = flat.cpu()
            dist.all_reduce(cpu)
            flat.copy_(cpu)
        red = GradReducer(model, bucket_bytes=(256 << 10) if name == "unet" else (8 << 20), collective=host_sum)
        assert model.grad_sink is not None and len(red.buckets) > 1
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        for p in model.parameters():
            p.grad = None
        _step(model, name, rank)
        launched = sum(b.work is not None for b in red.buckets)      # started inside the backward pass
        red.finish()
        torch.cuda.synchronize()
        grads = {n: _digest(p.grad) for n, p in model.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        params = {n: _digest(p)[3] for n, p in model.named_parameters()}
        out[name] = (grads, params, launched, len(red.buckets))
        red.detach()
        assert model.grad_sink is None
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_mean_of_per_shard_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    gathered = None
    for _ in range(60):                        # fail as soon as a rank dies, not after a long silent wait
        try:
            gathered = q.get(timeout=2)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    if gathered is None or (isinstance(gathered, tuple) and gathered[0] == "error"):
        for p in procs:
            if p.is_alive():
                p.kill()
        pytest.fail(f"a rank failed or timed out: {gathered}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for name, model in _models().items():
        per_shard = []
        for r in range(world):
            for p in model.parameters():
                p.grad = None
            _step(model, name, r)
            torch.cuda.synchronize()
            per_shard.append({n: p.grad.detach().cpu().double() for n, p in model.named_parameters()})
        for r in range(world):
            grads, params, launched, nbuckets = gathered[r][name]
            # every bucket but the last (and, for the Swin model, the bucket of the sums taken at the end) left
            # from inside the backward pass
            assert launched >= nbuckets - 2, (name, launched, nbuckets)
            for n in per_shard[0]:
                want = ((per_shard[0][n] + per_shard[1][n]) / world).numpy()
                head, total, norm, _ = grads[n]
                scale = max(float(np.abs(want).max()), 1e-12)
                assert float(np.abs(head - want.reshape(-1)[:32]).max()) <= 1e-5 * scale + 1e-9, (name, r, n)
                assert abs(norm - float(np.linalg.norm(want))) <= 1e-5 * float(np.linalg.norm(want)) + 1e-9, (name, r, n)
                assert abs(total - float(want.sum())) <= 1e-4 * float(np.abs(want).sum()) + 1e-9, (name, r, n)
        # ... and bit for bit: the sum of two fp32 tensors and the halving are exact operations on bits that the
        # deterministic kernels reproduce, so the reduced gradient equals the one formed here from the sink-less
        # per-shard runs (the Swin backward fills the buckets from two streams and hands the LayerNorm / bias-table
        # sums over after its closing launch: a bucket sent too early would show here)
        import hashlib
        for n in per_shard[0]:
            a, b = per_shard[0][n].float(), per_shard[1][n].float()
            want_hash = hashlib.sha1(((a + b) * 0.5).numpy().tobytes()).hexdigest()
            assert gathered[0][name][0][n][3] == want_hash, (name, n)
        for n in gathered[0][name][1]:
            assert gathered[0][name][1][n] == gathered[1][name][1][n], (name, n)   # replicas stay in step, bit for bit
        for n in gathered[0][name][0]:
            assert gathered[0][name][0][n][3] == gathered[1][name][0][n][3], (name, n)   # same reduced gradient on both


def test_one_rank_rccl_group_runs_the_default_collective():
    """The reducer's DEFAULT collective - an asynchronous dist.all_reduce on the "nccl" (= RCCL) backend, issued from
    inside the backward pass on whatever stream filled the bucket - with a one-rank group on the one GPU of the box:
    every call a multi-GPU step makes into RCCL is made, and the sum over one rank must hand back the sink-less
    gradients bit for bit, for both models and two steps in a row."""
    import socket
    import torch.distributed as dist
    from unet_amd.dist import GradReducer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        for name, model in _models().items():
            want = []
            for _ in range(2):
                for p in model.parameters():
                    p.grad = None
                _step(model, name, 0)
                torch.cuda.synchronize()
                want.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
            red = GradReducer(model, bucket_bytes=(256 << 10) if name == "unet" else (8 << 20), force=True)
            assert model.grad_sink is not None and len(red.buckets) > 1
            for k in range(2):
                for p in model.parameters():
                    p.grad = None
                _step(model, name, 0)
                launched = sum(b.work is not None for b in red.buckets)
                red.finish()
                torch.cuda.synchronize()
                assert launched >= len(red.buckets) - 2, (name, launched, len(red.buckets))
                for n, p in model.named_parameters():
                    assert torch.equal(p.grad, want[k][n]), (name, k, n)
            red.detach()
            t = torch.ones(4, device="cuda")
            dist.all_reduce(t)
            dist.barrier()
            assert t.sum().item() == 4.0
    finally:
        dist.destroy_process_group()
