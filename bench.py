#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: UNet training step (headline), Swin-UNet training step and tiled
large-area prediction.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With WORLD_SIZE unset and --gpus N > 1 this process only starts N fresh rank processes (before it makes
any GPU call) and waits for them.  Rank 0 prints ONE JSON line:

  headline  BASELINE config 2: OptimizedUNet(4,1,48) forward, the 'all' multi-loss, backward to every
            parameter (N > 1: bucketed RCCL all-reduce launched from inside the backward pass), global-norm
            clip and AdamW; batch 64 of 4x256x256 per GPU, bf16 storage with fp32 accumulation, inputs
            resident in HBM.  `roofline` is its dominant kernel (MFMA implicit-GEMM conv, 96-channel blocks):
            algorithmic FLOPs per launch / average launch duration measured with events on the launch
            stream inside the timed steps, against the 2.5 PFLOP/s dense bf16 peak.
  "swin"    BASELINE config 3 as the reference trainer runs it (swin_unet_model.py:809-823): SwinUNet
            forward, bilinear resize of the half-resolution prediction to the 256x256 label, the
            'tid_gradient' multi-loss, backward, clip, AdamW(1e-4, 0.05).
  "predict" BASELINE config 4: LargeAreaPredictor over a synthetic 8192x8192 grid, 256-px tiles, stride 128,
            UNet(4,1,48) eval, Gaussian blend; tile rows sharded over the ranks (strong scaling).
Each record carries its own `roofline` and, at N = 1, a `cpu_baseline`: the CPU oracle (torch fp32) timed on
this host on a bounded sample of the same workload.  --workload train|swin|predict runs one of them only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
LOSS_CFG = {   # training_variants.py:119-130 "all" preset
    "loss_combination": "all", "pixel_loss_weight": 1.0, "gradient_loss_weight": 0.3,
    "spectral_loss_weight": 0.2, "pixel_loss_type": "mse", "use_tid_weights": True,
    "tid_weight_scale": 1.0, "gradient_method": "sobel", "spectral_focus_high_freq": True,
    "high_freq_weight_factor": 2.0,
}
SWIN_LOSS_CFG = dict(LOSS_CFG, loss_combination="weighted_mse_grad", spectral_loss_weight=0.0)   # 'tid_gradient'


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-seconds", type=float, default=12.0,
                    help="untimed steps before the warm-up until the step time is steady (a box's first GPU process "
                         "runs slow for several seconds); 0 = none")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="all", choices=["all", "train", "swin", "predict"],
                    help="all: headline UNet step with the swin and predict records attached; or one of them alone")
    ap.add_argument("--grid", type=int, default=8192, help="predict: side of the synthetic grid")
    ap.add_argument("--swin-batch", type=int, default=64)
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the training step as one HIP graph (auto: the Swin step on one rank)")
    ap.add_argument("--no-small-batch", action="store_true", help="swin: skip the batch-6 graph/eager comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="predict: skip the host-to-host timing of predict_large_area")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------- rank launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, port, base=None):
    """Environment of each of the n rank processes (one per GPU, rendezvous on the loopback address)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        envs.append(e)
    return envs


def launch_ranks(n, argv):
    """Start n fresh rank processes of this script and wait for them.  The parent never touches the GPU
    and replaces no process image; a failing rank ends the others."""
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e)
             for e in rank_environments(n, free_port())]
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                r = p.poll()
                if r is None:
                    continue
                alive.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in alive:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---------------------------------------------------------------------------- timing helpers
# UNETK_BENCH_FORCE_DIST=1: take every multi-rank branch with a ONE-rank RCCL group (process group on "nccl",
# GradReducer with its default asynchronous all-reduce, barriers, rank maxima, the predictor's band exchange): the
# rehearsal of the --gpus N path that a box with one GPU can run.
FORCE_DIST = os.environ.get("UNETK_BENCH_FORCE_DIST") == "1"


def multi(world):
    return world > 1 or FORCE_DIST


def barrier_sync(world, gpu=True):
    if gpu:
        torch.cuda.synchronize()
    if multi(world):
        import torch.distributed as dist
        dist.barrier()
        if gpu:
            torch.cuda.synchronize()


def settle(fn, world, dev, max_seconds, batch=10):
    """Untimed steps until the step time has stopped moving (before the W warm-up steps and the timed region).
    The first GPU process on a box that has been idle runs 20-35 % slow for several seconds whatever it does
    (measured: 22-25 ms per UNet step during the first 2 s, 18.2 ms after 8 s or in any later process), so a fixed
    handful of warm-up steps measures the box's wake-up, not the kernels.  Groups of `batch` steps are run until two
    groups in a row are within 1.5 % of the group before them, for max_seconds at most; every rank takes the
    same decision (the group time is the maximum over ranks).  Returns the number of steps taken."""
    if max_seconds <= 0:
        return 0
    t_begin, prev, calm, n = time.perf_counter(), None, 0, 0
    while True:
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        n += batch
        stop = time.perf_counter() - t_begin > max_seconds
        if multi(world):
            import torch.distributed as dist
            v = torch.tensor([dt_, float(stop)], dtype=torch.float64, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            dt_, stop = float(v[0]), bool(v[1] > 0)
        calm = calm + 1 if (prev is not None and abs(dt_ - prev) <= 0.015 * prev) else 0
        prev = dt_
        if calm >= 2 or stop:
            return n


def timed(fn, steps, warmup, world, dev):
    """warmup untimed calls, then exactly `steps` calls between barriers; max over ranks, seconds."""
    for _ in range(warmup):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier_sync(world)
    el = time.perf_counter() - t0
    if multi(world):
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    return el


class ConvTimer:
    """Brackets every implicit-GEMM launch that goes through ops.conv_fwd / ops.conv_fwd_cat (3x3 convs forward
    and data gradient, and the 1x1 / linear layers) with events on the launch stream."""

    def __init__(self):
        from unet_amd import ops
        self.ops = ops
        self.orig = ops.conv_fwd
        self.orig_cat = ops.conv_fwd_cat
        self.orig_sums = ops.conv_bwd_data_bnsums
        self.records = []

    def __enter__(self):
        orig, records = self.orig, self.records

        def timed_conv(xin, pw, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(xin, pw, *a, **k)
            e1.record()
            npix = xin.numel() // xin.shape[-1]
            esz = xin.element_size()
            records.append(dict(e0=e0, e1=e1, flops=2.0 * npix * pw.Co * pw.Ci * pw.ks * pw.ks, bn=pw.bn, ks=pw.ks,
                                bytes=float(npix * (pw.Ci + pw.Co) * esz + pw.data.numel() * esz)))
            return r

        orig_cat = self.orig_cat

        def timed_conv_cat(x1, x2, pw, *a, **k):      # the decoder convs that read (resized, skip) in place
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_cat(x1, x2, pw, *a, **k)
            e1.record()
            npix = x1.numel() // x1.shape[-1]
            esz = x1.element_size()
            records.append(dict(e0=e0, e1=e1, flops=2.0 * npix * pw.Co * pw.Ci * pw.ks * pw.ks, bn=pw.bn, ks=pw.ks,
                                bytes=float(npix * (pw.Ci + pw.Co) * esz + pw.data.numel() * esz)))
            return r

        orig_sums = self.orig_sums

        def timed_conv_sums(dy, pw, *a, **k):         # the data-gradient launches that also leave BatchNorm-backward sums
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_sums(dy, pw, *a, **k)
            e1.record()
            npix = dy.numel() // dy.shape[-1]
            esz = dy.element_size()
            records.append(dict(e0=e0, e1=e1, flops=2.0 * npix * pw.Co * pw.Ci * pw.ks * pw.ks, bn=pw.bn, ks=pw.ks,
                                bytes=float(npix * (pw.Ci + 2 * pw.Co) * esz + pw.data.numel() * esz)))
            return r

        self.ops.conv_fwd = timed_conv
        self.ops.conv_fwd_cat = timed_conv_cat
        self.ops.conv_bwd_data_bnsums = timed_conv_sums
        return self

    def __exit__(self, *exc):
        self.ops.conv_fwd = self.orig
        self.ops.conv_fwd_cat = self.orig_cat
        self.ops.conv_bwd_data_bnsums = self.orig_sums

    def family(self, pred):
        rows = [(r["e0"].elapsed_time(r["e1"]), r["flops"], r["bytes"]) for r in self.records if pred(r)]
        ms = sum(t for t, _, _ in rows)
        return len(rows), ms, sum(f for _, f, _ in rows), sum(b for _, _, b in rows)


def pmc_traffic(kernel):
    """HBM bytes per launch of a kernel from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def mfma_roofline(timer, dt, steps, pred, kernel):
    peak = PEAK_BF16_TFLOPS if dt == torch.bfloat16 else PEAK_F32_TFLOPS
    n, ms, flops, _ = timer.family(pred)
    if n == 0 or ms <= 0:
        return None
    achieved = flops / (ms * 1e-3) / 1e12
    fn, fms, fflops, _ = timer.family(lambda r: r["ks"] == 3)
    return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": pmc_traffic(kernel.split(" ")[0]),
            "launches_per_step": n // steps, "avg_launch_us": round(ms * 1e3 / n, 2),
            "algorithmic_gflop_per_launch": round(flops / n / 1e9, 2),
            "conv_family": {"launches_per_step": fn // steps, "ms_per_step": round(fms / steps, 3),
                            "achieved": round(fflops / (fms * 1e-3) / 1e12, 2)} if fn else None}


def host_threads():
    """Threads for the CPU baseline: the cores this process may use (the box exports its share as
    OMP_NUM_THREADS; os.cpu_count() would report the whole machine)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        avail = min(avail, int(env))
    return max(1, avail)


def median_time(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


# ---------------------------------------------------------------------------- UNet training step
def run_train(args, rank, world, dev, dt):
    from oracle import weights as W
    from unet_amd import OptimizedUNet
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    model = OptimizedUNet(4, 1, 48, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
    model = model.to(dev).train()
    crit = MultiLossFunction(LOSS_CFG).to(dev)
    x, y, w = W.synthetic_batch(args.batch, 4, args.size, args.size, seed=42 + rank)
    x, y, w = x.to(dev), y.to(dev), w.to(dev)
    reducer = None
    if multi(world):
        from unet_amd.dist import GradReducer
        reducer = GradReducer(model, force=FORCE_DIST)
    # the trainer's optimizer tail (enhanced_training_multi_loss.py:758-763, 827-829)
    graphed = args.graph == "on" and not multi(world)
    opt = FusedAdamW(model.parameters(), lr=5e-5, weight_decay=5e-5, max_grad_norm=1.0, capturable=graphed)

    params = list(model.parameters())        # (zero_grad as the optimizer does it: over its own list, no module walk)

    def step():
        for p in params:
            p.grad = None
        loss, _ = crit(model(x), y, w)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()

    if graphed:
        from unet_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, crit, opt, x, y, w)
        step = lambda: gstep(x, y, w)
    settled = settle(step, world, dev, args.settle_seconds)
    for _ in range(args.warmup):
        step()
    # Two timed events around each of the 35 conv launches cost the step 0.15 ms when the box is warm and
    # several ms in a box's first GPU process (measured 22-25 ms per step against 18.2), so only the LAST
    # timed step carries them (as in the Swin record below).
    timer = ConvTimer() if (rank == 0 and not args.no_roofline and not graphed) else None
    calls = {"n": 0}

    def timed_step():
        calls["n"] += 1
        if timer is not None and calls["n"] == args.steps:
            with timer:
                step()
        else:
            step()

    el = timed(timed_step, args.steps, 0, world, dev)
    ms = el / args.steps * 1e3
    rec = None
    if rank == 0:
        tname = "bf16" if dt == torch.bfloat16 else "f32"
        roof = None
        if timer is not None:
            roof = mfma_roofline(timer, dt, 1, lambda r: r["bn"] == 96 and r["ks"] == 3,
                                 f"conv_igemm_kernel<{tname},6,3> (forward + data-gradient launches)")
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_train(8, args.size)
        rec = {
            "metric": "train patches/s (fwd+bwd)", "value": round(world * args.batch * args.steps / el, 2),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_steps": settled,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"UNet(4,1,48) training step fwd+loss(all)+bwd+clip+AdamW, batch {args.batch}/GPU of "
                                   f"4x{args.size}x{args.size}, {args.dtype} storage fp32 accumulate",
                       "global_batch": world * args.batch, "patch": args.size,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": roof, "cpu_baseline": cpu,
        }
    if reducer is not None:
        reducer.detach()
    del model, opt, crit, x, y, w
    torch.cuda.empty_cache()
    return rec


def cpu_baseline_train(batch, size):
    """The CPU oracle on a bounded sample: UNet(4,1,48) fwd + 'all' loss + bwd, batch `batch`."""
    from oracle import loss_oracle, unet_oracle
    from oracle import weights as W
    threads = host_threads()
    torch.set_num_threads(threads)
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=42)
    x, y, w = W.synthetic_batch(batch, 4, size, size, seed=42)
    cfg = loss_oracle.preset("all")
    t = median_time(lambda: unet_oracle.unet_fwd_bwd(sd, x, lambda o: loss_oracle.multi_loss(cfg, o, y, w)[0]))
    return {"value": round(batch / t, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU fp32 oracle, batch {batch} of 4x{size}x{size}, fwd+loss(all)+bwd, median of 5 after 2 "
                      f"warm-ups, {threads} threads (host reports {os.cpu_count()} logical CPUs)"}


# ---------------------------------------------------------------------------- Swin-UNet training step
GRAPH_MAX_BATCH = 16


def run_swin(args, rank, world, dev, dt):
    """BASELINE config 3 as swin_unet_model.py:809-823 runs it."""
    from oracle import swin_oracle
    from oracle import weights as W
    from unet_amd import ops
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    model = SwinUNet(cfg, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
    model = model.to(dev).train()
    B = args.swin_batch
    x, y, w = W.synthetic_batch(B, 4, args.size, args.size, seed=42 + rank)
    x, y, w = x.to(dev), y.to(dev), w.to(dev)
    crit = MultiLossFunction(SWIN_LOSS_CFG).to(dev)
    reducer = None
    if multi(world):
        from unet_amd.dist import GradReducer
        reducer = GradReducer(model, force=FORCE_DIST)
    # Small batches on one rank: the whole step is captured into a HIP graph and replayed (the ~600 launches of
    # a step cost the host ~13 ms, more than the GPU needs below ~32 patches; measured 6.3 ms graphed vs 14.4 ms
    # eager at the reference's batch of 6, but 16.2 vs 15.7 ms at batch 64, where the GPU is the limit and the
    # graph's per-node dispatch costs more than eager launches that run ahead).  Several ranks: eager, the bucket
    # all-reduces start from inside the backward pass.
    graphed = args.graph == "on" or (args.graph == "auto" and not multi(world) and B <= GRAPH_MAX_BATCH)
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05, max_grad_norm=1.0, capturable=graphed)

    def resize(out, labels):
        return ops.interpolate_bilinear(out, labels.shape[-2:]) if out.shape[-2:] != labels.shape[-2:] else out

    params = list(model.parameters())        # (zero_grad as the optimizer does it: over its own list, no module walk)

    def eager_step():
        for p in params:
            p.grad = None
        loss, _ = crit(resize(model(x), y), y, w)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()

    step = eager_step
    if graphed:
        if reducer is not None:
            raise SystemExit("--graph on: the captured step has no gradient all-reduce; use one rank")
        from unet_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, crit, opt, x, y, w, post=resize)
        step = lambda: gstep(x, y, w)      # copies the batch into the captured buffers, then one graph launch

    settle(step, world, dev, args.settle_seconds)
    for _ in range(args.warmup):
        step()
    # The step queues ~600 small launches: two events around each of the 117 linear launches of EVERY timed
    # step would slow the host enough to show in the step time, so only the last timed step carries them.
    # (A replayed graph has no per-launch host code to hang events on: there the events come from one
    # eager step run after the timed region.)
    timer = LinearTimer() if (rank == 0 and not args.no_roofline) else None
    calls = {"n": 0}

    def timed_step():
        calls["n"] += 1
        if timer is not None and not graphed and calls["n"] == args.steps:
            with timer:
                step()
        else:
            step()

    el = timed(timed_step, args.steps, 0, world, dev)
    if timer is not None and graphed:
        with timer:
            eager_step()
        torch.cuda.synchronize()
    ms = el / args.steps * 1e3
    rec = None
    if rank == 0:
        roof = timer.roofline(1) if timer is not None else None
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_swin(8, args.size)
        rec = {
            "metric": "train patches/s (fwd+bwd)", "value": round(world * B * args.steps / el, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SwinUNet (embed 96, depths 2-2-6-2, window 8) trainer step: fwd + bilinear x2 to the "
                                   f"label + loss(tid_gradient) + bwd + clip + AdamW(1e-4, 0.05), batch {B}/GPU of "
                                   f"4x{args.size}x{args.size}", "global_batch": world * B,
                       "launch": "one HIP graph per step" if graphed else "eager",
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": roof, "cpu_baseline": cpu}
    if reducer is not None:
        reducer.detach()
    if rec is not None and not multi(world) and not graphed and not args.no_small_batch:
        # the data-parallel form of the same step on this one GPU: gradients copied into the reducer's flat
        # buckets from both streams of the backward pass, each bucket's collective replaced by a no-op
        from unet_amd.dist import GradReducer
        reducer = GradReducer(model, collective=lambda flat: None, force=True)
        for _ in range(3):
            eager_step()
        el_dp = timed(eager_step, args.steps, 0, world, dev)
        rec["dp_path"] = {"ms_per_step": round(el_dp / args.steps * 1e3, 3), "buckets": len(reducer.buckets),
                          "what": "the same step with the data-parallel gradient sink installed (bucket fills on both "
                                  "streams of the backward pass, deferred sums in their own bucket; no-op collective)"}
        reducer.detach()
        reducer = None
    del model, opt, crit, x, y, w
    torch.cuda.empty_cache()
    if rec is not None and not multi(world) and args.graph == "auto" and B > GRAPH_MAX_BATCH and not args.no_small_batch:
        # the reference trainer's own batch size (swin_unet_model.py:728), where the step is launch-bound
        import copy
        small = copy.copy(args)
        small.swin_batch, small.no_roofline, small.no_cpu_baseline, small.steps = 6, True, True, max(args.steps, 10)
        rows = {}
        for mode in ("on", "off"):
            small.graph = mode
            r = run_swin(small, rank, world, dev, dt)
            rows["graph_ms_per_step" if mode == "on" else "eager_ms_per_step"] = r["ms_per_step"]
        rec["reference_batch"] = dict(batch=6, patches_per_s=round(6e3 / rows["graph_ms_per_step"], 1), **rows)
    return rec


RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)      # 312.5


def pmc_entry(table, kernel):
    """A row of a frozen rocprofv3 --pmc table (profiles/<table>.json, written by tools/pmc_traffic.py from separate
    counter passes of the same bench command), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", table + ".json")) as f:
            rows = json.load(f)
    except (OSError, ValueError):
        return {}
    if kernel in rows:
        return rows[kernel]
    # (template arguments added since the table was frozen: "name<2, 6>" also finds "name<2, 6, false>")
    stem = kernel[:-1] if kernel.endswith(">") else kernel
    return next((v for k, v in rows.items() if k.startswith(stem)), {})


class LinearTimer:
    """Events around every token-GEMM launch of the Swin step (the encoder's nn.Linear layers forward and data
    gradient, the decoder's ConvTranspose GEMMs forward and data gradient) and around the window-attention
    launches.  A token linear moves (Ci + Co) x 2 bytes per token against 2 x Ci x Co FLOP: stages 0/1 (96 / 192
    channels, most of the tokens) sit far below the 312 FLOP/byte ridge and are priced against HBM; stages 2/3
    and the ConvTranspose GEMMs (384+ channels) sit at or above it and are priced against the MFMA peak."""

    def __init__(self):
        from unet_amd import ops, swin
        self.swin, self.ops = swin, ops
        self.records = []          # (e0, e1, flops, bytes)
        self.attn = []             # (e0, e1, flops, bytes, "fwd" | "bwd")
        self.orig = None

    def __enter__(self):
        ops = self.ops
        self.orig = (self.swin._linear_launch, ops.convT2x2_fwd, ops.convT2x2_bwd_data, ops.window_attn_fwd,
                     ops.window_attn_bwd)
        o_lin, o_ct, o_ctb, o_af, o_ab = self.orig
        records, attn = self.records, self.attn

        def bracket(fn, price):
            def wrapped(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a, **k)
                e1.record()
                price(e0, e1, r, *a, **k)
                return r
            return wrapped

        def lin(e0, e1, r, x, *a, **k):
            out = r[0] if isinstance(r, tuple) else r
            ntok, ci, co = x.numel() // x.shape[-1], x.shape[-1], out.shape[-1]
            records.append((e0, e1, 2.0 * ntok * ci * co, float((ntok * (ci + co) + ci * co) * x.element_size())))

        def ct_fwd(e0, e1, r, x, pk, bias, cat, *a, **k):       # K -> 4*up GEMM, pixel-shuffled stores
            ntok, ci, co = x.numel() // x.shape[-1], x.shape[-1], pk.Co
            records.append((e0, e1, 2.0 * ntok * ci * co, float((ntok * (ci + co) + ci * co) * x.element_size())))

        def ct_bwd(e0, e1, r, dcat, up, pkT, h, w, *a, **k):    # 4*up -> Cin GEMM, gathered operand rows
            if r is None:
                return
            ntok, ci, co = r.numel() // r.shape[-1], 4 * up, r.shape[-1]
            records.append((e0, e1, 2.0 * ntok * ci * co, float((ntok * (ci + co) + ci * co) * r.element_size())))

        def at_fwd(e0, e1, r, qkv, table, heads, *a, **k):      # per token: read q, k, v, write the output
            ntok, C = qkv.numel() // qkv.shape[-1], qkv.shape[-1] // 3
            attn.append((e0, e1, 4.0 * 64 * 32 * ntok * heads, float(ntok * 4 * C * qkv.element_size()), "fwd"))

        def at_bwd(e0, e1, r, qkv, dout, table, heads, *a, **k):   # read qkv and d out, write d qkv; five products
            ntok, C = qkv.numel() // qkv.shape[-1], qkv.shape[-1] // 3
            attn.append((e0, e1, 10.0 * 64 * 32 * ntok * heads, float(ntok * 7 * C * qkv.element_size()), "bwd"))

        self.swin._linear_launch = bracket(o_lin, lin)
        ops.convT2x2_fwd, ops.convT2x2_bwd_data = bracket(o_ct, ct_fwd), bracket(o_ctb, ct_bwd)
        ops.window_attn_fwd, ops.window_attn_bwd = bracket(o_af, at_fwd), bracket(o_ab, at_bwd)
        return self

    def __exit__(self, *exc):
        ops = self.ops
        (self.swin._linear_launch, ops.convT2x2_fwd, ops.convT2x2_bwd_data, ops.window_attn_fwd,
         ops.window_attn_bwd) = self.orig

    @staticmethod
    def _sum(rows):
        ms = sum(r[0].elapsed_time(r[1]) for r in rows)
        return len(rows), ms, sum(r[2] for r in rows), sum(r[3] for r in rows)

    def roofline(self, steps):
        if not self.records:
            return None
        low = [r for r in self.records if r[2] / r[3] < 0.8 * RIDGE_FLOP_PER_BYTE]
        high = [r for r in self.records if r[2] / r[3] >= 0.8 * RIDGE_FLOP_PER_BYTE]
        n, ms, flops, byts = self._sum(low)
        gbs = byts / (ms * 1e-3) / 1e9
        pm = pmc_entry("pmc_traffic_swin", "void lin_gemm_kernel<2, 6>")
        out = {"bound": "hbm", "kernel": "lin_gemm_kernel: token GEMMs of Swin stages 0/1 (below 250 FLOP/byte), forward + "
                                         "data gradient",
               "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
               "traffic": pm.get("hbm_bytes_per_launch"), "traffic_note": "PMC bytes per launch averaged over ALL launches "
               "of the kernel (both groups); algorithmic bytes per launch of this group beside it",
               "launches_per_step": n // steps, "ms_per_step": round(ms / steps, 3),
               "algorithmic_mb_per_launch": round(byts / max(n, 1) / 1e6, 1),
               "mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1)}
        if high:
            n2, ms2, fl2, by2 = self._sum(high)
            tf = fl2 / (ms2 * 1e-3) / 1e12
            out["mfma_bound_members"] = {
                "bound": "mfma", "kernel": "lin_gemm_kernel: token GEMMs of stages 2/3 and the ConvTranspose GEMMs "
                                           "(at or above 250 FLOP/byte)",
                "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                "launches_per_step": n2 // steps, "ms_per_step": round(ms2 / steps, 3),
                "hbm_gbs": round(by2 / (ms2 * 1e-3) / 1e9, 1)}
        if self.attn:
            rows = {}
            for d, kern in (("fwd", "attn_fwd_bf16_kernel"), ("bwd", "attn_bwd_bf16_kernel")):
                sel = [r for r in self.attn if r[4] == d]
                if not sel:
                    continue
                n3, ms3, fl3, by3 = self._sum(sel)
                g3 = by3 / (ms3 * 1e-3) / 1e9
                pa = pmc_entry("pmc_traffic_swin", kern)
                rows[d] = {"kernel": kern, "bound": "hbm", "achieved": round(g3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(g3 / PEAK_HBM_GBS, 4), "launches_per_step": n3 // steps,
                           "ms_per_step": round(ms3 / steps, 3), "mfma_tflops": round(fl3 / (ms3 * 1e-3) / 1e12, 2),
                           "traffic": pa.get("hbm_bytes_per_launch"), "mfma_util_pmc": pa.get("mfma_util")}
            out["attention"] = rows
        return out


def cpu_baseline_swin(batch, size):
    from oracle import loss_oracle, step_oracle, swin_oracle
    from oracle import weights as W
    threads = host_threads()
    torch.set_num_threads(threads)
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    sd = W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42)
    x, y, w = W.synthetic_batch(batch, 4, size, size, seed=42)
    lcfg = loss_oracle.preset("weighted_mse_grad", grad_w=0.3, spec_w=0.0)
    t = median_time(lambda: step_oracle.swin_training_step(sd, x, y, w, cfg, lcfg), warmup=2, iters=5)
    return {"value": round(batch / t, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU fp32 oracle of the trainer step, batch {batch} of 4x{size}x{size}, median of 5 after 2 "
                      f"warm-ups, {threads} threads"}


# ---------------------------------------------------------------------------- tiled prediction
def run_predict(args, rank, world, dev, dt):
    """Tiled large-area inference (BASELINE config 4)."""
    from oracle import weights as W
    from unet_amd import OptimizedUNet
    from unet_amd.predictor import LargeAreaPredictionConfig, LargeAreaPredictor
    model = OptimizedUNet(4, 1, 48, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
    model = model.to(dev).eval()

    class Cfg(LargeAreaPredictionConfig):
        PATCH_SIZE = 256
        STRIDE = 128
        OUTPUT_DIR = "/tmp/unetk_bench_pred"
    g = torch.Generator(device="cpu").manual_seed(7)
    feats = torch.randn((4, args.grid, args.grid), generator=g).to(dev)
    pred = LargeAreaPredictor(Cfg)
    steps, warmup = max(1, min(args.steps, 3)), max(1, min(args.warmup, 1))
    with torch.no_grad():
        settle(lambda: pred.predict_device(model, feats), world, dev, args.settle_seconds, batch=2)
        for _ in range(warmup):
            pred.predict_device(model, feats)
        timer = ConvTimer() if (rank == 0 and not args.no_roofline) else None
        tiles = TileTimer(pred) if timer is not None else None
        calls = {"n": 0}

        def timed_pass():      # (the events ride on the last timed pass only: see run_train)
            calls["n"] += 1
            if timer is not None and calls["n"] == steps:
                with timer, tiles:
                    pred.predict_device(model, feats)
            else:
                pred.predict_device(model, feats)

        el = timed(timed_pass, steps, 0, world, dev)
        e2e = predict_end_to_end(pred, model, feats, world) if not args.no_e2e else None
    ms = el / steps * 1e3
    mpix = args.grid * args.grid / 1e6
    rec = None
    if rank == 0:
        tname = "bf16" if dt == torch.bfloat16 else "f32"
        roof = None
        if timer is not None:
            roof = mfma_roofline(timer, dt, 1, lambda r: r["bn"] == 96 and r["ks"] == 3,
                                 f"conv_igemm_kernel<{tname},6,3> (eval forward launches)")
            if roof is not None:
                roof["hbm_kernels"] = tiles.report(1)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_predict(feats, min(2048, args.grid))
        ntile = ((args.grid - 256) // 128 + 1) ** 2
        rec = {
            "metric": "tiled-predict Mpix/s", "value": round(mpix / (ms * 1e-3), 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"LargeAreaPredictor over a {args.grid}x{args.grid} grid, 256-px tiles, stride 128 "
                                   f"({ntile} tiles), UNet(4,1,48) eval, Gaussian blend",
                       "parallelism": f"tile-row bands over {world} GPU(s)"},
            "roofline": roof, "cpu_baseline": cpu, "e2e": e2e}
    del model, feats, pred
    torch.cuda.empty_cache()
    return rec


def predict_end_to_end(pred, model, feats, world):
    """The reference API end to end (large_area_predictor.py:133-219 with its host-device traffic, :187-216): four
    pageable host arrays in, one host array out.  Never the headline value: reported beside it."""
    import numpy as np
    from unet_amd.predictor import FEATURE_VARS

    class HostGrid(dict):
        def __init__(self, arrays):
            super().__init__(arrays)
            a = arrays[FEATURE_VARS[0]]
            self.lat, self.lon = np.arange(a.shape[0]), np.arange(a.shape[1])
    host = feats.cpu().numpy()
    grid = HostGrid({v: np.ascontiguousarray(host[i]) for i, v in enumerate(FEATURE_VARS)})
    for v in grid:
        grid[v] = type("V", (), {"values": grid[v]})()
    ts = []
    for _ in range(4):
        barrier_sync(world)
        t0 = time.perf_counter()
        out = pred.predict_large_area(model, grid)
        ts.append(time.perf_counter() - t0)
    assert out.shape == host.shape[1:]
    steady = sorted(ts[1:])[1]
    n = host.shape[1] * host.shape[2]
    return {"ms": round(steady * 1e3, 2), "Mpix_per_s": round(n / 1e6 / steady, 2), "ms_first_call": round(ts[0] * 1e3, 2),
            "includes": "predict_large_area from four pageable host arrays to a host array: band rows staged through "
                        "pinned memory in blocks of 8 tile rows on a copy stream under the network passes of the "
                        "block before, result back through pinned memory; first call also pins the staging blocks; "
                        "median of 3 calls after it"}


class TileTimer:
    """Events around the tile gather and blend launches: achieved GB/s of algorithmic bytes against HBM."""

    def __init__(self, predictor):
        self.be = predictor.backend
        self.rec = {"tile_gather": [], "blend_accumulate": []}

    def __enter__(self):
        be, rec = self.be, self.rec
        self.orig = (be.gather, be.accumulate)
        og, oa = self.orig

        def gather(feats, pos_d, first, n, P, dt):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = og(feats, pos_d, first, n, P, dt)
            e1.record()
            rec["tile_gather"].append((e0, e1, float(n * P * P * (feats.shape[0] * 4 + r.shape[-1] * r.element_size()))))
            return r

        def accumulate(out, wk, slot, t0, t1, nti, ntj, P, S, acc, cnt, r0, r1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            oa(out, wk, slot, t0, t1, nti, ntj, P, S, acc, cnt, r0, r1)
            e1.record()
            # per pixel of the rows touched: read+write acc and cnt; per tile pixel: read the prediction
            rec["blend_accumulate"].append((e0, e1, float((r1 - r0) * acc.shape[1] * 16 + (t1 - t0) * P * P * 4)))

        be.gather, be.accumulate = gather, accumulate
        return self

    def __exit__(self, *exc):
        self.be.gather, self.be.accumulate = self.orig

    def report(self, steps):
        out = {}
        for k, rows in self.rec.items():
            if not rows:
                continue
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in rows)
            byts = sum(b for _, _, b in rows)
            gbs = byts / (ms * 1e-3) / 1e9
            out[k] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                      "ms_per_grid": round(ms / steps, 3), "launches_per_grid": len(rows) // steps}
        return out


def cpu_baseline_predict(feats, n):
    import numpy as np
    from oracle import tile_oracle, unet_oracle
    from oracle import weights as W
    threads = host_threads()
    torch.set_num_threads(threads)
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=42)
    fn = lambda x: unet_oracle.unet_forward(sd, torch.from_numpy(x), training=False)[0].numpy()
    sub = lambda m: {v: np.ascontiguousarray(feats[i, :m, :m].cpu().numpy()) for i, v in enumerate(tile_oracle.FEATURE_VARS)}
    tile_oracle.predict_large_area(fn, sub(512), 256, 128, 16)             # warm-up: 9 tiles
    fs = sub(n)
    t0 = time.perf_counter()
    tile_oracle.predict_large_area(fn, fs, 256, 128, 16)
    t = time.perf_counter() - t0
    ntile = ((n - 256) // 128 + 1) ** 2
    return {"value": round(n * n / 1e6 / t, 4), "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": f"numpy/torch-CPU oracle on the {n}x{n} corner of the same grid ({ntile} tiles), one timed pass "
                      f"after a 512x512 warm-up, {threads} threads"}


# ---------------------------------------------------------------------------- main
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("UNETK_BENCH_DRYRUN"):
        # launcher rehearsal without GPUs: rendezvous over gloo, one reduction, the one JSON line
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "train patches/s (fwd+bwd)", "value": 0.0, "n_gpus": world, "dry_run": True,
                              "rank_sum": t.item()}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # stdout carries the one JSON line and nothing else: whatever a library prints there meanwhile (RCCL's version
    # banner under NCCL_DEBUG=VERSION, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi(world):
        import torch.distributed as dist
        if world == 1 and "MASTER_ADDR" not in os.environ:       # the one-rank rehearsal started by hand
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    runners = {"train": run_train, "swin": run_swin, "predict": run_predict}
    if args.workload != "all":
        line = runners[args.workload](args, rank, world, dev, dt)
    else:
        line = run_train(args, rank, world, dev, dt)
        extra = {}
        for name in ("swin", "predict"):
            try:
                extra[name] = runners[name](args, rank, world, dev, dt)
            except Exception as e:      # the headline line must still come out; the failure is recorded in it
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
                if multi(world):
                    raise
        if rank == 0:
            line.update(extra)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if multi(world):
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
