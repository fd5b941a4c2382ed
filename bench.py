#!/usr/bin/env python
"""Headline benchmark: UNet training step (forward + multi-loss + backward) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one synthetic batch that is already resident in HBM:
OptimizedUNet(4,1,48) forward, the 'all' multi-loss, backward to every parameter (for N > 1 the
data-parallel gradient all-reduce over RCCL), global-norm clip and the AdamW update.  Workload = BASELINE config 2: batch 64 of
4x256x256 patches per GPU, bf16 storage with fp32 accumulation.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     - the dominant kernel, the MFMA implicit-GEMM conv with 96-channel blocks (forward and
                 data-gradient launches): algorithmic FLOPs per launch / average launch duration,
                 measured with events on the launch stream inside the timed steps, against the
                 2.5 PFLOP/s dense bf16 peak (plus the same ratio over the whole conv family);
  cpu_baseline - the CPU oracle (torch fp32) timed on this host on a bounded sample of the same
                 workload (batch 4, same network and loss).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X
PEAK_F32_TFLOPS = 157.3
LOSS_CFG = {   # training_variants.py:119-130 "all" preset
    "loss_combination": "all", "pixel_loss_weight": 1.0, "gradient_loss_weight": 0.3,
    "spectral_loss_weight": 0.2, "pixel_loss_type": "mse", "use_tid_weights": True,
    "tid_weight_scale": 1.0, "gradient_method": "sobel", "spectral_focus_high_freq": True,
    "high_freq_weight_factor": 2.0,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="train", choices=["train", "swin", "predict"],
                    help="train: UNet step (headline); swin: Swin-UNet step; predict: tiled large-area inference")
    ap.add_argument("--grid", type=int, default=8192, help="predict: side of the synthetic grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def make_loss():
    try:
        from unet_amd.losses import MultiLossFunction
        return MultiLossFunction(LOSS_CFG), "all"
    except ImportError:
        return None, "mse"


def train_step(model, crit, x, y, w):
    for p in model.parameters():
        p.grad = None
    out = model(x)
    if crit is not None:
        loss, _ = crit(out, y, w)
    else:
        loss = ((out - y) ** 2).mean()
    loss.backward()
    return loss


class ConvTimer:
    """Brackets every implicit-GEMM conv launch (forward and data-gradient) of the timed steps with
    events on the launch stream.  The roofline entry is the dominant instantiation,
    conv_igemm_kernel<bf16, NT=6, KS=3> (96-channel blocks, 3x3): 27 of the 36 launches per step."""

    def __init__(self):
        from unet_amd import ops
        self.ops = ops
        self.orig = ops.conv_fwd
        self.records = []

    def __enter__(self):
        orig, records = self.orig, self.records

        def timed(xin, pw, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(xin, pw, *a, **k)
            e1.record()
            B, H, W_, _ = xin.shape
            records.append((e0, e1, 2.0 * B * H * W_ * pw.Co * pw.Ci * pw.ks * pw.ks, pw.bn, pw.ks))
            return r

        self.ops.conv_fwd = timed
        return self

    def __exit__(self, *exc):
        self.ops.conv_fwd = self.orig

    def roofline(self, dt, steps):
        peak = PEAK_BF16_TFLOPS if dt == torch.bfloat16 else PEAK_F32_TFLOPS
        dom = [(e0.elapsed_time(e1), f) for e0, e1, f, bn, ks in self.records if bn == 96 and ks == 3]
        fam = [(e0.elapsed_time(e1), f) for e0, e1, f, bn, ks in self.records]
        ms, flops = sum(t for t, _ in dom), sum(f for _, f in dom)
        fms, fflops = sum(t for t, _ in fam), sum(f for _, f in fam)
        achieved = flops / (ms * 1e-3) / 1e12
        tname = "bf16" if dt == torch.bfloat16 else "f32"
        traffic = None
        try:   # HBM bytes per launch of this kernel from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(f"conv_igemm_kernel<{tname},6,3>", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        return {"bound": "mfma", "kernel": f"conv_igemm_kernel<{tname},6,3> (forward + data-gradient launches)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "launches_per_step": len(dom) // steps,
                "avg_launch_us": round(ms * 1e3 / len(dom), 2),
                "algorithmic_gflop_per_launch": round(flops / len(dom) / 1e9, 2),
                "conv_family": {"launches_per_step": len(fam) // steps,
                                "ms_per_step": round(fms / steps, 3),
                                "achieved": round(fflops / (fms * 1e-3) / 1e12, 2)}}


def cpu_baseline(batch, size):
    """The CPU oracle on a bounded sample: UNet(4,1,48) fwd + 'all' loss + bwd, batch `batch`."""
    from oracle import loss_oracle, unet_oracle
    from oracle import weights as W
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=42)
    x, y, w = W.synthetic_batch(batch, 4, size, size, seed=42)
    cfg = loss_oracle.preset("all")
    times = []
    for i in range(3):
        t0 = time.perf_counter()
        unet_oracle.unet_fwd_bwd(sd, x, lambda o: loss_oracle.multi_loss(cfg, o, y, w)[0])
        times.append(time.perf_counter() - t0)
    t = min(times[1:])
    return {"value": round(batch / t, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU fp32 oracle, batch {batch} of 4x{size}x{size}, fwd+loss+bwd, best of 2 after 1 warm-up"}


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()


def timed(fn, steps, warmup, world, dev):
    """warmup untimed calls, then `steps` calls between barriers; max over ranks, seconds."""
    for _ in range(warmup):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier_sync(world)
    el = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    return el


def run_predict(args, rank, world, dev, dt):
    """Tiled large-area inference (BASELINE config 4): UNet(4,1,48) eval over a synthetic grid,
    256-px tiles with 50 % overlap, Gaussian blend; tile rows sharded over the ranks (strong scaling)."""
    import numpy as np
    from oracle import weights as W
    from unet_amd import OptimizedUNet
    from unet_amd.predictor import LargeAreaPredictor, LargeAreaPredictionConfig
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
    with torch.no_grad():
        el = timed(lambda: pred.predict_device(model, feats), args.steps, args.warmup, world, dev)
    ms = el / args.steps * 1e3
    mpix = args.grid * args.grid / 1e6
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tile_oracle, unet_oracle
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=42)
        n = 768
        fs = {v: feats[i, :n, :n].cpu().numpy() for i, v in enumerate(tile_oracle.FEATURE_VARS)}
        fn = lambda x: unet_oracle.unet_forward(sd, torch.from_numpy(x), training=False)[0].numpy()
        t0 = time.perf_counter()
        tile_oracle.predict_large_area(fn, fs, 256, 128, 16)
        t = time.perf_counter() - t0
        cpu = {"value": round(n * n / 1e6 / t, 4), "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"numpy/torch-CPU oracle on a {n}x{n} corner of the same grid (25 tiles)"}
    if rank == 0:
        ntile = ((args.grid - 256) // 128 + 1) ** 2
        print(json.dumps({
            "metric": "tiled-predict Mpix/s", "value": round(mpix / (ms * 1e-3), 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"LargeAreaPredictor over a {args.grid}x{args.grid} grid, 256-px tiles, stride 128 "
                                   f"({ntile} tiles), UNet(4,1,48) eval, Gaussian blend",
                       "parallelism": f"tile-row bands over {world} GPU(s)"},
            "roofline": None, "cpu_baseline": cpu}), flush=True)


def run_swin(args, rank, world, dev, dt):
    """Swin-UNet training step (BASELINE config 3): window 8, shifted windows, 256x256 input."""
    from oracle import swin_oracle
    from oracle import weights as W
    from unet_amd.swin import SwinUNet
    cfg = dict(swin_oracle.DEFAULT_CONFIG)
    model = SwinUNet(cfg, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
    model = model.to(dev).train()
    B = args.batch
    x, _, _ = W.synthetic_batch(B, 4, args.size, args.size, seed=42 + rank)
    x = x.to(dev)
    tgt = torch.randn((B, 1, args.size // 2, args.size // 2), device=dev)
    reducer = None
    if world > 1:
        from unet_amd.dist import GradReducer
        reducer = GradReducer(model)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = ((model(x) - tgt) ** 2).mean()
        loss.backward()
        if reducer is not None:
            reducer.allreduce()

    el = timed(step, args.steps, args.warmup, world, dev)
    ms = el / args.steps * 1e3
    if rank == 0:
        print(json.dumps({
            "metric": "train patches/s (fwd+bwd)", "value": round(world * B * args.steps / el, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"SwinUNet (embed 96, depths 2-2-6-2, window 8) training step fwd+MSE+bwd, batch {B}/GPU "
                                   f"of 4x{args.size}x{args.size}", "global_batch": world * B,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": None, "cpu_baseline": None}), flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from oracle import weights as W
    from unet_amd import OptimizedUNet
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload != "train":
        (run_predict if args.workload == "predict" else run_swin)(args, rank, world, dev, dt)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = OptimizedUNet(4, 1, 48, compute_dtype=dt)
    model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
    model = model.to(dev).train()
    crit, loss_name = make_loss()
    if crit is not None:
        crit = crit.to(dev)
    x, y, w = W.synthetic_batch(args.batch, 4, args.size, args.size, seed=42 + rank)
    x, y, w = x.to(dev), y.to(dev), w.to(dev)

    reducer = None
    if world > 1:
        from unet_amd.dist import GradReducer
        reducer = GradReducer(model)
    # the trainer's optimizer tail (enhanced_training_multi_loss.py:758-763, 827-829)
    from unet_amd.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=5e-5, weight_decay=5e-5, max_grad_norm=1.0)

    def step():
        loss = train_step(model, crit, x, y, w)
        if reducer is not None:
            reducer.allreduce()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    timer = ConvTimer() if (rank == 0 and not args.no_roofline) else None
    if timer is not None:
        timer.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if timer is not None:
        timer.__exit__()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    ms = el / args.steps * 1e3
    value = world * args.batch * args.steps / el

    roof = None
    cpu = None
    if rank == 0:
        if timer is not None:
            roof = timer.roofline(dt, args.steps)
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(4, args.size)
        line = {
            "metric": "train patches/s (fwd+bwd)", "value": round(value, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"UNet(4,1,48) training step fwd+loss({loss_name})+bwd+clip+AdamW, batch {args.batch}/GPU of "
                                   f"4x{args.size}x{args.size}, {args.dtype} storage fp32 accumulate",
                       "global_batch": world * args.batch, "patch": args.size,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
