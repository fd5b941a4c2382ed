#!/usr/bin/env python
"""Host time to QUEUE one Swin-UNet trainer step against the GPU time to run it (b64, eager, as bench.py --workload swin)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import swin_oracle, weights as W          # (tooling: the default configuration and the synthetic batch)
from unet_amd import ops
from unet_amd.losses import MultiLossFunction
from unet_amd.optim import FusedAdamW
from unet_amd.swin import SwinUNet
import bench

dev = torch.device("cuda")
cfg = dict(swin_oracle.DEFAULT_CONFIG)
model = SwinUNet(cfg, compute_dtype=torch.bfloat16)
model.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=42))
model = model.to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, y, w = [t.to(dev) for t in W.synthetic_batch(B, 4, 256, 256, seed=42)]
crit = MultiLossFunction(bench.SWIN_LOSS_CFG).to(dev)
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)

def step():
    for p in model.parameters():
        p.grad = None
    out = model(x)
    out = ops.interpolate_bilinear(out, y.shape[-2:])
    loss, _ = crit(out, y, w)
    loss.backward()
    opt.step()

for _ in range(10):
    step()
torch.cuda.synchronize()
n = 30
host = 0.0
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"batch {B}: wall {wall / n * 1e3:.2f} ms per step, host time inside step() {host / n * 1e3:.2f} ms "
      f"(equal: the host is the limit and the queue never fills)")
# the same with the queue drained before every step: pure host issue time
host = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print(f"host time to queue one step on an idle GPU: {host / 10 * 1e3:.2f} ms")

if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    # the backward pass runs in the autograd engine's thread: it gets a profiler of its own
    from unet_amd import swin as swin_mod
    pr2 = cProfile.Profile()
    orig_bwd = swin_mod._SwinTrainFn.backward

    def prof_bwd(ctx, dout):
        pr2.enable()
        try:
            return orig_bwd(ctx, dout)
        finally:
            pr2.disable()
    swin_mod._SwinTrainFn.backward = staticmethod(prof_bwd)
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    swin_mod._SwinTrainFn.backward = staticmethod(orig_bwd)
    print("---- backward thread")
    pstats.Stats(pr2).sort_stats(os.environ.get("HOST_PROFILE_SORT", "tottime")).print_stats(40)
    print("---- calling thread")
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats(os.environ.get("HOST_PROFILE_SORT", "tottime")).print_stats(45)
