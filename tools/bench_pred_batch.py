import sys, os, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/unet_amd") else ".")
import torch
from oracle import weights as W
from unet_amd import OptimizedUNet
from unet_amd.predictor import LargeAreaPredictionConfig, LargeAreaPredictor
dev = torch.device("cuda")
model = OptimizedUNet(4, 1, 48, compute_dtype=torch.bfloat16)
model.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 48), seed=42))
model = model.to(dev).eval()
feats = torch.randn((4, 8192, 8192), generator=torch.Generator().manual_seed(7)).to(dev)
for db in (64, 32, 96, 128, 192, 64):
    class Cfg(LargeAreaPredictionConfig):
        PATCH_SIZE = 256
        STRIDE = 128
        OUTPUT_DIR = "/tmp/unetk_bench_pred"
        DEVICE_BATCH = db
    pred = LargeAreaPredictor(Cfg)
    with torch.no_grad():
        for _ in range(2):
            pred.predict_device(model, feats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pred.predict_device(model, feats)
        torch.cuda.synchronize()
    print(f"DEVICE_BATCH {db}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per grid", flush=True)
