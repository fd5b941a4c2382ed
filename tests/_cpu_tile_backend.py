"""Test double for unet_amd.predictor.HipTileBackend: the four tile kernels restated with numpy on
CPU tensors, and a toy per-sample network.  Lets the multi-rank band/strip protocol of
LargeAreaPredictor be exercised on CPU with the gloo backend; it is never part of the product."""
import numpy as np
import torch

from oracle import tile_oracle


class ToyNet(torch.nn.Module):
    compute_dtype = torch.float32

    def __init__(self):
        super().__init__()
        g = np.random.Generator(np.random.PCG64(5))
        self.w = torch.nn.Parameter(torch.from_numpy(g.normal(0, 0.3, (1, 4, 3, 3)).astype(np.float32)))

    def forward(self, x):
        # explicit shifted multiply-adds in a fixed order and IEEE-exact elementwise ops only, so
        # the bits of a sample's output do not depend on batch size, threading or vector tails
        xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
        H, Wd = x.shape[2], x.shape[3]
        acc = torch.zeros((x.shape[0], H, Wd), dtype=x.dtype)
        for c in range(4):
            for ky in range(3):
                for kx in range(3):
                    acc = acc + xp[:, c, ky:ky + H, kx:kx + Wd] * self.w.detach()[0, c, ky, kx]
        return (acc / (1.0 + acc.abs())).unsqueeze(1)


class NumpyTileBackend:
    def valid(self, feats, P, S, nti, ntj):
        f = feats.numpy()
        out = np.zeros(nti * ntj, bool)
        for ti in range(nti):
            for tj in range(ntj):
                win = f[:, ti * S:ti * S + P, tj * S:tj * S + P]
                out[ti * ntj + tj] = not any(np.all(np.isnan(win[v])) for v in range(f.shape[0]))
        return out

    def make_engine(self, model):
        def engine(batch):   # NHWC -> NCHW, one sample at a time so batching cannot change bits
            with torch.no_grad():
                return torch.cat([model(batch[i:i + 1].permute(0, 3, 1, 2).contiguous())
                                  for i in range(batch.shape[0])])
        return engine, torch.float32

    def gather(self, feats, pos_d, first, n, P, dt):
        f = feats.numpy()
        pos = pos_d.numpy()[first:first + n]
        tiles = [np.nan_to_num(f[:, i:i + P, j:j + P].copy(), 0) for i, j in pos]
        return torch.from_numpy(np.stack(tiles)).permute(0, 2, 3, 1).contiguous()

    def accumulate(self, out, wk, slot, t0, t1, nti, ntj, P, S, acc, cnt, r0, r1):
        a, c, w = acc.numpy(), cnt.numpy(), wk.numpy()
        sl = slot.numpy()
        o = out.numpy()
        for t in range(nti * ntj):
            s = sl[t]
            if t0 <= s < t1:
                i, j = (t // ntj) * S, (t % ntj) * S
                a[i:i + P, j:j + P] += np.nan_to_num(o[s - t0, 0], 0) * w
                c[i:i + P, j:j + P] += w

    def normalize(self, acc, cnt):
        a, c = acc.numpy(), cnt.numpy()
        return torch.from_numpy(np.divide(a, np.where(c > 1e-9, c, 1.0)))


def toy_features(lat, lon, seed, nan_block=None, nan_speckle=0.0):
    g = np.random.Generator(np.random.PCG64(seed))
    arrays = {}
    for v in tile_oracle.FEATURE_VARS:
        a = g.normal(0, 1, (lat, lon)).astype(np.float32)
        if nan_speckle:
            a[g.random((lat, lon)) < nan_speckle] = np.nan
        arrays[v] = a
    if nan_block is not None:
        i0, i1, j0, j1 = nan_block
        arrays["DOV_NS"][i0:i1, j0:j1] = np.nan
    return arrays
