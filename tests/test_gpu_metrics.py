"""GPU tests: streaming validation metrics against the reference's library calls."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import metrics_oracle

pytestmark = pytest.mark.gpu


def _batches(seed, shapes, corr):
    g = np.random.Generator(np.random.PCG64(seed))
    out = []
    for s in shapes:
        t = g.normal(0.3, 1.2, s).astype(np.float32)
        p = (corr * t + (1 - corr) * g.normal(0, 1, s)).astype(np.float32)
        out.append((p, t))
    return out


@pytest.mark.parametrize("shapes,corr", [([(4, 1, 32, 32)] * 3, 0.8), ([(1, 1, 7, 5), (3, 1, 7, 5)], -0.5),
                                         ([(64, 1, 256, 256)], 0.99), ([(2, 1, 1, 1)], 0.3)])
def test_streaming_metrics_match_library_calls(shapes, corr):
    from scipy.stats import pearsonr
    from unet_amd.metrics import StreamingMetrics
    data = _batches(11, shapes, corr)
    m = StreamingMetrics()
    for p, t in data:
        m.update(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())
    got = m.compute()
    want = metrics_oracle.validation_metrics([p for p, _ in data], [t for _, t in data])
    assert got["count"] == want["count"]
    assert abs(got["correlation"] - want["correlation"]) < 1e-9
    assert abs(got["mse"] - want["mse"]) < 1e-10 * max(1, want["mse"]) and abs(got["mae"] - want["mae"]) < 1e-10
    # the calls predict.py makes, on the concatenated tensors
    P = torch.from_numpy(np.concatenate([p.reshape(-1) for p, _ in data]))
    T = torch.from_numpy(np.concatenate([t.reshape(-1) for _, t in data]))
    assert abs(got["mse"] - F.mse_loss(P, T).item()) < 2e-5 * max(1, want["mse"])
    assert abs(got["mae"] - F.l1_loss(P, T).item()) < 2e-5
    if P.numel() > 2:
        assert abs(got["correlation"] - pearsonr(P.numpy(), T.numpy())[0]) < 1e-5
    m.reset()
    assert m.compute()["count"] == 0
    with pytest.raises(ValueError):
        m.update(torch.zeros(2, 3).cuda(), torch.zeros(3, 2).cuda())
