"""A training step captured into a HIP graph replays the same arithmetic as the eager step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(kind):
    from oracle import swin_oracle
    from oracle import weights as W
    if kind == "unet":
        from unet_amd import OptimizedUNet

        def mk():
            m = OptimizedUNet(4, 1, 8)
            m.load_state_dict(W.fill_state_dict(W.unet_template(4, 1, 8), seed=11))
            for mod in m.modules():                  # both runs must see the same (no) dropout draws
                if isinstance(mod, torch.nn.Dropout2d):
                    mod.p = 0.0
            return m
        shape, oshape, post = (4, 4, 32, 32), (4, 1, 32, 32), None
    else:
        from unet_amd import ops
        from unet_amd.swin import SwinUNet
        cfg = dict(swin_oracle.DEFAULT_CONFIG, drop_path_rate=0.0, drop_rate=0.0)

        def mk():
            m = SwinUNet(cfg)
            m.load_state_dict(W.fill_state_dict(swin_oracle.swin_template(cfg), seed=11))
            return m
        shape, oshape = (2, 4, 64, 64), (2, 1, 64, 64)       # the model predicts 32x32: resized to the label
        post = lambda out, y: ops.interpolate_bilinear(out, y.shape[-2:])
    return mk, shape, oshape, post


@pytest.mark.parametrize("kind", ["unet", "swin"])
def test_graphed_step_matches_eager(kind):
    from oracle import loss_oracle
    from unet_amd.graph import GraphedTrainStep
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    mk, shape, oshape, post = _models(kind)
    ref = mk().to(dev).train()
    got = mk().to(dev).train()
    crit = MultiLossFunction(loss_oracle.preset("weighted_mse_grad", grad_w=0.3, spec_w=0.0)).to(dev)
    opt_ref = FusedAdamW(ref.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0)
    opt_got = FusedAdamW(got.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0, capturable=True)
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(shape, generator=g).to(dev), torch.randn(oshape, generator=g).to(dev),
                torch.rand(oshape, generator=g).to(dev)) for _ in range(5)]
    before = [p.detach().clone() for p in got.parameters()]
    step = GraphedTrainStep(got, crit, opt_got, *batches[0], post=post, warmup=2)
    for p, b in zip(got.parameters(), before):            # constructing the graph trains nothing
        assert torch.equal(p, b)
    assert opt_got.device_step_count() == 0
    for i, (x, y, w) in enumerate(batches):
        if i == 3:                                        # a scheduler changes the learning rate mid-run
            for o in (opt_ref, opt_got):
                o.param_groups[0]["lr"] = 5e-4
        opt_ref.zero_grad(set_to_none=True)
        out = ref(x)
        if post is not None:
            out = post(out, y)
        loss_ref, parts_ref = crit(out, y, w)
        loss_ref.backward()
        opt_ref.step()
        loss_got, parts_got = step(x, y, w)
        assert abs(float(loss_got) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref))), i
        assert abs(parts_got["pixel"] - parts_ref["pixel"]) <= 1e-5 * max(1.0, abs(parts_ref["pixel"]))
    assert opt_got.device_step_count() == len(batches)
    for (n, a), b in zip(ref.named_parameters(), got.parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=n)
    for (n, a), b in zip(ref.named_buffers(), got.buffers()):
        np.testing.assert_allclose(b.detach().float().cpu().numpy(), a.detach().float().cpu().numpy(), rtol=1e-4,
                                   atol=2e-6, err_msg=n)


def test_graphed_step_rejects_host_side_optimizer():
    from oracle import loss_oracle
    from unet_amd import OptimizedUNet
    from unet_amd.graph import GraphedTrainStep
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    m = OptimizedUNet(4, 1, 8).to(dev)
    opt = FusedAdamW(m.parameters())
    x = torch.zeros(1, 4, 32, 32, device=dev)
    with pytest.raises(ValueError):
        GraphedTrainStep(m, MultiLossFunction(loss_oracle.preset("all")), opt, x, torch.zeros(1, 1, 32, 32, device=dev))


def test_graphed_step_resumes_from_a_checkpoint_and_survives_an_eager_step():
    """An optimizer loaded from a checkpoint with step > 0 keeps that step through the construction of the graph
    (the warm-up steps are undone back to the loaded counter, not to zero), graph replays and an eager step in
    between (a short last batch) follow the eager optimizer, and state_dict() reports the true step count."""
    from oracle import loss_oracle
    from unet_amd.graph import GraphedTrainStep
    from unet_amd.losses import MultiLossFunction
    from unet_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    mk, shape, oshape, post = _models("unet")
    crit = MultiLossFunction(loss_oracle.preset("weighted_mse_grad", grad_w=0.3, spec_w=0.0)).to(dev)
    g = torch.Generator().manual_seed(9)
    batches = [(torch.randn(shape, generator=g).to(dev), torch.randn(oshape, generator=g).to(dev),
                torch.rand(oshape, generator=g).to(dev)) for _ in range(7)]
    short = (torch.randn((2,) + shape[1:], generator=g).to(dev), torch.randn((2,) + oshape[1:], generator=g).to(dev),
             torch.rand((2,) + oshape[1:], generator=g).to(dev))

    def eager(m, opt, x, y, w):
        opt.zero_grad(set_to_none=True)
        loss, _ = crit(m(x), y, w)
        loss.backward()
        opt.step()

    # three eager steps, then a checkpoint of model + optimizer (what the reference trainer saves)
    first = mk().to(dev).train()
    opt0 = FusedAdamW(first.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0)
    for b in batches[:3]:
        eager(first, opt0, *b)
    ck_model = {k: v.clone() for k, v in first.state_dict().items()}
    import copy
    ck_opt = copy.deepcopy(opt0.state_dict())
    assert all(float(s["step"]) == 3 for s in ck_opt["state"].values())
    ref, got = mk().to(dev).train(), mk().to(dev).train()
    ref.load_state_dict(ck_model)
    got.load_state_dict(ck_model)
    opt_ref = FusedAdamW(ref.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0)
    opt_got = FusedAdamW(got.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0, capturable=True)
    opt_ref.load_state_dict(copy.deepcopy(ck_opt))      # (load_state_dict adopts tensors that need no cast)
    opt_got.load_state_dict(copy.deepcopy(ck_opt))
    step = GraphedTrainStep(got, crit, opt_got, *batches[3], post=post, warmup=2)
    assert opt_got.device_step_count() == 3
    for i, b in enumerate(batches[3:]):
        eager(ref, opt_ref, *b)
        step(*b)
        if i == 1:          # a batch of another shape cannot use the captured graph: eager step in between
            eager(ref, opt_ref, *short)
            eager(got, opt_got, *short)
    nsteps = 3 + len(batches[3:]) + 1
    assert opt_got.device_step_count() == nsteps
    sd_got, sd_ref = opt_got.state_dict(), opt_ref.state_dict()
    assert all(float(s["step"]) == nsteps for s in sd_got["state"].values())
    assert all(float(s["step"]) == nsteps for s in sd_ref["state"].values())
    for (n, a), b in zip(ref.named_parameters(), got.parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=n)
    for k, s in sd_ref["state"].items():
        np.testing.assert_allclose(sd_got["state"][k]["exp_avg"].cpu().numpy(), s["exp_avg"].cpu().numpy(),
                                   rtol=1e-3, atol=1e-7)
    # and the checkpoint written in capturable mode resumes an ordinary optimizer at the right step
    opt_new = FusedAdamW(ref.parameters(), lr=2e-3, weight_decay=0.05, max_grad_norm=1.0)
    opt_new.load_state_dict(copy.deepcopy(sd_got))
    assert all(int(opt_new.state[p]["step"]) == nsteps for p in ref.parameters())
