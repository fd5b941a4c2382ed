"""CPU tests: reading the reference trainer's checkpoint layout without executing anything from the file.

enhanced_training_multi_loss.py:956-971 saves {'epoch', 'model_state_dict', 'optimizer_state_dict',
'scheduler_state_dict', 'val_loss', 'config': <config class INSTANCE>, 'loss_config', 'history'}; the history
holds numpy float64 scalars (np.corrcoef, :889).  large_area_predictor.py:86-97 reads such files.
"""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from oracle import weights as W


class FakeTrainingConfig:
    """Plays the trainer's config class: pickled by qualified name, attributes in the instance dict."""
    PATCH_SIZE = 128

    def __init__(self):
        self.LEARNING_RATE = 5e-5
        self.LOSS_CONFIG = {"loss_combination": "all"}


def _reference_layout(path, sd, as_main=False):
    cfg = FakeTrainingConfig()
    if as_main:      # the trainer scripts are run as __main__, which is what their checkpoints name
        FakeTrainingConfig.__module__ = "__main__"
        setattr(sys.modules["__main__"], "FakeTrainingConfig", FakeTrainingConfig)
    opt_state = {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.zeros(3)}},
                 "param_groups": [{"lr": 5e-5, "betas": (0.9, 0.999), "params": [0]}]}
    try:
        torch.save({"epoch": 7, "model_state_dict": sd, "optimizer_state_dict": opt_state,
                    "scheduler_state_dict": {"best": np.float64(0.25), "num_bad_epochs": 0},
                    "val_loss": 0.0184, "config": cfg, "loss_config": cfg.LOSS_CONFIG,
                    "history": {"val_correlation": [np.float64(0.99), np.float64(0.991)],
                                "train_loss_total": [0.5, 0.25], "curve": np.arange(4.0)}}, path)
    finally:
        if as_main:
            FakeTrainingConfig.__module__ = __name__
            delattr(sys.modules["__main__"], "FakeTrainingConfig")


@pytest.mark.parametrize("as_main", [False, True])
def test_reference_trainer_checkpoint_loads_bit_equal(tmp_path, as_main):
    from unet_amd.predictor import InertObject, load_checkpoint
    sd = W.fill_state_dict(W.unet_template(4, 1, 8), seed=3)
    path = str(tmp_path / "best_model.pth")
    _reference_layout(path, sd, as_main)
    with pytest.raises(pickle.UnpicklingError):          # what the plain weights-only load makes of it
        torch.load(path, map_location="cpu", weights_only=True)
    ckpt = load_checkpoint(path)
    assert set(ckpt["model_state_dict"]) == set(sd)
    for k, v in sd.items():
        assert torch.equal(ckpt["model_state_dict"][k], v), k
    assert ckpt["epoch"] == 7 and ckpt["loss_config"] == {"loss_combination": "all"}
    cfg = ckpt["config"]
    assert isinstance(cfg, InertObject) and not isinstance(cfg, FakeTrainingConfig)
    assert cfg.LEARNING_RATE == 5e-5                     # attributes survive; the class itself was never imported
    assert type(cfg).__name__ == "FakeTrainingConfig"
    assert torch.equal(ckpt["optimizer_state_dict"]["state"][0]["step"], torch.tensor(3.0))


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /tmp/unetk_pwned",))


def test_callable_pickles_are_refused(tmp_path):
    from unet_amd.predictor import LargeAreaPredictionConfig, LargeAreaPredictor, load_checkpoint
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=3)
    path = str(tmp_path / "evil.pth")
    torch.save({"model_state_dict": sd, "config": _Evil()}, path)
    if os.path.exists("/tmp/unetk_pwned"):
        os.remove("/tmp/unetk_pwned")
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(path)
    cfg = type("C", (LargeAreaPredictionConfig,), {"OUTPUT_DIR": str(tmp_path)})
    assert LargeAreaPredictor(cfg).load_model(path, device=torch.device("cpu")) is None
    assert not os.path.exists("/tmp/unetk_pwned")


def test_load_model_reads_the_trainer_layout_on_cpu(tmp_path):
    """load_model end to end up to the device move (CPU here; the kernels need a GPU to run the model)."""
    from unet_amd.predictor import LargeAreaPredictionConfig, LargeAreaPredictor
    sd = W.fill_state_dict(W.unet_template(4, 1, 48), seed=3)
    path = str(tmp_path / "best_model.pth")
    _reference_layout(path, sd, as_main=True)
    cfg = type("C", (LargeAreaPredictionConfig,), {"OUTPUT_DIR": str(tmp_path)})
    model = LargeAreaPredictor(cfg).load_model(path, device=torch.device("cpu"))
    assert model is not None and not model.training
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert LargeAreaPredictor(cfg).load_model(str(tmp_path / "missing.pth"), device=torch.device("cpu")) is None
