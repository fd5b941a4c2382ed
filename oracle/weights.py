"""Seeded synthetic weights and inputs, reproducible without the reference.

Every tensor is drawn from its own numpy PCG64 stream keyed on (seed, name), so
the GPU box can regenerate exactly the weights the golden fixtures were made
with.  Magnitudes are chosen so that every term of the arithmetic matters
(non-trivial BatchNorm affine and running statistics, relative-position biases
of order one) instead of the near-identity values a fresh initialisation has.
"""
import hashlib

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def tensor_for(name: str, shape, seed: int = 42) -> torch.Tensor:
    """Synthetic value for one state_dict entry, chosen by the entry's role."""
    g = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if leaf == "relative_position_index":
        raise ValueError("index buffers are structural, not synthetic")
    if leaf == "running_mean":
        a = g.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        a = g.uniform(0.5, 1.5, shape)
    elif leaf == "relative_position_bias_table":
        a = g.normal(0.0, 0.5, shape)
    elif len(shape) == 4:  # conv / conv-transpose kernels
        fan = shape[1] * shape[2] * shape[3]
        a = g.normal(0.0, np.sqrt(2.0 / fan), shape)
    elif len(shape) == 2:  # linear
        a = g.normal(0.0, 1.0 / np.sqrt(shape[1]), shape)
    elif leaf == "weight":  # norm scale
        a = g.uniform(0.5, 1.5, shape)
    elif leaf == "bias":
        a = g.normal(0.0, 0.1, shape)
    else:
        a = g.normal(0.0, 1.0, shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def fill_state_dict(template: dict, seed: int = 42) -> dict:
    """Replace every float entry of ``template`` (name -> tensor) by its synthetic value.

    Integer buffers other than ``num_batches_tracked`` (the Swin
    relative-position index) are kept as they are.
    """
    out = {}
    for name, t in template.items():
        leaf = name.split(".")[-1]
        if leaf == "relative_position_index":
            out[name] = t.clone()
        else:
            out[name] = tensor_for(name, t.shape, seed)
    return out


def unet_template(in_channels=4, out_channels=1, init_features=48) -> dict:
    """Names and shapes of the UNet state_dict (reference training.py:458-494, 522-531)."""
    f = init_features
    blocks = [
        ("encoder1", in_channels, f), ("encoder2", f, 2 * f), ("encoder3", 2 * f, 4 * f),
        ("encoder4", 4 * f, 8 * f), ("bottleneck", 8 * f, 16 * f),
        ("decoder4", 24 * f, 8 * f), ("decoder3", 12 * f, 4 * f),
        ("decoder2", 6 * f, 2 * f), ("decoder1", 3 * f, f),
    ]
    sd = {}
    for name, ci, co in blocks:
        for conv_i, bn_i, cin in ((0, 1, ci), (3, 4, co)):
            sd[f"{name}.{conv_i}.weight"] = torch.empty(co, cin, 3, 3)
            sd[f"{name}.{bn_i}.weight"] = torch.empty(co)
            sd[f"{name}.{bn_i}.bias"] = torch.empty(co)
            sd[f"{name}.{bn_i}.running_mean"] = torch.empty(co)
            sd[f"{name}.{bn_i}.running_var"] = torch.empty(co)
            sd[f"{name}.{bn_i}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    sd["conv.weight"] = torch.empty(out_channels, f, 1, 1)
    sd["conv.bias"] = torch.empty(out_channels)
    return sd


def synthetic_batch(batch, channels, height, width, seed=42):
    """Features, labels ~ N(0,1); TID weights from {0,1,5} with p={0.2,0.7,0.1}.

    The value set and probabilities follow the weight classes the reference's
    preprocessing emits (reference data_preparation.py:82-88); the data are
    z-scored there (data_preparation.py:466-468), hence unit normals.
    """
    g = _rng(seed, f"batch:{batch}x{channels}x{height}x{width}")
    x = g.normal(0.0, 1.0, (batch, channels, height, width)).astype(np.float32)
    y = g.normal(0.0, 1.0, (batch, 1, height, width)).astype(np.float32)
    w = g.choice(np.array([0.0, 1.0, 5.0], dtype=np.float32), size=(batch, 1, height, width),
                 p=[0.2, 0.7, 0.1]).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w)
