"""CPU oracle (test infrastructure): the UNet forward as a function of a state_dict.

Restates reference training.py:496-531 (== enhanced_training_multi_loss.py:563-602)
with plain torch-CPU fp32 ops.  Backward comes from torch autograd over this
function.  Train-mode Dropout2d is made explicit: the caller passes the per-block
channel keep masks, or asks for them to be drawn from torch's CPU generator in the
same order and with the same calls the reference's nn.Dropout2d makes, which
reproduces a seeded reference run bit for bit.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BLOCKS = ("encoder1", "encoder2", "encoder3", "encoder4", "bottleneck",
          "decoder4", "decoder3", "decoder2", "decoder1")
DROPOUT_P = 0.05      # reference training.py:530
BN_EPS = 1e-5         # nn.BatchNorm2d default
BN_MOMENTUM = 0.1     # nn.BatchNorm2d default


def draw_dropout_masks(batch: int, init_features: int, p: float = DROPOUT_P) -> List[torch.Tensor]:
    """Keep masks (B, C) in {0, 1/(1-p)}, in the order the nine blocks run.

    nn.Dropout2d draws one Bernoulli(1-p) per (sample, channel) on a (B, C, 1, 1)
    tensor and divides by (1-p); the blocks execute in BLOCKS order.
    """
    f = init_features
    chans = [f, 2 * f, 4 * f, 8 * f, 16 * f, 8 * f, 4 * f, 2 * f, f]
    masks = []
    for c in chans:
        if p == 0.0:
            masks.append(torch.ones(batch, c))
        else:
            noise = torch.empty(batch, c, 1, 1).bernoulli_(1 - p).div_(1 - p)
            masks.append(noise.view(batch, c))
    return masks


def _r(t):
    """Round to bfloat16 storage and back, with a straight-through gradient."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bfloat16 storage: models a gradient tensor kept in bf16."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


BF16_DEFAULT = frozenset({"w", "y", "a"})
BF16_ALL = frozenset({"w", "y", "a", "dy", "da"})


def _classes(emulate_bf16):
    """emulate_bf16: False, True (weights, raw conv outputs and activations are rounded) or a set of the
    stored-tensor classes to round: 'w' packed weights, 'y' raw conv output, 'a' activation (and the
    network input and resized tensors), 'dy' gradient of the conv output, 'da' gradient of the activation."""
    if not emulate_bf16:
        return frozenset()
    if emulate_bf16 is True:
        return BF16_DEFAULT
    return frozenset(emulate_bf16)


def _double_conv_bf16(sd, name, x, training, mask, new_stats, classes=BF16_DEFAULT):
    """_double_conv with the roundings a bfloat16-storage pipeline performs.

    Conv operands (activations, weights) and every stored activation are rounded to bf16;
    accumulation, statistics and the normalisation arithmetic stay in fp32.  In training the
    raw conv output is stored (rounded) before BatchNorm is applied to it, while the batch
    statistics are taken from the unrounded accumulators; in eval the folded BatchNorm acts on
    the accumulators directly.  This is the model the bf16 GPU path is held to tightly; its
    distance from the fp32 model is the price of bf16 storage, not of the kernels.
    """
    for unit, (conv_i, bn_i) in enumerate(((0, 1), (3, 4))):
        wgt = sd[f"{name}.{conv_i}.weight"]
        y32 = F.conv2d(x, _r(wgt) if "w" in classes else wgt, None, padding=1)
        if "dy" in classes:
            y32 = _RoundGrad.apply(y32)
        g, b = sd[f"{name}.{bn_i}.weight"], sd[f"{name}.{bn_i}.bias"]
        if training:
            mean = y32.mean((0, 2, 3))
            var = y32.var((0, 2, 3), unbiased=False)
            n = y32.numel() / y32.shape[1]
            new_stats[f"{name}.{bn_i}.running_mean"] = (
                (1 - BN_MOMENTUM) * sd[f"{name}.{bn_i}.running_mean"] + BN_MOMENTUM * mean.detach())
            new_stats[f"{name}.{bn_i}.running_var"] = (
                (1 - BN_MOMENTUM) * sd[f"{name}.{bn_i}.running_var"]
                + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            y = _r(y32) if "y" in classes else y32
        else:
            mean, var = sd[f"{name}.{bn_i}.running_mean"], sd[f"{name}.{bn_i}.running_var"]
            y = y32
        scale = g / torch.sqrt(var + BN_EPS)
        shift = b - mean * scale
        x = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        if unit == 1 and training and mask is not None:   # the mask is applied before the single rounding
            x = x * mask.view(mask.shape[0], mask.shape[1], 1, 1)
        if "a" in classes:
            x = _r(x)
        if "da" in classes:
            x = _RoundGrad.apply(x)
    return x


def _double_conv(sd, name, x, training, mask, new_stats):
    # reference training.py:522-531: conv3x3(no bias) -> BN -> ReLU, twice, then Dropout2d
    for conv_i, bn_i in ((0, 1), (3, 4)):
        x = F.conv2d(x, sd[f"{name}.{conv_i}.weight"], None, padding=1)
        rm = sd[f"{name}.{bn_i}.running_mean"].clone()
        rv = sd[f"{name}.{bn_i}.running_var"].clone()
        x = F.batch_norm(x, rm, rv, sd[f"{name}.{bn_i}.weight"], sd[f"{name}.{bn_i}.bias"],
                         training, BN_MOMENTUM, BN_EPS)
        if training:
            new_stats[f"{name}.{bn_i}.running_mean"] = rm
            new_stats[f"{name}.{bn_i}.running_var"] = rv
        x = F.relu(x)
    if training and mask is not None:
        x = x * mask.view(mask.shape[0], mask.shape[1], 1, 1)
    return x


def unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool = False,
                 dropout_masks: Optional[List[torch.Tensor]] = None, emulate_bf16: bool = False
                 ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Returns (output (B,out,H,W), updated running statistics when training).

    reference training.py:496-520: four encoder blocks with 2x2 max-pool between
    them, bottleneck, then four times [bilinear resize to the skip's size
    (align_corners=False), concat (upsampled, skip), block], then the 1x1 conv.
    """
    new_stats: Dict[str, torch.Tensor] = {}
    m = dropout_masks if dropout_masks is not None else [None] * 9
    classes = _classes(emulate_bf16)
    if classes:
        block = lambda *a: _double_conv_bf16(*a, classes=classes)
    else:
        block = _double_conv
    rnd = _r if "a" in classes else (lambda t: t)
    x = rnd(x)
    e1 = block(sd, "encoder1", x, training, m[0], new_stats)
    e2 = block(sd, "encoder2", F.max_pool2d(e1, 2, 2), training, m[1], new_stats)
    e3 = block(sd, "encoder3", F.max_pool2d(e2, 2, 2), training, m[2], new_stats)
    e4 = block(sd, "encoder4", F.max_pool2d(e3, 2, 2), training, m[3], new_stats)
    d = block(sd, "bottleneck", F.max_pool2d(e4, 2, 2), training, m[4], new_stats)
    for i, (name, skip) in enumerate((("decoder4", e4), ("decoder3", e3),
                                      ("decoder2", e2), ("decoder1", e1))):
        d = rnd(F.interpolate(d, size=skip.shape[2:], mode="bilinear", align_corners=False))
        d = torch.cat((d, skip), dim=1)
        d = block(sd, name, d, training, m[5 + i], new_stats)
    out = F.conv2d(d, sd["conv.weight"], sd["conv.bias"])
    return out, new_stats


def unet_fwd_bwd(sd, x, loss_fn, dropout_masks=None, emulate_bf16=False):
    """One training-mode forward + backward.  Returns (out, loss, grads by name, new stats)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    full = dict(sd)
    full.update(params)
    xin = x.clone().requires_grad_(True)
    out, stats = unet_forward(full, xin, True, dropout_masks, emulate_bf16)
    loss = loss_fn(out)
    loss.backward()
    grads = {k: p.grad for k, p in params.items()}
    grads["__input__"] = xin.grad
    return out.detach(), loss.detach(), grads, stats
