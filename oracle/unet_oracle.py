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
                 dropout_masks: Optional[List[torch.Tensor]] = None
                 ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Returns (output (B,out,H,W), updated running statistics when training).

    reference training.py:496-520: four encoder blocks with 2x2 max-pool between
    them, bottleneck, then four times [bilinear resize to the skip's size
    (align_corners=False), concat (upsampled, skip), block], then the 1x1 conv.
    """
    new_stats: Dict[str, torch.Tensor] = {}
    m = dropout_masks if dropout_masks is not None else [None] * 9
    e1 = _double_conv(sd, "encoder1", x, training, m[0], new_stats)
    e2 = _double_conv(sd, "encoder2", F.max_pool2d(e1, 2, 2), training, m[1], new_stats)
    e3 = _double_conv(sd, "encoder3", F.max_pool2d(e2, 2, 2), training, m[2], new_stats)
    e4 = _double_conv(sd, "encoder4", F.max_pool2d(e3, 2, 2), training, m[3], new_stats)
    d = _double_conv(sd, "bottleneck", F.max_pool2d(e4, 2, 2), training, m[4], new_stats)
    for i, (name, skip) in enumerate((("decoder4", e4), ("decoder3", e3),
                                      ("decoder2", e2), ("decoder1", e1))):
        d = F.interpolate(d, size=skip.shape[2:], mode="bilinear", align_corners=False)
        d = torch.cat((d, skip), dim=1)
        d = _double_conv(sd, name, d, training, m[5 + i], new_stats)
    out = F.conv2d(d, sd["conv.weight"], sd["conv.bias"])
    return out, new_stats


def unet_fwd_bwd(sd, x, loss_fn, dropout_masks=None):
    """One training-mode forward + backward.  Returns (out, loss, grads by name, new stats)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    full = dict(sd)
    full.update(params)
    xin = x.clone().requires_grad_(True)
    out, stats = unet_forward(full, xin, True, dropout_masks)
    loss = loss_fn(out)
    loss.backward()
    grads = {k: p.grad for k, p in params.items()}
    grads["__input__"] = xin.grad
    return out.detach(), loss.detach(), grads, stats
