"""CPU oracle (test infrastructure): the Swin-UNet forward as a function of a state_dict.

Restates reference swin_unet_model.py:139-686 with torch-CPU fp32 ops.  Backward
comes from torch autograd.  DropPath is made explicit through per-block keep
scales (B,) so that train-mode runs are reproducible.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

DEFAULT_CONFIG = {   # reference swin_unet_model.py:710-725
    "patch_size": 4, "in_chans": 4, "embed_dim": 96, "depths": [2, 2, 6, 2],
    "num_heads": [3, 6, 12, 24], "window_size": 8, "mlp_ratio": 4.0, "drop_rate": 0.0,
    "attn_drop_rate": 0.0, "drop_path_rate": 0.1, "decoder_channels": [512, 256, 128, 64],
}


def relative_position_index(ws: int) -> torch.Tensor:
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table (swin_unet_model.py:222-232)."""
    r = torch.arange(ws)
    cy, cx = torch.meshgrid(r, r, indexing="ij")
    cy, cx = cy.reshape(-1), cx.reshape(-1)
    dy = cy[:, None] - cy[None, :] + ws - 1
    dx = cx[:, None] - cx[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shift_mask(H: int, W: int, ws: int) -> torch.Tensor:
    """(nW, ws*ws, ws*ws) additive mask, -100 between different shift regions.

    swin_unet_model.py:459-479.  Built for every stage, also when the padded map is
    a single window.
    """
    s = ws // 2
    Hp = int(math.ceil(H / ws)) * ws
    Wp = int(math.ceil(W / ws)) * ws
    region = torch.zeros(Hp, Wp)
    bounds = (slice(0, -ws), slice(-ws, -s), slice(-s, None))
    n = 0
    for hs in bounds:
        for wsl in bounds:
            region[hs, wsl] = n
            n += 1
    win = region.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def _partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _unpartition(wins, ws, H, W):
    nh, nw = H // ws, W // ws
    if nh == 0 or nw == 0:
        raise ValueError(f"window_reverse: H ({H}) or W ({W}) too small for window {ws}")
    B = wins.shape[0] // (nh * nw)
    return wins.view(B, nh, nw, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def window_attention(sd, pre, x, heads, ws, mask):
    """swin_unet_model.py:241-272: softmax(q*scale @ k^T + bias[index] + mask) @ v, then proj."""
    Bn, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.view(Bn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    a = q @ k.transpose(-2, -1)
    idx = sd[pre + "relative_position_index"].view(-1)
    bias = sd[pre + "relative_position_bias_table"][idx].view(N, N, heads).permute(2, 0, 1)
    a = a + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        a = (a.view(Bn // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    a = a.softmax(-1)
    y = (a @ v).transpose(1, 2).reshape(Bn, N, C)
    return F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def swin_block(sd, pre, x, H, W, heads, ws, shift, mask, keep1: Optional[torch.Tensor] = None,
               keep2: Optional[torch.Tensor] = None):
    """swin_unet_model.py:305-374.

    ``keep1``/``keep2`` are the (B,) DropPath scales (mask / keep_prob) of the attention
    and MLP residual branches, or None for no drop.
    """
    B, L, C = x.shape
    assert L == H * W
    h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]).view(B, H, W, C)
    He, We = max(H, ws), max(W, ws)
    if He != H or We != W:     # zero-pad bottom/right AFTER the norm
        h = F.pad(h, (0, 0, 0, We - W, 0, He - H))
    if shift > 0:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    a = window_attention(sd, pre + "attn.", _partition(h, ws), heads, ws, mask)
    h = _unpartition(a, ws, He, We)
    if shift > 0:
        h = torch.roll(h, (shift, shift), (1, 2))
    h = h[:, :H, :W, :].reshape(B, L, C)
    k1 = 1.0 if keep1 is None else keep1.view(B, 1, 1)
    k2 = 1.0 if keep2 is None else keep2.view(B, 1, 1)
    x = x + h * k1
    y = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    y = F.linear(y, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[pre + "mlp.3.weight"], sd[pre + "mlp.3.bias"])
    return x + y * k2


def patch_merging(sd, pre, x, H, W):
    """swin_unet_model.py:386-421; concat order (0,0) (1,0) (0,1) (1,1) in (row, col) parity."""
    B, L, C = x.shape
    s = x.view(B, H, W, C)
    if H % 2 or W % 2:
        s = F.pad(s, (0, 0, 0, W % 2, 0, H % 2))
    m = torch.cat([s[:, 0::2, 0::2], s[:, 1::2, 0::2], s[:, 0::2, 1::2], s[:, 1::2, 1::2]], -1)
    m = m.reshape(B, -1, 4 * C)
    m = F.layer_norm(m, (4 * C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return F.linear(m, sd[pre + "reduction.weight"])


def _bn_relu(sd, pre, x, training, new_stats):
    rm, rv = sd[pre + "running_mean"].clone(), sd[pre + "running_var"].clone()
    x = F.batch_norm(x, rm, rv, sd[pre + "weight"], sd[pre + "bias"], training, 0.1, 1e-5)
    if training:
        new_stats[pre + "running_mean"], new_stats[pre + "running_var"] = rm, rv
    return F.relu(x)


def decoder_block(sd, pre, x, skip, training, new_stats):
    """swin_unet_model.py:520-535"""
    up = F.conv_transpose2d(x, sd[pre + "upsample.weight"], sd[pre + "upsample.bias"], stride=2)
    if skip is not None and (pre + "skip_projection.weight") in sd:
        sp = F.conv2d(skip, sd[pre + "skip_projection.weight"])
        if sp.shape[2:] != up.shape[2:]:
            sp = F.interpolate(sp, size=up.shape[2:], mode="bilinear", align_corners=False)
        up = torch.cat([up, sp], 1)
    y = F.conv2d(up, sd[pre + "conv_block.0.weight"], None, padding=1)
    y = _bn_relu(sd, pre + "conv_block.1.", y, training, new_stats)
    y = F.conv2d(y, sd[pre + "conv_block.3.weight"], None, padding=1)
    return _bn_relu(sd, pre + "conv_block.4.", y, training, new_stats)


def drop_path_rates(cfg) -> List[float]:
    """torch.linspace(0, rate, sum(depths)) (swin_unet_model.py:566)"""
    return [v.item() for v in torch.linspace(0, cfg["drop_path_rate"], sum(cfg["depths"]))]


def draw_drop_path(cfg, batch: int) -> List[torch.Tensor]:
    """Per-block (B,) scales in execution order, drawn as drop_path() does (:139-148).

    Each block calls drop_path twice (attention branch, then MLP branch); blocks
    with rate 0 use nn.Identity and draw nothing.
    """
    out = []
    for p in drop_path_rates(cfg):
        if p == 0.0:
            out.append([None, None])
            continue
        pair = []
        for _ in range(2):
            r = (1 - p) + torch.rand((batch, 1, 1))
            pair.append(r.floor_().view(batch) / (1 - p))
        out.append(pair)
    return out


def swin_unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: Dict = DEFAULT_CONFIG,
                      training: bool = False, drop_path=None):
    """reference swin_unet_model.py:632-686.  Returns (output (B,1,H/2,W/2), new BN stats)."""
    new_stats: Dict[str, torch.Tensor] = {}
    ws = cfg["window_size"]
    ps = cfg["patch_size"]
    t = F.conv2d(x, sd["patch_embed.weight"], sd["patch_embed.bias"], stride=ps)
    B, C, H, W = t.shape
    t = t.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (C,), sd["norm_embed.weight"], sd["norm_embed.bias"])
    skips = []
    nl = len(cfg["depths"])
    blk_no = 0
    for li in range(nl):
        mask = shift_mask(H, W, ws) if ws // 2 > 0 else None
        for bi in range(cfg["depths"][li]):
            shift = 0 if bi % 2 == 0 else ws // 2
            keep = None
            k2 = None
            if training and drop_path is not None:
                keep, k2 = drop_path[blk_no]
            pre = f"layers.{li}.blocks.{bi}."
            # the two DropPath draws of a block scale its two residual branches separately
            t = swin_block(sd, pre, t, H, W, cfg["num_heads"][li], ws, shift, mask, keep, k2)
            blk_no += 1
        skips.append(t.transpose(1, 2).reshape(B, -1, H, W))
        if li < nl - 1:
            t = patch_merging(sd, f"layers.{li}.downsample.", t, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    if H == 0 or W == 0:
        return torch.zeros(B, 1, x.shape[2], x.shape[3]), new_stats
    d = t.transpose(1, 2).reshape(B, -1, H, W)
    for i in range(len(cfg["decoder_channels"])):
        si = nl - 2 - i
        d = decoder_block(sd, f"decoders.{i}.", d, skips[si] if si >= 0 else None, training, new_stats)
    y = F.conv2d(d, sd["final_conv.0.weight"], sd["final_conv.0.bias"], padding=1)
    y = _bn_relu(sd, "final_conv.1.", y, training, new_stats)
    return F.conv2d(y, sd["final_conv.3.weight"], sd["final_conv.3.bias"]), new_stats


def swin_template(cfg: Dict = DEFAULT_CONFIG) -> Dict[str, torch.Tensor]:
    """Names and shapes of the SwinUNet state_dict (swin_unet_model.py:543-612)."""
    sd: Dict[str, torch.Tensor] = {}
    E, ws, ps = cfg["embed_dim"], cfg["window_size"], cfg["patch_size"]
    sd["patch_embed.weight"] = torch.empty(E, cfg["in_chans"], ps, ps)
    sd["patch_embed.bias"] = torch.empty(E)
    sd["norm_embed.weight"] = torch.empty(E)
    sd["norm_embed.bias"] = torch.empty(E)
    nl = len(cfg["depths"])
    dim = E
    hid = lambda d: int(d * cfg["mlp_ratio"])
    for li in range(nl):
        heads = cfg["num_heads"][li]
        for bi in range(cfg["depths"][li]):
            p = f"layers.{li}.blocks.{bi}."
            sd[p + "norm1.weight"] = torch.empty(dim)
            sd[p + "norm1.bias"] = torch.empty(dim)
            sd[p + "attn.relative_position_bias_table"] = torch.empty((2 * ws - 1) ** 2, heads)
            sd[p + "attn.relative_position_index"] = relative_position_index(ws)
            sd[p + "attn.qkv.weight"] = torch.empty(3 * dim, dim)
            sd[p + "attn.qkv.bias"] = torch.empty(3 * dim)
            sd[p + "attn.proj.weight"] = torch.empty(dim, dim)
            sd[p + "attn.proj.bias"] = torch.empty(dim)
            sd[p + "norm2.weight"] = torch.empty(dim)
            sd[p + "norm2.bias"] = torch.empty(dim)
            sd[p + "mlp.0.weight"] = torch.empty(hid(dim), dim)
            sd[p + "mlp.0.bias"] = torch.empty(hid(dim))
            sd[p + "mlp.3.weight"] = torch.empty(dim, hid(dim))
            sd[p + "mlp.3.bias"] = torch.empty(dim)
        if li < nl - 1:
            p = f"layers.{li}.downsample."
            sd[p + "reduction.weight"] = torch.empty(2 * dim, 4 * dim)
            sd[p + "norm.weight"] = torch.empty(4 * dim)
            sd[p + "norm.bias"] = torch.empty(4 * dim)
            dim *= 2
    cin = dim
    for i, co in enumerate(cfg["decoder_channels"]):
        p = f"decoders.{i}."
        up = cin // 2
        sd[p + "upsample.weight"] = torch.empty(cin, up, 2, 2)
        sd[p + "upsample.bias"] = torch.empty(up)
        si = nl - 2 - i
        cb_in = up
        if si >= 0:
            sd[p + "skip_projection.weight"] = torch.empty(up, E * 2 ** si, 1, 1)
            cb_in += up
        for conv_i, bn_i, c_in in ((0, 1, cb_in), (3, 4, co)):
            sd[p + f"conv_block.{conv_i}.weight"] = torch.empty(co, c_in, 3, 3)
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                sd[p + f"conv_block.{bn_i}.{leaf}"] = torch.empty(co)
            sd[p + f"conv_block.{bn_i}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        cin = co
    sd["final_conv.0.weight"] = torch.empty(32, cin, 3, 3)
    sd["final_conv.0.bias"] = torch.empty(32)
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        sd[f"final_conv.1.{leaf}"] = torch.empty(32)
    sd["final_conv.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    sd["final_conv.3.weight"] = torch.empty(1, 32, 1, 1)
    sd["final_conv.3.bias"] = torch.empty(1)
    return sd
