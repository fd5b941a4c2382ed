"""SwinUNet on hand-written gfx950 kernels.

Drop-in for the reference class (swin_unet_model.py:139-686): ``SwinUNet(config_dict)`` with the
same config keys, the same 249 ``state_dict`` entries in the same order, the same initialisation and
``forward(x: f32[B,in_chans,H,W]) -> f32[B,1,H/2,W/2]`` with autograd.  Tokens live as
(B,H,W,C) channel-innermost tensors in the compute dtype; every linear layer is the MFMA implicit-GEMM
kernel run as a 1x1 convolution (bias / residual / DropPath scale in its epilogue), window attention
is one fused kernel per direction, and the CNN decoder reuses the conv + BatchNorm + ReLU kernels of
the UNet.  Window size 8 and head dimension 32 (every stage of the reference configuration) are what
the attention kernel implements; other values raise ValueError.
"""
import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops

_WS = 8
# GELU' inside the epilogue of fc2's data gradient: on the wide token grids it saves a write and a read of the hidden
# gradient, on the deep stages a launch (measured per layer at batch 64: 55 against 36 + 32 us at 16x16 tokens)
_FUSE_GELU_BWD_ELEMS = int(os.environ.get("UNETK_SWIN_GELU_BWD_ELEMS", "0"))
FUSE_GELU_FWD = os.environ.get("UNETK_SWIN_FUSE_GELU_FWD", "1") != "0"
_GELU_FWD_MAX_TOKENS = 1 << 15


def hidden_small(x, blk):
    """Whether the block's token grid is one of the deep ones, where fc1 writes GELU as a second output."""
    return x.numel() // x.shape[-1] <= _GELU_FWD_MAX_TOKENS


# ------------------------------------------------------------------ parameter containers
class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size_tuple = (window_size, window_size) if isinstance(window_size, int) else window_size
        ws = self.window_size_tuple[0]
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        r = torch.arange(ws)
        cy, cx = torch.meshgrid(r, r, indexing="ij")
        cy, cx = cy.reshape(-1), cx.reshape(-1)
        idx = (cy[:, None] - cy[None, :] + ws - 1) * (2 * ws - 1) + (cx[:, None] - cx[None, :] + ws - 1)
        self.register_buffer("relative_position_index", idx)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, drop=0.,
                 attn_drop=0., drop_path=0.):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.mlp_ratio = mlp_ratio
        self.H = self.W = None
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        hid = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Dropout(drop), nn.Linear(hid, dim),
                                 nn.Dropout(drop))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, drop=0., attn_drop=0.,
                 drop_path=0., downsample=None):
        super().__init__()
        self.dim, self.depth, self.window_size = dim, depth, window_size
        self.shift_size = window_size // 2
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else self.shift_size, mlp_ratio,
                                 qkv_bias, drop, attn_drop,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)])
        self.downsample = downsample(dim=dim) if downsample is not None else None


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels):
        super().__init__()
        up = in_channels // 2
        self.upsample = nn.ConvTranspose2d(in_channels, up, kernel_size=2, stride=2)
        cin = up
        if skip_channels is not None and skip_channels > 0:
            self.skip_projection = nn.Conv2d(skip_channels, up, kernel_size=1, bias=False)
            cin += up
        else:
            self.skip_projection = None
        self.conv_block = nn.Sequential(
            nn.Conv2d(cin, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class SwinUNet(nn.Module):
    def __init__(self, config_dict, compute_dtype=torch.float32):
        super().__init__()
        c = config_dict
        self.patch_size_val, self.in_chans, self.embed_dim_val = c["patch_size"], c["in_chans"], c["embed_dim"]
        self.depths, self.num_heads, self.window_size = c["depths"], c["num_heads"], c["window_size"]
        self.mlp_ratio, self.drop_rate = c["mlp_ratio"], c["drop_rate"]
        self.attn_drop_rate, self.drop_path_rate = c["attn_drop_rate"], c["drop_path_rate"]
        self.num_layers = len(self.depths)
        self.patch_norm = True
        self.compute_dtype = compute_dtype
        # (window 8 with head dimension 32 - the reference configuration - runs on the MFMA attention kernels,
        # any other geometry the general kernels take on those: checked per stage below)
        if self.drop_rate != 0.0 or self.attn_drop_rate != 0.0:
            raise ValueError("element dropout inside attention / MLP (drop_rate, attn_drop_rate) is not "
                             "implemented; the reference configuration uses 0.0")
        self.patch_embed = nn.Conv2d(self.in_chans, self.embed_dim_val, kernel_size=self.patch_size_val,
                                     stride=self.patch_size_val)
        self.norm_embed = nn.LayerNorm(self.embed_dim_val)
        self.pos_drop = nn.Dropout(p=self.drop_rate)
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, sum(self.depths))]
        self.layers = nn.ModuleList()
        dim = self.embed_dim_val
        for i in range(self.num_layers):
            if dim % self.num_heads[i] or not _lib.load().unetk_window_attn_any_supported(
                    int(self.window_size), dim // self.num_heads[i]):
                raise ValueError(f"window attention with window_size {self.window_size} and head dimension "
                                 f"{dim / self.num_heads[i]:g} (stage {i}: {dim} channels, {self.num_heads[i]} heads) is "
                                 "not implemented: window side 2...16, head dimension up to 64, window tokens x head "
                                 "dimension up to 9216")
            self.layers.append(BasicLayer(
                dim, self.depths[i], self.num_heads[i], self.window_size, self.mlp_ratio, True, self.drop_rate,
                self.attn_drop_rate, dpr[sum(self.depths[:i]):sum(self.depths[:i + 1])],
                PatchMerging if i < self.num_layers - 1 else None))
            if i < self.num_layers - 1:
                dim = int(dim * 2)
        self.decoder_channels_conf = c["decoder_channels"]
        self.decoders = nn.ModuleList()
        cin = dim
        for i, co in enumerate(self.decoder_channels_conf):
            si = self.num_layers - 2 - i
            self.decoders.append(DecoderBlock(cin, int(self.embed_dim_val * 2 ** si) if si >= 0 else None, co))
            cin = co
        self.final_conv = nn.Sequential(nn.Conv2d(cin, 32, kernel_size=3, padding=1), nn.BatchNorm2d(32),
                                        nn.ReLU(inplace=True), nn.Conv2d(32, 1, kernel_size=1))
        self.apply(self._init_weights)
        # test hook: per block [keep1, keep2] (B,) DropPath scales (or None) used instead of fresh draws
        self.drop_path_override: Optional[List] = None
        # data-parallel hook, see unet_amd.dist.GradReducer
        self.grad_sink = None

    def _init_weights(self, m):
        # reference swin_unet_model.py:616-630 (ConvTranspose2d is not a Conv2d: keeps its default init)
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def late_grad_names(self):
        """Parameters whose gradients the backward pass completes with its closing launches (the LayerNorm weight /
        bias sums and the relative-position table sums of every block, taken together at the end): a data-parallel
        reducer keeps them in a bucket of their own."""
        if not DEFER_LN_SUMS:
            return []
        return [n for n, _ in _named_params(self)
                if n.endswith("relative_position_bias_table") or ".norm" in n or n.startswith("norm_embed.")]

    def _draw_drop_path(self, B, device):
        if self.drop_path_override is not None:
            return [[None if k is None else k.to(device=device, dtype=torch.float32).contiguous() for k in pair]
                    for pair in self.drop_path_override]
        # floor(keep_prob + U[0,1)) / keep_prob per sample (:139-148), once per branch: all blocks drawn together
        # (four small launches per step instead of eight per block)
        probs = [blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0
                 for layer in self.layers for blk in layer.blocks]
        live = [i for i, p in enumerate(probs) if p]
        if not live:
            return [[None, None] for _ in probs]
        cache = getattr(self, "_dp_keep_cache", None)
        key = (str(device), tuple(probs))
        if cache is None or cache[0] != key:
            keep = torch.tensor([1.0 - probs[i] for i in live for _ in range(2)], dtype=torch.float32).view(-1, 1)
            cache = self._dp_keep_cache = (key, keep.to(device))
        keep = cache[1]
        k = (torch.rand(keep.shape[0], B, device=device) + keep).floor_().div_(keep)
        out = [[None, None] for _ in probs]
        for j, i in enumerate(live):
            out[i] = [k[2 * j], k[2 * j + 1]]
        return out

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("unet_amd.SwinUNet runs on the GPU only (hand-written gfx950 kernels, no CPU "
                               "fallback); move the module and its input to 'cuda'")
        if x.dim() != 4 or x.shape[1] != self.in_chans:
            raise RuntimeError(f"expected input (B,{self.in_chans},H,W), got {tuple(x.shape)}")
        ps = self.patch_size_val
        Hp, Wp = x.shape[2] // ps, x.shape[3] // ps
        for _ in range(self.num_layers - 1):
            Hp, Wp = (Hp + 1) // 2, (Wp + 1) // 2
        if Hp == 0 or Wp == 0:     # reference fallback (:664-668)
            return torch.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), device=x.device, dtype=x.dtype)
        if self.training:
            dp = self._draw_drop_path(x.shape[0], x.device)
            return _SwinTrainFn.apply(x, self, dp, *[p for _, p in _named_params(self)])
        with torch.no_grad():
            return _forward(self, x, None, None)


def _named_params(model):
    """list(model.named_parameters()), kept on the model: the walk over the module tree costs the host ~0.5 ms and
    runs twice per step; the set of parameters of a built model does not change."""
    hit = model.__dict__.get("_named_params_list")
    if hit is None:
        hit = model.__dict__["_named_params_list"] = list(model.named_parameters())
    return hit


# ------------------------------------------------------------------ packed weights
class _Ctx:
    """What one forward (and its backward) carries around: the compute dtype and the packed forms of
    every Linear / Conv2d weight, made by ONE launch at the start of the forward (forward form, and
    for training the data-gradient form) and looked up by the parameter's address; anything not
    found there (a reshaped copy) is packed on the spot.  Owned by the autograd node of the step
    (or dropped after an inference call): interleaved models or steps cannot see each other's packs."""
    __slots__ = ("dt", "pool", "side", "fork", "held", "tracked", "sums")

    def __init__(self, model, training):
        self.dt = dt = model.compute_dtype
        self.side = None        # set by the backward pass: the stream the weight gradients run on
        self.fork = None
        self.held = []
        self.tracked = []       # BatchNorm step counters of this forward: advanced by one launch at its end
        self.sums = None        # set by the backward pass: the LayerNorm parameter-gradient sums taken at its end
        # (the walk over the module tree costs the host ~1 ms: its result is kept on the model, which does not
        # grow layers between steps; what can change - device, dtype, layout of a weight - is checked every time)
        cand = model.__dict__.get("_packable_modules")
        if cand is None:
            cand = model.__dict__["_packable_modules"] = [m for m in model.modules() if isinstance(m, (nn.Linear, nn.Conv2d))]
            model.__dict__["_convt_modules"] = [m for m in model.modules() if isinstance(m, nn.ConvTranspose2d) and
                                                tuple(m.kernel_size) == (2, 2) and tuple(m.stride) == (2, 2)]
        # which weights are packed in which forms depends on the model's tensors only (their addresses change when
        # it is moved or cast): worked out once per (mode, type, addresses), ~1 ms of list building per step otherwise
        allw = [m.weight for m in cand] + [m.weight for m in model.__dict__.get("_convt_modules", ())]
        key = (training, dt, FUSED_MLP, FUSED_MLP_BWD, tuple(w.data_ptr() for w in allw))
        hit = model.__dict__.get("_pack_plan")
        if hit is None or hit[0] != key:
            ws = []
            for m in cand:
                w = m.weight
                if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and \
                        (w.dim() == 2 or (w.shape[2] == w.shape[3] and w.shape[2] in (1, 3))):
                    ws.append(w)
            # the ConvTranspose2d(k=2, s=2) weights of the decoder go into the same launch, as the GEMM matrices they
            # are used as (else: a permuted copy and a pack of its own per decoder block, forward and backward)
            wt = [m.weight for m in model.__dict__.get("_convt_modules", ())
                  if m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()]
            # fc1 of the blocks whose MLP branch runs as one launch: 64-channel blocks
            w64 = []
            if FUSED_MLP and dt == torch.bfloat16:
                w64 = [blk.mlp[0].weight for layer in getattr(model, "layers", ()) for blk in layer.blocks
                       if blk.dim in (96, 192) and blk.mlp[0].weight.shape == (4 * blk.dim, blk.dim) and
                       any(blk.mlp[0].weight is w for w in ws)]
            # ... and, for the backward pass of that launch, fc2 in its data-gradient form with 64-channel blocks
            w64t = []
            if w64 and training and FUSED_MLP_BWD:
                w64t = [blk.mlp[3].weight for layer in model.layers for blk in layer.blocks
                        if any(blk.mlp[0].weight is w for w in w64)]
            hit = model.__dict__["_pack_plan"] = (key, ws, wt, w64, w64t)
        _, ws, wt, w64, w64t = hit
        forms = (False, True) if training else (False,)
        packs = ops.pack_many([(w, tr) for tr in forms for w in ws] +
                              [(w, "convt_t" if tr else "convt") for tr in forms for w in wt] +
                              [(w, "bn64") for w in w64] + [(w, "bn64_t") for w in w64t], dt)
        self.pool = {}
        base = len(forms) * (len(ws) + len(wt))
        for j, w in enumerate(w64):
            self.pool[(w.data_ptr(), "bn64")] = (packs[base + j], w._version)
        for j, w in enumerate(w64t):
            self.pool[(w.data_ptr(), "bn64_t")] = (packs[base + len(w64) + j], w._version)
        for i, tr in enumerate(forms):
            for j, w in enumerate(ws):
                self.pool[(w.data_ptr(), tr)] = (packs[i * len(ws) + j], w.shape[0], w.shape[1],
                                                 1 if w.dim() == 2 else w.shape[2], w._version)
            for j, w in enumerate(wt):
                self.pool[(w.data_ptr(), tr, "convt")] = (packs[len(forms) * len(ws) + i * len(wt) + j], w._version)

    def packed_convt(self, w, transpose=False):
        """The pack of a ConvTranspose2d(k=2, s=2) weight (Cin, up, 2, 2) as its (4*up, Cin) GEMM matrix (or that
        matrix's data-gradient form), from the step's one launch when it is there."""
        hit = self.pool.get((w.data_ptr(), bool(transpose), "convt"))
        if hit is not None:
            if hit[1] != w._version:
                raise RuntimeError("a SwinUNet parameter was modified in place between the forward pass and its "
                                   "backward pass (the packed copies made by the forward no longer match)")
            return hit[0]
        up, Cin = w.shape[1], w.shape[0]
        w2 = w.permute(2, 3, 1, 0).reshape(4 * up, Cin)       # [(ky,kx,co)][ci]
        return ops.PackedConv(w2, self.dt, transpose=transpose)

    def packed64(self, w, form="bn64"):
        """fc1 of a fused MLP branch in 64-channel blocks (form "bn64_t": fc2 in its data-gradient form), or None
        when this step did not pack it."""
        hit = self.pool.get((w.data_ptr(), form))
        if hit is None or hit[1] != w._version:
            return None
        return hit[0]

    def packed(self, w, transpose=False):
        hit = self.pool.get((w.data_ptr(), bool(transpose)))
        ks = 1 if w.dim() == 2 else w.shape[2]
        if hit is not None and hit[1:4] == (w.shape[0], w.shape[1], ks):
            if hit[4] != w._version:
                raise RuntimeError("a SwinUNet parameter was modified in place between the forward pass and its "
                                   "backward pass (the packed copy made by the forward no longer matches it)")
            return hit[0]
        return ops.PackedConv(w, self.dt, transpose=transpose)


# Weight gradients are off the critical chain of the backward pass (data gradient -> norm backward ->
# data gradient ...).  The Swin encoder's late stages launch many small kernels (512-768 workgroups, one
# round of the chip or less): run on a second HIP stream, their weight gradients fill the CUs the
# chain leaves idle at every kernel's ramp-up and tail.  A data-parallel gradient sink receives each of those
# gradients on that second stream (unet_amd.dist.GradReducer orders its buckets' collectives behind both streams).
OVERLAP_WGRAD = os.environ.get("UNETK_SWIN_OVERLAP_WGRAD", "1") != "0"
# the decoder's data-gradient convs leave the BatchNorm-backward sums of the unit in front (see UNet's backward)
FUSE_BN_SUMS = os.environ.get("UNETK_FUSE_BN_SUMS", "1") != "0"
BN_SUMS_MAX_BYTES = int(os.environ.get("UNETK_BN_SUMS_MAX_BYTES", str(64 << 20)))
DEFER_LN_SUMS = os.environ.get("UNETK_SWIN_DEFER_LN_SUMS", "1") != "0"     # LayerNorm dgamma / dbeta sums: one launch at the end of the backward
CONVT_BWD_GATHER = os.environ.get("UNETK_SWIN_CONVT_BWD_GATHER", "1") != "0"   # ConvTranspose data gradient reads the gradient image in place
# The MLP branch of the 96- and 192-channel blocks as ONE launch (LayerNorm, fc1, GELU, fc2, residual): the hidden
# tensor no longer travels through HBM between four kernels.  With FUSED_MLP_SAVE_ACT off the GELU output is not
# stored either: the backward pass recomputes it from the pre-activation next to the weight gradient that reads it
# (on that gradient's stream, off the data-gradient chain).
FUSED_MLP = os.environ.get("UNETK_SWIN_FUSED_MLP", "1") != "0"
FUSED_MLP_SAVE_ACT = os.environ.get("UNETK_SWIN_MLP_SAVE_ACT", "1") != "0"
# ... and its data-gradient chain (fc2's data gradient, GELU', fc1's data gradient, LayerNorm backward) as one launch
FUSED_MLP_BWD = os.environ.get("UNETK_SWIN_FUSED_MLP_BWD", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return st


class _Fork:
    """What _on_side needs of the two streams of a backward pass, looked up once: raw handles for the library's
    fork call and the ids torch's stream switch takes.  (The framework's own route - current_stream(), an Event,
    record, wait_event, the stream context manager - costs the host ~50 us per job, 72 jobs per step.)"""
    __slots__ = ("main", "side", "main_raw", "side_raw", "main_ids", "side_ids")

    def __init__(self, main, side):
        self.main, self.side = main, side
        self.main_raw, self.side_raw = _lib.c_void_p(main.cuda_stream), _lib.c_void_p(side.cuda_stream)
        self.main_ids = (main.stream_id, main.device_index, main.device_type)
        self.side_ids = (side.stream_id, side.device_index, side.device_type)


def _on_side(cx, fn, *operands):
    """Runs fn() (weight-gradient launches reading ``operands``) on the side stream, after everything queued
    so far on the current stream; the operands stay allocated until the streams are joined.  Returns fn()'s
    result, whose tensors are marked as used by the main stream too."""
    if cx.side is None:
        return fn()
    fk = cx.fork
    main = fk.main
    _lib.call("unetk_stream_fork", fk.main_raw, fk.side_raw)
    cx.held.append(operands)
    torch._C._cuda_setStream(*fk.side_ids)
    try:
        res = fn()
    finally:
        torch._C._cuda_setStream(*fk.main_ids)
    for t in (res if isinstance(res, tuple) else (res,)):
        if t is not None:
            t.record_stream(main)
            ops.on_stream(t, cx.side)
    return res


def _linear_launch(x, pw, shift=None, act=0, resid=None, resid_keep=None, out=None, out2=False, gelu_grad_of=None):
    """Every token-linear GEMM launch (forward and data gradient) goes through here: the GEMM kernel
    where it applies (bfloat16, regular shapes), else the 1x1 form of the convolution kernel with the
    GELU passes as separate launches."""
    if ops.linear_supported(x, pw):
        return ops.linear_fwd(x, pw, bias=shift, act=act, resid=resid, resid_keep=resid_keep, out=out, out2=out2,
                              gelu_grad_of=gelu_grad_of)
    if out2:        # (pre-activation, activation)
        pre = ops.conv_fwd(x, pw, scale=None, shift=shift, act=0, out=out)[0]
        return pre, ops.gelu_fwd(pre)
    y = ops.conv_fwd(x, pw, scale=None, shift=shift, act=act, resid=resid, resid_keep=resid_keep, out=out)[0]
    if gelu_grad_of is not None:
        y = ops.gelu_bwd(y, gelu_grad_of)
    return y


# ------------------------------------------------------------------ building blocks
def _linear(x, w, b, cx, act=0, resid=None, keep=None, out=None, out2=False):
    return _linear_launch(x, cx.packed(w), shift=b, act=act, resid=resid, resid_keep=keep, out=out, out2=out2)


def _linear_bwd(x, dy, w, cx, grads, wname, bname, need_dx=True, gelu_grad_of=None, x_from=None):
    """x_from: x is not stored; it is GELU(x_from), recomputed on the stream of the weight gradient that reads it."""
    Co, Ci = w.shape[0], w.shape[1]
    if x is None:
        dw, db = _on_side(cx, lambda: ops.linear_wgrad(ops.gelu_fwd(x_from), dy, Ci, Co, want_bias=bname is not None),
                          x_from, dy)
    else:
        dw, db = _on_side(cx, lambda: ops.linear_wgrad(x, dy, Ci, Co, want_bias=bname is not None), x, dy)
    if bname is not None:
        grads[bname] = db
    grads[wname] = ops.view_as_param(dw, w.shape)
    if need_dx:
        return _linear_launch(dy, cx.packed(w.view(Co, Ci), transpose=True), gelu_grad_of=gelu_grad_of)
    return None


def _check_grid(H, W, ws=_WS):
    for n in (H, W):
        if n > ws and n % ws:
            raise RuntimeError(f"shape is invalid for window partition: a {H}x{W} token grid is not a multiple "
                               f"of the window size {ws}")


def _pad_tokens(t, He, We):
    B, H, W, C = t.shape
    if (He, We) == (H, W):
        return t
    p = torch.zeros((B, He, We, C), dtype=t.dtype, device=t.device)
    p[:, :H, :W] = t
    return p


def _block_fwd(blk, pre, x, keep1, keep2, cx, save):
    """One SwinTransformerBlock (:305-374) on tokens x (B,H,W,C)."""
    B, H, W, C = x.shape
    ws = blk.window_size
    _check_grid(H, W, ws)
    He, We = max(H, ws), max(W, ws)
    padded = (He, We) != (H, W)
    a = blk.attn
    ln1, m1, r1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    ln1p = _pad_tokens(ln1, He, We)
    qkv = _linear(ln1p, a.qkv.weight, a.qkv.bias, cx)
    # the shift mask is built for every stage and handed to every block, shifted or not (:458-484)
    att = ops.window_attn_fwd(qkv, a.relative_position_bias_table, a.num_heads, blk.shift_size, True, ws=ws)
    if not padded:
        x1 = _linear(att, a.proj.weight, a.proj.bias, cx, resid=x, keep=keep1)
    else:
        pr = _linear(att, a.proj.weight, a.proj.bias, cx)
        x1 = ops.scale_sample(pr[:, :H, :W].contiguous(), keep1, x)
    p64 = cx.packed64(blk.mlp[0].weight) if FUSED_MLP else None
    p96 = cx.packed(blk.mlp[3].weight) if p64 is not None else None
    if p64 is not None and p96.bn == 96 and blk.mlp[0].bias is not None and blk.mlp[3].bias is not None and \
            ops.swin_mlp_supported(x1):
        x2, ln2, m2, r2, hpre, hact = ops.swin_mlp_fwd(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, p64,
                                                        blk.mlp[0].bias, p96, blk.mlp[3].bias, keep=keep2,
                                                        want_act=FUSED_MLP_SAVE_ACT or save is None)
    else:
        ln2, m2, r2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        if FUSE_GELU_FWD and hidden_small(x1, blk):
            # the deep stages (few tokens, wide layers): GELU as a second output of fc1's epilogue saves a launch
            # in a part of the step where launches, not bytes, set the time
            hpre, hact = _linear(ln2, blk.mlp[0].weight, blk.mlp[0].bias, cx, act=2, out2=True)
        else:
            # (on the wide token grids fc1 writing both tensors is bound by the same two writes: measured equal or slower)
            hpre = _linear(ln2, blk.mlp[0].weight, blk.mlp[0].bias, cx)
            hact = ops.gelu_fwd(hpre)
        x2 = _linear(hact, blk.mlp[3].weight, blk.mlp[3].bias, cx, resid=x1, keep=keep2)
    if save is not None:
        save.append(("block", blk, pre, (x, m1, r1, ln1p, qkv, att, x1, m2, r2, ln2, hpre, hact, keep1, keep2,
                                          (H, W, He, We))))
    return x2


def _block_bwd(blk, pre, ctx, dx2, cx, grads, db2=None, keep_next=None):
    """dx2: gradient of the block's output; db2: the same times this block's second DropPath factors when the
    producer of dx2 wrote it (else formed here); keep_next: the second DropPath factors of the block whose
    backward runs next.  Returns (dx, dx times keep_next or None)."""
    x, m1, r1, ln1p, qkv, att, x1, m2, r2, ln2, hpre, hact, keep1, keep2, (H, W, He, We) = ctx
    a = blk.attn
    padded = (He, We) != (H, W)
    # the DropPath scaling of a gradient is a second output of the LayerNorm backward that produces it
    # (one launch less per branch), where that backward runs in its one-pass form
    fuse = cx.sums is not None and not padded
    # MLP branch
    if db2 is not None:
        db = db2
    else:
        db = ops.scale_sample(dx2, keep2) if keep2 is not None else dx2
    p64t = cx.packed64(blk.mlp[3].weight, "bn64_t") if (FUSED_MLP_BWD and cx.sums is not None and not padded) else None
    if p64t is not None and ops.swin_mlp_supported(x1) and db.is_contiguous():
        # the whole data-gradient chain of the branch in one launch; the two weight gradients read db and its dhp
        w1, w2 = blk.mlp[0].weight, blk.mlp[3].weight
        dhp, dx1, grads[pre + "norm2.weight"], grads[pre + "norm2.bias"], dxs = ops.swin_mlp_bwd(
            db, hpre, p64t, cx.packed(w1, transpose=True), x1, m2, r2, blk.norm2.weight, dx2, cx.sums,
            scale_next=(keep1, H * W) if keep1 is not None else None)
        _linear_bwd(hact, db, w2, cx, grads, pre + "mlp.3.weight", pre + "mlp.3.bias", need_dx=False, x_from=hpre)
        _linear_bwd(ln2, dhp, w1, cx, grads, pre + "mlp.0.weight", pre + "mlp.0.bias", need_dx=False)
        res = (dx1, None, None, dxs)
    else:
        # d hidden = (d out @ W2) * GELU'(pre-activation): inside the GEMM's epilogue where the hidden tensor is
        # large (it saves one write and one read of it); on the small late stages the separate pass is as fast
        if hpre.numel() >= _FUSE_GELU_BWD_ELEMS:
            dhp = _linear_bwd(hact, db, blk.mlp[3].weight, cx, grads, pre + "mlp.3.weight", pre + "mlp.3.bias",
                              gelu_grad_of=hpre, x_from=hpre)
        else:
            dh = _linear_bwd(hact, db, blk.mlp[3].weight, cx, grads, pre + "mlp.3.weight", pre + "mlp.3.bias",
                             x_from=hpre)
            dhp = ops.gelu_bwd(dh, hpre)
        dln2 = _linear_bwd(ln2, dhp, blk.mlp[0].weight, cx, grads, pre + "mlp.0.weight", pre + "mlp.0.bias")
        res = ops.layernorm_bwd(dln2, x1, blk.norm2.weight, m2, r2, add=dx2, defer=cx.sums,
                                scale_next=(keep1, H * W) if (fuse and keep1 is not None) else None)
        dx1, grads[pre + "norm2.weight"], grads[pre + "norm2.bias"] = res[:3]
    # attention branch
    if len(res) == 4 and res[3] is not None:
        db = res[3]
    else:
        db = ops.scale_sample(dx1, keep1) if keep1 is not None else dx1
    dbp = _pad_tokens(db, He, We) if padded else db
    datt = _linear_bwd(att, dbp, a.proj.weight, cx, grads, pre + "attn.proj.weight", pre + "attn.proj.bias")
    dqkv, grads[pre + "attn.relative_position_bias_table"] = ops.window_attn_bwd(
        qkv, datt, a.relative_position_bias_table, a.num_heads, blk.shift_size, True, defer=cx.sums, ws=blk.window_size)
    dln1 = _linear_bwd(ln1p, dqkv, a.qkv.weight, cx, grads, pre + "attn.qkv.weight", pre + "attn.qkv.bias")
    if padded:
        dln1 = dln1[:, :H, :W].contiguous()
    res = ops.layernorm_bwd(dln1, x, blk.norm1.weight, m1, r1, add=dx1, defer=cx.sums,
                            scale_next=(keep_next, H * W) if (fuse and keep_next is not None) else None)
    dx, grads[pre + "norm1.weight"], grads[pre + "norm1.bias"] = res[:3]
    return dx, (res[3] if len(res) == 4 else None)


def _conv_bn_relu_fwd(conv, bn, t, cx, training, save, tag):
    """Conv3x3 -> BatchNorm2d -> ReLU.  A conv bias in front of BatchNorm only moves running_mean."""
    pk = cx.packed(conv.weight)
    if not training:
        sc, sh = ops.bn_fold_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias, bn.eps)
        return ops.conv_fwd(t, pk, scale=sc, shift=sh, act=1)[0]
    y, stats = ops.conv_fwd(t, pk, want_stats=True)
    Bn, H, W, _ = y.shape
    mean, invstd, sc, sh = ops.bn_finalize(stats, Bn * H * W, bn.weight, bn.bias, bn.running_mean,
                                           bn.running_var, bn.eps, bn.momentum if bn.momentum is not None else 0.1)
    if conv.bias is not None:
        bn.running_mean.add_(conv.bias.detach(), alpha=bn.momentum if bn.momentum is not None else 0.1)
    cx.tracked.append(bn.num_batches_tracked)
    a, _ = ops.bn_act_fwd(y, sc, sh, None, relu=True)
    if save is not None:
        save.append(("cbr", conv, bn, tag, (t, y, mean, invstd, sc, sh)))
    return a


def _conv_bn_relu_bwd(conv, bn, tag, ctx, g, cx, grads, need_dx=True, feeds=None, presummed=None):
    """``feeds``: the saved tensors of the Conv-BN-ReLU unit whose activation this conv read, when the gradient goes
    straight into that unit's BatchNorm backward: the data-gradient conv then leaves that pass's two sums, returned as
    the second value (None otherwise) and handed back in as ``presummed`` with the next call."""
    t, y, mean, invstd, sc, sh = ctx
    dy, dgamma, dbeta = ops.bn_act_bwd(g, y, sc, sh, mean, invstd, None, relu=True, presummed=presummed)
    cname, bname = tag
    grads[bname + ".weight"], grads[bname + ".bias"] = dgamma, dbeta
    grads[cname + ".weight"] = _on_side(cx, lambda: ops.conv_wgrad(t, dy, conv.in_channels, conv.out_channels, 3), t, dy)
    if conv.bias is not None:
        grads[cname + ".bias"] = torch.zeros_like(conv.bias)      # exactly 0 in front of BatchNorm
    if not need_dx:
        return None, None
    pwT = cx.packed(conv.weight, transpose=True)
    if feeds is not None and FUSE_BN_SUMS and feeds[1].shape == t.shape and \
            t.numel() * t.element_size() <= BN_SUMS_MAX_BYTES and ops.conv_bwd_data_bnsums_supported(pwT, cx.dt):
        _, y0, mean0, invstd0, sc0, sh0 = feeds
        return ops.conv_bwd_data_bnsums(dy, pwT, y0, sc0, sh0, mean0, invstd0)
    return ops.conv_fwd(dy, pwT)[0], None


def _decoder_fwd(dec, pre, x, skip, cx, training, save):
    """DecoderBlock (:520-535): ConvTranspose(2,2)+bias, 1x1 skip projection, concat, DoubleConv."""
    B, h, w, Cin = x.shape
    up = dec.upsample.out_channels
    has_skip = skip is not None and dec.skip_projection is not None
    cat = torch.empty((B, 2 * h, 2 * w, up * (2 if has_skip else 1)), dtype=cx.dt, device=x.device)
    pk2 = cx.packed_convt(dec.upsample.weight)      # the (4*up, Cin) matrix [(ky,kx,co)][ci]
    if ops.linear_supported(x, pk2) and up % 8 == 0 and x.is_contiguous():
        ops.convT2x2_fwd(x, pk2, dec.upsample.bias, cat)          # the pixel shuffle is the GEMM's store address
    else:
        tmp = _linear_launch(x, pk2)
        ops.pixel_shuffle(tmp, cat, dec.upsample.bias, up)
    resized = None
    if has_skip:
        if skip.shape[1:3] == (2 * h, 2 * w):
            _linear_launch(skip, cx.packed(dec.skip_projection.weight), out=cat[..., up:])
        else:       # bilinear resize of the projected skip to the upsampled size (:526-529)
            sp = _linear(skip, dec.skip_projection.weight.view(up, -1), None, cx)
            resized = sp.shape[1:3]
            cat[..., up:] = ops.upsample_concat(sp, torch.empty((B, 2 * h, 2 * w, 0), dtype=cx.dt, device=x.device))
    if save is not None:
        save.append(("dec_head", dec, pre, (x, skip, None, resized)))
    t = _conv_bn_relu_fwd(dec.conv_block[0], dec.conv_block[1], cat, cx, training, save,
                          (pre + "conv_block.0", pre + "conv_block.1"))
    return _conv_bn_relu_fwd(dec.conv_block[3], dec.conv_block[4], t, cx, training, save,
                             (pre + "conv_block.3", pre + "conv_block.4"))


def _decoder_head_bwd(dec, pre, ctx, dcat, cx, grads):
    """Returns (dx, dskip or None)."""
    x, skip, _, resized = ctx
    B, h, w, Cin = x.shape
    up = dec.upsample.out_channels
    grads[pre + "upsample.bias"] = _on_side(cx, lambda: ops.channel_sum(dcat, up), dcat)     # (a leaf: off the chain)
    pkT = cx.packed_convt(dec.upsample.weight, transpose=True)
    # the data gradient gathers its operand rows from dcat (the inverse pixel shuffle is the GEMM's load address);
    # the un-shuffled token matrix is then only the weight gradient's operand, made on its stream
    dx = ops.convT2x2_bwd_data(dcat, up, pkT, h, w) if CONVT_BWD_GATHER else None
    if dx is not None:
        def wjob():
            dtmp = torch.empty((B, h, w, 4 * up), dtype=cx.dt, device=x.device)
            ops.pixel_shuffle(dtmp, dcat, None, up, backward=True)
            return ops.linear_wgrad(x, dtmp, Cin, 4 * up, want_bias=False)[0].view(2, 2, up, Cin) \
                .permute(3, 2, 0, 1).contiguous()
        grads[pre + "upsample.weight"] = _on_side(cx, wjob, x, dcat)
    else:
        dtmp = torch.empty((B, h, w, 4 * up), dtype=cx.dt, device=x.device)
        ops.pixel_shuffle(dtmp, dcat, None, up, backward=True)
        grads[pre + "upsample.weight"] = _on_side(
            cx, lambda: ops.linear_wgrad(x, dtmp, Cin, 4 * up, want_bias=False)[0].view(2, 2, up, Cin)
            .permute(3, 2, 0, 1).contiguous(), x, dtmp)
        dx = _linear_launch(dtmp, pkT)
    dskip = None
    if skip is not None and dec.skip_projection is not None:
        wsk = dec.skip_projection.weight
        dsp = dcat[..., up:]
        if resized is not None:
            dsp = ops.upsample_bwd(dsp.contiguous(), up, resized[0], resized[1])
        grads[pre + "skip_projection.weight"] = ops.view_as_param(_on_side(
            cx, lambda: ops.linear_wgrad(skip, dsp, wsk.shape[1], up, want_bias=False)[0], skip, dsp), wsk.shape)
        dskip = _linear_launch(dsp, cx.packed(wsk, transpose=True))
    return dx, dskip


def _forward(model, x, drop_path, save):
    """Whole network.  save=None: inference (BatchNorm folded, nothing kept)."""
    training = save is not None
    cx = _Ctx(model, training)
    dt = cx.dt
    if training:
        save.append(("packs", cx))
    ps = model.patch_size_val
    E = model.embed_dim_val
    tok_in = ops.patchify(x.detach(), ps, dt)
    wpe = model.patch_embed.weight.reshape(E, -1)
    t0 = _linear(tok_in, wpe, model.patch_embed.bias, cx)
    t, m0, r0 = ops.layernorm_fwd(t0, model.norm_embed.weight, model.norm_embed.bias, model.norm_embed.eps)
    if training:
        save.append(("embed", (tok_in, t0, m0, r0)))
    skips = []
    bno = 0
    for li, layer in enumerate(model.layers):
        for bi, blk in enumerate(layer.blocks):
            k1, k2 = drop_path[bno] if (training and drop_path is not None) else (None, None)
            t = _block_fwd(blk, f"layers.{li}.blocks.{bi}.", t, k1, k2, cx, save)
            bno += 1
        skips.append(t)
        if layer.downsample is not None:
            ds = layer.downsample
            mg = ops.merge_gather(t)
            ln, mm, rr = ops.layernorm_fwd(mg, ds.norm.weight, ds.norm.bias, ds.norm.eps)
            nxt = _linear(ln, ds.reduction.weight, None, cx)
            if training:
                save.append(("merge", ds, f"layers.{li}.downsample.", li, (t.shape, mg, mm, rr, ln)))
            t = nxt
    d = t
    nl = model.num_layers
    for i, dec in enumerate(model.decoders):
        si = nl - 2 - i
        d = _decoder_fwd(dec, f"decoders.{i}.", d, skips[si] if si >= 0 else None, cx, training, save)
        if training:
            save.append(("dec_tail", i, si))
    fc = model.final_conv
    f = _conv_bn_relu_fwd(fc[0], fc[1], d, cx, training, save, ("final_conv.0", "final_conv.1"))
    out = ops.head1x1_fwd(f, fc[3].weight, fc[3].bias)
    if cx.tracked:
        torch._foreach_add_(cx.tracked, 1)
        cx.tracked = []
    if training:
        save.append(("head", f))
    return out


class _SwinTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, drop_path, *params):
        save: List = []
        out = _forward(model, x, drop_path, save)
        ctx.model, ctx.save, ctx.in_shape = model, save, x.shape
        ctx.need_dx = ctx.needs_input_grad[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        model, save = ctx.model, ctx.save
        if save is None:
            raise RuntimeError("the SwinUNet step was already back-propagated: its saved activations are freed "
                               "by the first backward (run the forward again)")
        cx = save[0][1]                 # dtype and the packs made by this step's forward
        main = torch.cuda.current_stream()
        cx.side = _side_stream(dout.device) if OVERLAP_WGRAD else None
        cx.fork = _Fork(main, cx.side) if cx.side is not None else None
        # dgamma / dbeta of the 25 LayerNorms: their partial tables are summed by one launch at the end
        # (not while a HIP graph is being captured: the job table would be uploaded from host memory inside the graph)
        cx.sums = ops.DeferredSums() if (DEFER_LN_SUMS and not torch.cuda.is_current_stream_capturing()) else None
        # a gradient sink (data-parallel buckets) gets every gradient on the stream that produced it, and the ones
        # the closing launch completes after that launch
        grads = ops.GradOut(model.grad_sink, model.late_grad_names() if cx.sums is not None else ())
        nl = model.num_layers
        dskips: Dict[int, torch.Tensor] = {}
        g = None
        pending_skip = None
        g_scaled = None                  # g times the next block's DropPath factors, when its producer wrote it
        presummed = None                 # BatchNorm-backward sums of the next unit, when g's producer left them
        for ri in range(len(save) - 1, -1, -1):
            rec = save[ri]
            kind = rec[0]
            if kind != "block":
                g_scaled = None
            if kind == "packs":
                continue
            if kind == "head":
                fc = model.final_conv
                g, dw, db = ops.head1x1_bwd(rec[1], fc[3].weight, dout)
                grads["final_conv.3.weight"], grads["final_conv.3.bias"] = dw, db
            elif kind == "cbr":
                _, conv, bn, tag, c = rec
                below = ri - 1
                while save[below][0] == "dec_tail":
                    below -= 1
                feeds = save[below][4] if save[below][0] == "cbr" else None
                g, presummed = _conv_bn_relu_bwd(conv, bn, tag, c, g, cx, grads, feeds=feeds, presummed=presummed)
            elif kind == "dec_tail":
                pass
            elif kind == "dec_head":
                _, dec, pre, c = rec
                i = int(pre.split(".")[1])
                g, dsk = _decoder_head_bwd(dec, pre, c, g, cx, grads)
                if dsk is not None:
                    dskips[nl - 2 - i] = dsk
            elif kind == "merge":
                _, ds, pre, li, (shape, mg, mm, rr, ln) = rec
                dln = _linear_bwd(ln, g, ds.reduction.weight, cx, grads, pre + "reduction.weight", None)
                dmg, grads[pre + "norm.weight"], grads[pre + "norm.bias"] = ops.layernorm_bwd(
                    dln, mg, ds.norm.weight, mm, rr, defer=cx.sums)
                g = ops.merge_scatter(dmg, shape[1], shape[2])
                if li in dskips:       # the stage output also feeds a decoder skip
                    g = ops.scale_sample(g, None, dskips.pop(li))
            elif kind == "block":
                _, blk, pre, c = rec
                # the block that runs next (the previous one of the same stage) gets its scaled gradient with g
                nxt = save[ri - 1] if ri > 0 else None
                keep_next = nxt[3][13] if (nxt is not None and nxt[0] == "block" and nxt[3][14] == c[14]) else None
                g, g_scaled = _block_bwd(blk, pre, c, g, cx, grads, db2=g_scaled, keep_next=keep_next)
                continue
            elif kind == "embed":
                tok_in, t0, m0, r0 = rec[1]
                E = model.embed_dim_val
                dt0, grads["norm_embed.weight"], grads["norm_embed.bias"] = ops.layernorm_bwd(
                    g, t0, model.norm_embed.weight, m0, r0, defer=cx.sums)
                wpe = model.patch_embed.weight.reshape(E, -1)
                K = wpe.shape[1]
                grads["patch_embed.weight"] = ops.conv_wgrad(tok_in, dt0, K, E, 1).view(model.patch_embed.weight.shape)
                grads["patch_embed.bias"] = _on_side(cx, lambda: ops.channel_sum(dt0, E), dt0)
                g = None
                if ctx.need_dx:
                    dtok = _linear_launch(dt0, cx.packed(wpe, transpose=True))
                    B, Cin, H, W = ctx.in_shape
                    g = ops.unpatchify(dtok, Cin, H, W, model.patch_size_val)
        if cx.sums is not None:
            cx.sums.flush()
            cx.sums = None
        grads.flush_late()
        if cx.side is not None:
            main.wait_stream(cx.side)
            cx.side = None
        cx.held.clear()
        ctx.save = None
        outg = [grads.get(n) if p.requires_grad else None for n, p in _named_params(model)]
        return (g if ctx.need_dx else None, None, None, *outg)
