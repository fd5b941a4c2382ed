"""OptimizedUNet on hand-written gfx950 kernels.

Drop-in for the reference class (training.py:455-531 == enhanced_training_multi_loss.py:516-602):
same constructor, same ``state_dict`` keys and shapes (110 entries), same initialisation, same
``forward(x: f32[B,C,H,W]) -> f32[B,out,H,W]`` and autograd behaviour.  Underneath, the whole
network runs as one ``torch.autograd.Function``: activations are NHWC in the compute dtype
(float32, or bfloat16 with fp32 accumulation), convolutions are MFMA implicit GEMMs with the
BatchNorm statistics taken in their epilogue, and BatchNorm+ReLU+Dropout2d+MaxPool are one pass.
"""
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops

BLOCKS = ("encoder1", "encoder2", "encoder3", "encoder4", "bottleneck",
          "decoder4", "decoder3", "decoder2", "decoder1")
_DROPOUT_P = 0.05


class OptimizedUNet(nn.Module):
    """UNet regressor; ``compute_dtype`` (float32 or bfloat16) selects the kernel precision."""

    def __init__(self, in_channels=4, out_channels=1, init_features=48, compute_dtype=torch.float32):
        super().__init__()
        f = init_features
        self.in_channels, self.out_channels, self.init_features = in_channels, out_channels, f
        self.compute_dtype = compute_dtype
        self.encoder1 = self._block(in_channels, f)
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder2 = self._block(f, f * 2)
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder3 = self._block(f * 2, f * 4)
        self.pool3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder4 = self._block(f * 4, f * 8)
        self.pool4 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.bottleneck = self._block(f * 8, f * 16)
        self.decoder4 = self._block(f * 8 + f * 16, f * 8)
        self.decoder3 = self._block(f * 4 + f * 8, f * 4)
        self.decoder2 = self._block(f * 2 + f * 4, f * 2)
        self.decoder1 = self._block(f + f * 2, f)
        self.conv = nn.Conv2d(f, out_channels, kernel_size=1)
        self._initialize_weights()
        # test hook: a list of nine (B, C) keep masks (0 or 1/(1-p)) used instead of fresh draws
        self.dropout_masks_override: Optional[List[torch.Tensor]] = None
        # data-parallel hook: callable (name, grad) -> grad that receives every parameter gradient as
        # soon as the backward pass has queued its kernel (unet_amd.dist.GradReducer installs itself)
        self.grad_sink = None

    @staticmethod
    def _block(cin, cout):
        # parameter containers only: their own forward is never called
        return nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True), nn.Dropout2d(_DROPOUT_P))

    def _initialize_weights(self):
        # reference training.py:486-494
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("unet_amd.OptimizedUNet runs on the GPU only (hand-written gfx950 "
                               "kernels, no CPU fallback); move the module and its input to 'cuda'")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input (B,{self.in_channels},H,W), got {tuple(x.shape)}")
        params = list(self.parameters())
        if self.training:
            masks = self._draw_masks(x.shape[0], x.device)
            return _UNetTrainFn.apply(x, self, masks, *params)
        with torch.no_grad():
            return _eval_forward(self, x)

    def _draw_masks(self, B, device):
        if self.dropout_masks_override is not None:
            return [None if m is None else m.to(device=device, dtype=torch.float32).contiguous()
                    for m in self.dropout_masks_override]
        # one draw and one scaling per distinct rate for all blocks (each block's (B, c) factors are a contiguous
        # piece of it), not two small launches per block
        plan = [(getattr(self, name)[6].p, getattr(self, name)[3].out_channels) for name in BLOCKS]
        flat = {}
        for p in {p for p, _ in plan if p != 0.0}:
            n = sum(B * c for q, c in plan if q == p)
            flat[p] = [torch.empty(n, dtype=torch.float32, device=device).bernoulli_(1 - p).div_(1 - p), 0]
        masks = []
        for p, c in plan:
            if p == 0.0:
                masks.append(None)
                continue
            buf, off = flat[p]
            masks.append(buf[off:off + B * c].view(B, c))
            flat[p][1] = off + B * c
        return masks


# Weight gradients on a second HIP stream: measured neutral on MI355X once both kernel families ran
# near their own limits (they compete for the same CUs), so it is off unless asked for.
OVERLAP_WGRAD = os.environ.get("UNETK_OVERLAP_WGRAD", "0") != "0"
# the gradient through an encoder's max-pool (+ its skip gradient) formed inside the BatchNorm-backward passes
# instead of stored by a pass of its own (0: the separate pass, for A/B timing; same bits either way)
FUSE_BN_SUMS = os.environ.get("UNETK_FUSE_BN_SUMS", "1") != "0"
# ... for activations up to this many bytes: a larger one is reduced faster by the separate pass, which finds much of
# the gradient just written still in the last-level cache, than by a conv epilogue that has to fetch y again
BN_SUMS_MAX_BYTES = int(os.environ.get("UNETK_BN_SUMS_MAX_BYTES", str(64 << 20)))
FUSE_POOL_BWD = os.environ.get("UNETK_FUSE_POOL_BWD", "1") != "0"
# the last BatchNorm + ReLU + Dropout2d of the bottleneck / a decoder block formed inside the 2x up-sampling that is its
# only reader, instead of stored by a pass of its own (0: the separate pass, for A/B timing; same bits either way)
FUSE_BN_UPSAMPLE = os.environ.get("UNETK_FUSE_BN_UPSAMPLE", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return st


def _pack_all(model, dt, transpose):
    """All 3x3 weights of the network packed in one launch.  transpose: False (forward form), True
    (data-gradient form) or "both" -> (forward dict, data-gradient dict)."""
    keys = [(name, ci) for name in BLOCKS for ci in (0, 3)]
    forms = (False, True) if transpose == "both" else (transpose,)
    packs = ops.pack_many([(getattr(model, name)[ci].weight, tr) for tr in forms for name, ci in keys], dt)
    dicts = [dict(zip(keys, packs[i * len(keys):(i + 1) * len(keys)])) for i in range(len(forms))]
    return tuple(dicts) if transpose == "both" else dicts[0]


def _reads_pair(up_channels: int) -> bool:
    """Whether a decoder block's first conv (and its weight gradient) can read the resized tensor and the skip
    tensor as two operands instead of their concatenation: the first one has to fill whole 32-channel chunks
    of the kernels' K loop."""
    return up_channels > 0 and up_channels % 32 == 0 and os.environ.get("UNETK_CONCAT_COPY", "0") == "0"


class EvalEngine:
    """Inference path with BatchNorm folded into the convolution epilogues and Dropout off.

    Packs the weights and folds the running statistics once; ``__call__`` takes an NHWC batch in
    the compute dtype and returns the (B,out,H,W) fp32 prediction.  Rebuild after changing weights.
    """

    def __init__(self, model):
        self.model = model
        self.dt = model.compute_dtype
        self.pk = _pack_all(model, self.dt, False)
        self.fold = {}
        for name in BLOCKS:
            seq = getattr(model, name)
            for ci, bi in ((0, 1), (3, 4)):
                bn = seq[bi]
                self.fold[(name, ci)] = ops.bn_fold_eval(bn.weight, bn.bias, bn.running_mean,
                                                         bn.running_var, None, bn.eps)
        self.head_w = model.conv.weight.detach()
        self.head_b = model.conv.bias.detach() if model.conv.bias is not None else None

    def _block(self, name, t, out=None, pool=False, head=None):
        """Two conv+BN+ReLU units; the second can also hand back the max-pooled output, or only
        the network's final 1x1 conv of it (both computed in its epilogue).  ``t`` may be the pair
        (resized, skip) that the reference concatenates: it is then read in place."""
        sc, sh = self.fold[(name, 0)]
        if isinstance(t, tuple):
            t, _ = ops.conv_fwd_cat(t[0], t[1], self.pk[(name, 0)], scale=sc, shift=sh, act=1)
        else:
            t, _ = ops.conv_fwd(t, self.pk[(name, 0)], scale=sc, shift=sh, act=1)
        sc, sh = self.fold[(name, 3)]
        if pool or head is not None:
            return ops.conv_fwd(t, self.pk[(name, 3)], scale=sc, shift=sh, act=1, out=out, pool=pool, head=head,
                                write_out=head is None)
        return ops.conv_fwd(t, self.pk[(name, 3)], scale=sc, shift=sh, act=1, out=out)[0]

    def _encoder(self, name, dec, t):
        """An encoder block's output is the skip input of its decoder: kept as a dense tensor that the
        decoder's first conv reads next to the resized one (no concatenated copy), or, when the channel
        counts do not allow that, written straight into the skip half of a concat buffer."""
        seq, dseq = getattr(self.model, name), getattr(self.model, dec)
        C2 = seq[3].out_channels
        C1 = dseq[0].in_channels - C2
        B, H, W, _ = t.shape
        if _reads_pair(C1):
            skip, pooled, _ = self._block(name, t, pool=True)
            return skip, pooled
        cat = torch.empty((B, H, W, C1 + C2), dtype=t.dtype, device=t.device)
        _, pooled, _ = self._block(name, t, out=cat[..., C1:], pool=True)
        return cat, pooled

    def __call__(self, h):
        s1, p = self._encoder("encoder1", "decoder1", h)
        s2, p = self._encoder("encoder2", "decoder2", p)
        s3, p = self._encoder("encoder3", "decoder3", p)
        s4, p = self._encoder("encoder4", "decoder4", p)
        d = self._block("bottleneck", p)

        def joined(name, enc, d, s):
            C1 = getattr(self.model, name)[0].in_channels - getattr(self.model, enc)[3].out_channels
            if _reads_pair(C1):
                return (ops.upsample(d, s.shape[1], s.shape[2]), s)
            return ops.upsample_into(d, s)

        for name, enc, s in (("decoder4", "encoder4", s4), ("decoder3", "encoder3", s3), ("decoder2", "encoder2", s2)):
            d = self._block(name, joined(name, enc, d, s))
        last = self.pk[("decoder1", 3)]
        if last.Co <= last.bn:           # the head needs every channel in one workgroup
            return self._block("decoder1", joined("decoder1", "encoder1", d, s1), head=(self.head_w, self.head_b))[2]
        d = self._block("decoder1", joined("decoder1", "encoder1", d, s1))
        return ops.head1x1_fwd(d, self.head_w, self.head_b)


def _eval_forward(model, x):
    return EvalEngine(model)(ops.nchw_to_nhwc(x, model.compute_dtype))


class _UNetTrainFn(torch.autograd.Function):
    """Whole-network training forward/backward (BatchNorm uses batch statistics)."""

    @staticmethod
    def forward(ctx, x, model, masks, *params):
        dt = model.compute_dtype
        # forward and data-gradient forms of all 18 weights: one launch
        pk, ctx.pkT = _pack_all(model, dt, "both")
        saved = {}
        tracked = []                 # the BatchNorm step counters: advanced by one launch at the end
        B = x.shape[0]
        h = ops.nchw_to_nhwc(x.detach(), dt)
        saved["x"] = h

        def block(name, t, keep, pool, cat_c1=0, up_to=None):
            """conv-BN-ReLU twice (+Dropout2d mask, + 2x2 max-pool for encoder blocks).  An encoder
            block writes its output straight into the skip half of its decoder's concat buffer
            (channels [cat_c1:]) and keeps the pooling argmax map for the backward.  up_to: the (H, W) the
            block's output is resized to by its only reader; where the fused pass applies the block returns
            ("deferred", y, scale, shift, keep) instead of the activation."""
            seq = getattr(model, name)
            units = []
            pooled = argmax = cat = None
            for u, (ci, bi) in enumerate(((0, 1), (3, 4))):
                bn = seq[bi]
                if isinstance(t, tuple):       # (resized, skip): the concatenation the reference stores
                    y, stats = ops.conv_fwd_cat(t[0], t[1], pk[(name, ci)], want_stats=True)
                else:
                    y, stats = ops.conv_fwd(t, pk[(name, ci)], want_stats=True)
                Bn, H, W, C = y.shape
                mean, invstd, sc, sh = ops.bn_finalize(
                    stats, Bn * H * W, bn.weight, bn.bias,
                    bn.running_mean if bn.track_running_stats else None,
                    bn.running_var if bn.track_running_stats else None, bn.eps,
                    bn.momentum if bn.momentum is not None else 0.1)
                if bn.track_running_stats:
                    tracked.append(bn.num_batches_tracked)
                last = u == 1
                if last and pool and _reads_pair(cat_c1):
                    a, pooled, argmax = ops.bn_act_fwd(y, sc, sh, keep, relu=True, pool=True, want_argmax=True)
                    cat = a                    # the skip tensor itself: its decoder reads it in place
                elif last and pool:
                    cat = torch.empty((Bn, H, W, cat_c1 + C), dtype=y.dtype, device=y.device)
                    a, pooled, argmax = ops.bn_act_fwd(y, sc, sh, keep, relu=True, pool=True,
                                                       out=cat[..., cat_c1:], want_argmax=True)
                elif (last and FUSE_BN_UPSAMPLE and up_to is not None and tuple(up_to) == (2 * H, 2 * W)
                      and ops.bn_act_upsample_supported(y, sc, sh, keep)):
                    a = ("deferred", y, sc, sh, keep)
                else:
                    a, _ = ops.bn_act_fwd(y, sc, sh, keep if last else None, relu=True)
                units.append((t, y, mean, invstd, sc, sh))
                t = a
            saved[name] = (units, argmax, keep)
            return t, pooled, cat

        def up_channels(dec, enc):      # channels of the resized (non-skip) input of a decoder block
            return getattr(model, dec)[0].in_channels - getattr(model, enc)[3].out_channels

        _, p1, cat1 = block("encoder1", h, masks[0], True, up_channels("decoder1", "encoder1"))
        _, p2, cat2 = block("encoder2", p1, masks[1], True, up_channels("decoder2", "encoder2"))
        _, p3, cat3 = block("encoder3", p2, masks[2], True, up_channels("decoder3", "encoder3"))
        _, p4, cat4 = block("encoder4", p3, masks[3], True, up_channels("decoder4", "encoder4"))
        cats = (("decoder4", cat4), ("decoder3", cat3), ("decoder2", cat2), ("decoder1", cat1))
        d, _, _ = block("bottleneck", p4, masks[4], False, up_to=cat4.shape[1:3])
        for i, (name, cat) in enumerate(cats):
            if isinstance(d, tuple):        # BatchNorm + ReLU + Dropout2d ride on the resize
                _, y_low, sc_low, sh_low, keep_low = d
                saved[name + ".low"] = (y_low.shape[1:3], y_low.shape[3])
                if _reads_pair(y_low.shape[3]):
                    joined = (ops.bn_act_upsample(y_low, sc_low, sh_low, keep_low), cat)
                else:
                    joined = ops.bn_act_upsample(y_low, sc_low, sh_low, keep_low, cat=cat)
            else:
                saved[name + ".low"] = (d.shape[1:3], d.shape[3])
                if _reads_pair(d.shape[3]):
                    joined = (ops.upsample(d, cat.shape[1], cat.shape[2]), cat)
                else:
                    joined = ops.upsample_into(d, cat)
            d, _, _ = block(name, joined, masks[5 + i], False,
                            up_to=cats[i + 1][1].shape[1:3] if i + 1 < len(cats) else None)
        saved["head_in"] = d
        out = ops.head1x1_fwd(d, model.conv.weight, model.conv.bias)
        if tracked:
            torch._foreach_add_(tracked, 1)
        ctx.model = model
        ctx.saved = saved
        ctx.versions = [p._version for p in params]
        ctx.need_dx = ctx.needs_input_grad[0]
        ctx.in_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        model, saved = ctx.model, ctx.saved
        if saved is None:
            raise RuntimeError("the UNet step was already back-propagated: its saved activations are freed by "
                               "the first backward (run the forward again)")
        if ctx.versions != [p._version for p in model.parameters()]:
            raise RuntimeError("a UNet parameter was modified in place between the forward pass and its backward "
                               "pass (the packed copies made by the forward no longer match)")
        dt = model.compute_dtype
        pkT = ctx.pkT
        ctx.pkT = None
        grads = ops.GradOut(model.grad_sink)
        # Weight gradients are off the critical chain (dgrad -> BatchNorm backward -> dgrad ...):
        # optionally they run on a second HIP stream, overlapping the HBM-bound BatchNorm / pool /
        # resize backward passes of the following layers.
        main = torch.cuda.current_stream()
        side = _side_stream(dout.device) if (OVERLAP_WGRAD and model.grad_sink is None) else main
        held = []          # operands of side-stream kernels stay allocated until the join

        # final 1x1 conv.  With one output channel the gradient of its input is the outer product
        # dout[pixel] * w[channel]: it is handed to the BatchNorm backward of the last block as its two
        # factors and never written (a (B,H,W,C) tensor written once and read twice otherwise).
        if model.conv.out_channels == 1:
            _, dw, db = ops.head1x1_bwd(saved["head_in"], model.conv.weight, dout, need_dx=False)
            g = (dout.detach().contiguous().float().view(dout.shape[0], dout.shape[2], dout.shape[3]),
                 model.conv.weight.detach().reshape(-1).float().contiguous())
        else:
            g, dw, db = ops.head1x1_bwd(saved["head_in"], model.conv.weight, dout)
        grads["conv.weight"], grads["conv.bias"] = dw, db

        def block_bwd(name, g, need_dx=True, split=0):
            seq = getattr(model, name)
            units, _, keep = saved[name]
            presummed = None
            for u in (1, 0):
                ci, bi = ((0, 1), (3, 4))[u]
                t_in, y, mean, invstd, sc, sh = units[u]
                dy, dgamma, dbeta = ops.bn_act_bwd(g, y, sc, sh, mean, invstd,
                                                   keep if u == 1 else None, relu=True, presummed=presummed)
                presummed = None
                grads[f"{name}.{bi}.weight"], grads[f"{name}.{bi}.bias"] = dgamma, dbeta
                conv = seq[ci]
                def wgrad():
                    return ops.conv_wgrad(t_in, dy, conv.in_channels, conv.out_channels, 3)

                def dgrad():
                    nonlocal presummed
                    if not (u == 1 or need_dx):
                        return None
                    pw = pkT[(name, ci)]
                    if (u == 1 and FUSE_BN_SUMS and units[0][1].numel() * units[0][1].element_size() <= BN_SUMS_MAX_BYTES
                            and ops.conv_bwd_data_bnsums_supported(pw, dt)):
                        # the gradient this conv produces goes straight into the first unit's BatchNorm backward:
                        # the conv leaves that pass's two sums beside it
                        _, y0, mean0, invstd0, sc0, sh0 = units[0]
                        gg, presummed = ops.conv_bwd_data_bnsums(dy, pw, y0, sc0, sh0, mean0, invstd0)
                        return gg
                    # the first conv of a decoder block read cat([up, skip]): its data gradient comes
                    # out as two dense tensors (when the boundary falls on a channel block)
                    sp = split if (u == 0 and split and split % pw.bn == 0) else 0
                    gg, _ = ops.conv_fwd(dy, pw, split=sp)
                    if u == 0 and split and not sp:
                        gg = (gg[..., :split], gg[..., split:])
                    return gg

                if side is main:
                    dw = wgrad()
                    g = dgrad()
                else:
                    # the side stream may start this weight gradient once the data-gradient conv of
                    # the same layer is through: it then shares the chip with the HBM-bound
                    # BatchNorm backward of the next layer rather than with another MFMA kernel
                    g = dgrad()
                    ready = torch.cuda.Event()
                    ready.record(main)
                    held.append(dy)
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        dw = wgrad()
                    dw.record_stream(main)
                grads[f"{name}.{ci}.weight"] = dw
            return g

        enc_names = ("encoder1", "encoder2", "encoder3", "encoder4")
        dskips = {}
        for name, enc in (("decoder1", "encoder1"), ("decoder2", "encoder2"), ("decoder3", "encoder3"),
                          ("decoder4", "encoder4")):
            (lh, lw), C1 = saved[name + ".low"]
            g_up, dskips[enc] = block_bwd(name, g, split=C1)
            g = ops.upsample_bwd(g_up, C1, lh, lw)
        g = block_bwd("bottleneck", g)
        for enc in reversed(enc_names):
            if FUSE_POOL_BWD and dskips[enc].shape[1] >= 2 and dskips[enc].shape[2] >= 2:
                g = ops.PooledGrad(g, dskips[enc], saved[enc][1])      # formed inside the BatchNorm-backward passes
            else:
                g = ops.pool_bwd(None, g, dskips[enc], argmax=saved[enc][1])
            g = block_bwd(enc, g, need_dx=(enc != "encoder1") or ctx.need_dx)
        dx = None
        if ctx.need_dx:
            dx = ops.nhwc_to_nchw(g, ctx.in_shape[1])
        if side is not main:
            main.wait_stream(side)
        held.clear()
        ctx.saved = None
        out_grads = []
        for n, p in model.named_parameters():
            out_grads.append(grads[n] if p.requires_grad else None)
        return (dx, None, None, *out_grads)
