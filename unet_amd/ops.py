"""Tensor-level wrappers over the libunetk C ABI.

Activations are NHWC tensors of the compute dtype (float32 or bfloat16) living on the
GPU; torch only provides their memory and the current stream.  Every function
enqueues kernels on ``torch.cuda.current_stream()`` and returns without synchronising.
"""
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream, dtype_code

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class GradOut(dict):
    """Parameter gradients by name, as a whole-network backward produces them.  With a sink (the
    data-parallel reducer's ``grad_sink(name, tensor) -> tensor``) every gradient is handed over the
    moment its kernel is queued, so its bucket's all-reduce can start while the backward continues."""

    def __init__(self, sink=None, late=()):
        """late: names whose tensors are complete only once the backward's closing launches have run (sums taken at
        its end): they are handed to the sink by flush_late(), not when they are assigned."""
        super().__init__()
        self.sink = sink
        self.late = frozenset(late) if sink is not None else frozenset()
        self._held = []

    def __setitem__(self, name, grad):
        if self.sink is not None and grad is not None:
            if name in self.late:
                self._held.append(name)
            else:
                # a gradient computed on another stream (see on_stream) is handed over on that stream: the sink's
                # copy into its bucket is then ordered behind the kernel that produces the gradient
                st = getattr(grad, "_unetk_stream", None)
                if st is not None:
                    with torch.cuda.stream(st):
                        grad = self.sink(name, grad)
                else:
                    grad = self.sink(name, grad)
        dict.__setitem__(self, name, grad)

    def flush_late(self):
        for name in self._held:
            dict.__setitem__(self, name, self.sink(name, dict.__getitem__(self, name)))
        self._held = []


def on_stream(t, st):
    """Marks tensor t as produced on HIP stream st (read by GradOut); returns t."""
    if t is not None and st is not None:
        t._unetk_stream = st
    return t


def view_as_param(t, shape):
    """t.view(shape) that keeps the stream mark."""
    v = t.view(shape)
    st = getattr(t, "_unetk_stream", None)
    if st is not None:
        v._unetk_stream = st
    return v


def _vec(dt):
    return 8 if dt == torch.bfloat16 else 4


def pad_channels(c: int, dt) -> int:
    v = _vec(dt)
    return (c + v - 1) // v * v


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"unet_amd: {what} must live on the GPU; this package drives hand-written gfx950 "
            "kernels and has no CPU fallback")


def nchw_to_nhwc(x: torch.Tensor, dt, ld: Optional[int] = None) -> torch.Tensor:
    """fp32 (B,C,H,W) -> compute-dtype (B,H,W,ld), zero-filled channel padding."""
    _require_cuda(x, "input")
    x = x.contiguous().float()
    B, C, H, W = x.shape
    ld = ld or pad_channels(C, dt)
    out = torch.empty((B, H, W, ld), dtype=dt, device=x.device)
    call("unetk_nchw_to_nhwc", ptr(x), ptr(out), B, C, H, W, ld, dtype_code(dt), stream())
    return out


def nhwc_to_nchw(x: torch.Tensor, C: int) -> torch.Tensor:
    B, H, W, ld = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    call("unetk_nhwc_to_nchw", ptr(x), ld, ptr(out), B, C, H, W, dtype_code(x.dtype), stream())
    return out


class PackedConv:
    """A convolution weight re-packed for the implicit-GEMM kernel."""
    __slots__ = ("data", "Co", "Ci", "ks", "bn", "dt")

    def __init__(self, w: torch.Tensor, dt, transpose: bool = False, ci_pad: Optional[int] = None):
        """w: fp32 (Co, Ci, k, k) or (Co, Ci) on the GPU.  transpose packs the data-gradient conv."""
        _require_cuda(w, "weight")
        if w.dim() == 2:
            w = w.view(w.shape[0], w.shape[1], 1, 1)
        w = w.detach().contiguous().float()
        Co, Ci, ks, _ = w.shape
        lib = _lib.load()
        pCo, pCi = (Ci, Co) if transpose else (Co, Ci)
        self.bn = lib.unetk_conv_pick_bn(pCo)
        self.Co, self.Ci, self.ks, self.dt = pCo, pCi, ks, dt
        n = lib.unetk_conv_packed_elems(pCo, pCi, ks, self.bn, dtype_code(dt))
        self.data = torch.empty(n, dtype=dt, device=w.device)
        call("unetk_conv_pack_weight", ptr(w), ptr(self.data), Co, Ci, ks, self.bn, dtype_code(dt),
             1 if transpose else 0, stream())

    @classmethod
    def _view(cls, meta, flat, dt):
        """A pack that lives inside a buffer shared by a batch of packs (see pack_many)."""
        pCo, pCi, ks, bn, o, n = meta
        pk = cls.__new__(cls)
        pk.bn, pk.Co, pk.Ci, pk.ks, pk.dt = bn, pCo, pCi, ks, dt
        pk.data = flat[o:o + n]
        return pk


_PACK_TABLES = {}      # (pointers, shapes) -> (jobs, blocks, nblocks, pinned host copies); a handful of entries


def pack_many(weights, dt):
    """Pack several weights with ONE launch.  weights: list of (fp32 GPU tensor (Co, Ci[, k, k]), transpose);
    returns the list of PackedConv (views of one buffer).  transpose may also be "convt" / "convt_t": the tensor is
    a ConvTranspose2d(k=2, s=2) weight (Cin, up, 2, 2), packed as the (4*up, Cin) matrix whose row
    (ky*2+kx)*up + co is weight[:, co, ky, kx] (forward form) or as its data-gradient form; or "bn64": the forward
    form with 64-channel blocks whatever the width (the fused MLP kernel's fc1 operand), "bn64_t": the data-gradient
    form with 64-channel blocks.  The device-side job
    table depends on the source addresses and shapes only, and parameters do not move: a training step re-uses it
    (a new one goes up from pinned memory without blocking the host)."""
    if not weights:
        return []
    import numpy as np
    # fast path: the same parameters as last time (addresses, shapes, forms)
    light = (tuple((w.data_ptr(), w.shape, tr, w.dtype, w.is_contiguous()) for w, tr in weights), dt)
    plan = _PACK_TABLES.get(light)
    if plan is not None:
        metas, off, hit = plan
        flat = torch.empty(off, dtype=dt, device=weights[0][0].device)
        call("unetk_conv_pack_weights", ptr(hit[0]), ptr(hit[1]), hit[2], ptr(flat), dtype_code(dt), stream())
        return [PackedConv._view(m, flat, dt) for m in metas]
    lib = _lib.load()
    kc = 32 if dt == torch.bfloat16 else 16
    metas, srcs, rows, off = [], [], [], 0
    for w, transpose in weights:
        _require_cuda(w, "weight")
        force_bn = 0
        if transpose in ("bn64", "bn64_t"):
            force_bn, transpose = 64, transpose == "bn64_t"
        convt = isinstance(transpose, str)
        if convt:
            if transpose not in ("convt", "convt_t") or w.dim() != 4 or tuple(w.shape[2:]) != (2, 2):
                raise ValueError("pack_many: the ConvTranspose form takes a (Cin, up, 2, 2) weight")
            transpose = transpose == "convt_t"
        elif w.dim() == 2:
            w = w.view(w.shape[0], w.shape[1], 1, 1)
        w = w.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.contiguous().float()
        if convt:
            Co, Ci, ks = 4 * w.shape[1], w.shape[0], 1
        else:
            Co, Ci, ks, _ = w.shape
        pCo, pCi = (Ci, Co) if transpose else (Co, Ci)
        bn = force_bn or lib.unetk_conv_pick_bn(pCo)
        if pCo % bn and force_bn:
            raise ValueError(f"pack_many: {pCo} output channels are not whole blocks of {bn}")
        n = lib.unetk_conv_packed_elems(pCo, pCi, ks, bn, dtype_code(dt))
        metas.append((pCo, pCi, ks, bn, off, n))
        srcs.append(w)
        rows.append((w.data_ptr(), off, Co, Ci, ks, bn, -(-pCi // kc), (1 if transpose else 0) + (2 if convt else 0), n))
        off += -(-n // 8) * 8                     # keep every pack 16-byte aligned
    dev = srcs[0].device
    key = (tuple(rows), dt, dev)
    hit = _PACK_TABLES.get(key)
    if hit is None:
        jobs_h = torch.from_numpy(np.asarray(rows, dtype=np.int64).reshape(-1)).pin_memory()
        nblk = np.asarray([lib.unetk_conv_pack_groups(m[0], m[1], m[2], m[3], dtype_code(dt)) for m in metas], dtype=np.int64)
        job_of = np.repeat(np.arange(len(rows), dtype=np.int32), nblk)
        first = np.repeat(np.cumsum(nblk) - nblk, nblk)
        blocks = np.stack([job_of, (np.arange(int(nblk.sum())) - first).astype(np.int32)], 1).astype(np.int32)
        blocks_h = torch.from_numpy(np.ascontiguousarray(blocks).reshape(-1)).pin_memory()
        hit = (jobs_h.to(dev, non_blocking=True), blocks_h.to(dev, non_blocking=True), int(nblk.sum()), jobs_h, blocks_h)
        if len(_PACK_TABLES) >= 16:
            _PACK_TABLES.pop(next(iter(_PACK_TABLES)))
        _PACK_TABLES[key] = hit
    if all(w.dtype == torch.float32 and w.is_contiguous() for w, _ in weights):    # no temporaries among the sources
        _PACK_TABLES[light] = (metas, off, hit)
    flat = torch.empty(off, dtype=dt, device=dev)
    call("unetk_conv_pack_weights", ptr(hit[0]), ptr(hit[1]), hit[2], ptr(flat), dtype_code(dt), stream())
    return [PackedConv._view(m, flat, dt) for m in metas]


def conv_fwd_cat(x1: torch.Tensor, x2: torch.Tensor, pw: PackedConv, want_stats: bool = False, scale=None, shift=None,
                 act: int = 0):
    """3x3 conv of the channel concatenation [x1 | x2] (both (B,H,W,C) NHWC, read in place: the concatenated
    tensor is never stored).  x1 must hold a multiple of the packing chunk (32 bf16 / 16 f32 channels)."""
    B, H, W, C1 = x1.shape
    C2 = x2.shape[-1]
    dt = x1.dtype
    Ci, Co = C1 + C2, pad_channels(pw.Co, dt)
    if x2.shape[:3] != x1.shape[:3] or x2.dtype != dt or pw.ks != 3 or pad_channels(pw.Ci, dt) != Ci:
        raise ValueError("conv_fwd_cat: the two inputs and the packed weight do not fit together")
    out = torch.empty((B, H, W, Co), dtype=dt, device=x1.device)
    stats = None
    if want_stats:
        nt = _lib.load().unetk_conv_num_tiles(B, H, W, 3)
        stats = torch.empty((nt, 2, Co), dtype=torch.float32, device=x1.device)
    call("unetk_conv_fwd_cat", ptr(x1), x1.stride(2), C1, ptr(x2), x2.stride(2), ptr(pw.data), ptr(out), Co, B, H, W,
         Ci, Co, pw.bn, dtype_code(dt), ptr(stats), ptr(scale), ptr(shift), act, stream())
    return out, stats


def conv_fwd(x: torch.Tensor, pw: PackedConv, Ci: Optional[int] = None, Co: Optional[int] = None,
             out: Optional[torch.Tensor] = None, want_stats: bool = False, scale=None, shift=None,
             act: int = 0, resid=None, resid_keep=None, split: int = 0, pool: bool = False, head=None,
             write_out: bool = True):
    """x: (B,H,W,ld) NHWC.  Channel counts default to the packed weight's, rounded up to the
    vector width for the output (padding channels come out as exact zeros).  With ``split`` (a
    multiple of the packed channel block) the result is two tensors, channels [0,split) and
    [split,Co), and the return value is ((out_a, out_b), stats).  Inference-time fusions (3x3 only):
    ``pool`` also returns the 2x2 max-pool of the output, ``head=(w, bias)`` the final 1x1 conv
    (NCHW fp32) applied to it; with either the return value is (out, pooled, head_out) and
    ``write_out=False`` skips the output itself."""
    B, H, W, _ = x.shape
    ldin = x.stride(2)
    dt = x.dtype
    v = _vec(dt)
    Ci = Ci if Ci is not None else min(x.shape[-1], pad_channels(pw.Ci, dt))
    if Co is None:
        Co = out.shape[-1] if out is not None else pad_channels(pw.Co, dt)
    if split:
        out = torch.empty((B, H, W, split), dtype=dt, device=x.device)
        out2 = torch.empty((B, H, W, Co - split), dtype=dt, device=x.device)
        call("unetk_conv_fwd_split", ptr(x), ldin, ptr(pw.data), ptr(out), split, ptr(out2), Co - split, split,
             B, H, W, Ci, Co, pw.ks, pw.bn, dtype_code(dt), None, ptr(scale), ptr(shift), act, None, 0, None,
             stream())
        return (out, out2), None
    if pool or head is not None:
        if out is None and write_out:
            out = torch.empty((B, H, W, Co), dtype=dt, device=x.device)
        pooled = torch.empty((B, H // 2, W // 2, Co), dtype=dt, device=x.device) if pool else None
        hw = hb = hout = None
        hco = 0
        if head is not None:
            hw, hb = head
            hco = hw.shape[0]
            hw = hw.detach().reshape(hco, -1).float().contiguous()
            hout = torch.empty((B, hco, H, W), dtype=torch.float32, device=x.device)
        call("unetk_conv_fwd_fused", ptr(x), ldin, ptr(pw.data), ptr(out), out.stride(2) if out is not None else 0,
             None, 0, 0, ptr(pooled), ptr(hw), ptr(hb), ptr(hout), hco, B, H, W, Ci, Co, pw.ks, pw.bn,
             dtype_code(dt), None, ptr(scale), ptr(shift), act, None, 0, None, stream())
        return out, pooled, hout
    if out is None:
        out = torch.empty((B, H, W, Co), dtype=dt, device=x.device)
    ldout = out.stride(2)          # `out` may be a channel slice of a wider (concat) buffer
    stats = None
    if want_stats:
        if resid is not None:
            raise ValueError("conv_fwd: batch statistics and a residual are exclusive")
        nt = _lib.load().unetk_conv_stats_rows(B, H, W, Ci, Co, pw.ks, pw.bn, dtype_code(dt), ldin, act)
        stats = torch.empty((nt, 2, Co), dtype=torch.float32, device=x.device)
    call("unetk_conv_fwd", ptr(x), ldin, ptr(pw.data), ptr(out), ldout, B, H, W, Ci, Co, pw.ks, pw.bn,
         dtype_code(dt), ptr(stats), ptr(scale), ptr(shift), act, ptr(resid),
         resid.stride(2) if resid is not None else 0, ptr(resid_keep), stream())
    return out, stats


def conv_bwd_data_bnsums_supported(pw: PackedConv, dt) -> bool:
    return pw.ks == 3 and bool(_lib.load().unetk_conv_bwd_data_bnsums_supported(pad_channels(pw.Co, dt), pw.bn,
                                                                                dtype_code(dt)))


def conv_bwd_data_bnsums(dy: torch.Tensor, pwT: PackedConv, y, scale, shift, mean, invstd):
    """g = data gradient of a 3x3 conv (``pwT``: its weights packed transposed), the gradient of relu(y*scale+shift);
    returns (g, partials): the per-tile sums that ``bn_act_bwd(..., presummed=partials)`` takes instead of reducing
    g and y itself."""
    B, H, W, _ = dy.shape
    dt = dy.dtype
    Ci, Co = min(dy.shape[-1], pad_channels(pwT.Ci, dt)), pad_channels(pwT.Co, dt)
    if y.shape != (B, H, W, Co) or y.dtype != dt or y.stride(3) != 1 or y.stride(1) != W * y.stride(2):
        raise ValueError("conv_bwd_data_bnsums: y does not match the gradient being produced")
    out = torch.empty((B, H, W, Co), dtype=dt, device=dy.device)
    nt = _lib.load().unetk_conv_num_tiles(B, H, W, 3)
    partials = torch.empty((nt, 2, Co), dtype=torch.float32, device=dy.device)
    call("unetk_conv_bwd_data_bnsums", ptr(dy), dy.stride(2), ptr(pwT.data), ptr(out), B, H, W, Ci, Co, pwT.bn,
         dtype_code(dt), ptr(y), y.stride(2), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(partials), stream())
    return out, partials


def linear_supported(x: torch.Tensor, pw: PackedConv) -> bool:
    """Whether the token-linear GEMM kernel takes this product (bfloat16, K a multiple of 32, N a multiple of the
    packed channel block); everything else goes through the 1x1 form of the convolution kernel."""
    if pw.ks != 1 or x.dtype != torch.bfloat16 or x.shape[-1] != pw.Ci or x.stride(-1) != 1:
        return False
    M = x.numel() // x.shape[-1]
    return bool(_lib.load().unetk_linear_supported(M, pw.Ci, pw.Co, pw.bn, dtype_code(x.dtype)))


def linear_fwd(x, pw: PackedConv, bias=None, act=0, resid=None, resid_keep=None, out=None, out2=False,
               gelu_grad_of=None):
    """out[..., n] = epilogue(sum_k x[..., k] * w[n][k]) for a channels-last token tensor x (..., K) whose token rows
    are equally spaced (a dense tensor or a channel slice of one).  ``out`` may be a channel slice of a wider dense
    buffer.  ``out2=True`` (with GELU): returns (pre-activation, GELU(pre-activation)).  ``gelu_grad_of``: the
    result is multiplied by GELU'(that tensor).  ``resid``/``resid_keep``: resid + keep[sample] * value."""
    K, N = pw.Ci, pw.Co
    M = x.numel() // x.shape[-1]
    ldx = x.stride(-2)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    second = torch.empty_like(out) if out2 else None
    tps = 1
    if resid_keep is not None:
        tps = M // x.shape[0]
    call("unetk_linear_fwd", ptr(x), ldx, ptr(pw.data), ptr(out), out.stride(-2), ptr(second),
         second.stride(-2) if second is not None else 0, M, K, N, pw.bn, dtype_code(x.dtype), ptr(bias), act,
         ptr(resid), resid.stride(-2) if resid is not None else 0, ptr(resid_keep), tps, ptr(gelu_grad_of),
         gelu_grad_of.stride(-2) if gelu_grad_of is not None else 0, stream())
    return (out, second) if out2 else out


def convT2x2_bwd_data(dout, up, pwT, h, w):
    """Data gradient of ConvTranspose2d(k=2, s=2): dout (B, 2h, 2w, >= up channels, the first ``up`` are read) ->
    dx (B, h, w, Cin); pwT: the data-gradient pack of the (4*up, Cin) matrix.  None when the GEMM path does not take
    the shape (the caller un-shuffles into a token matrix and runs the plain product)."""
    B = dout.shape[0]
    Cin = pwT.Co
    if not (dout.dtype == torch.bfloat16 and up % 32 == 0 and pwT.Ci == 4 * up and dout.stride(3) == 1 and
            dout.stride(1) == dout.shape[2] * dout.stride(2) and dout.stride(0) == dout.shape[1] * dout.stride(1) and
            _lib.load().unetk_linear_supported(B * h * w, 4 * up, Cin, pwT.bn, dtype_code(dout.dtype))):
        return None
    dx = torch.empty((B, h, w, Cin), dtype=dout.dtype, device=dout.device)
    call("unetk_convt2x2_bwd_data", ptr(dout), dout.stride(2), ptr(pwT.data), ptr(dx), Cin, B, h, w, up, Cin, pwT.bn,
         dtype_code(dout.dtype), stream())
    return dx


def convT2x2_fwd(x, pw: PackedConv, bias, out):
    """ConvTranspose2d(k=2, s=2) + bias of channels-last x (B,h,w,K) into channels [0, up) of out (B,2h,2w,ld): one
    GEMM launch whose epilogue stores every value at its shuffled pixel.  pw: the pack of the (4*up, K) matrix with
    rows ordered (ky, kx, co).  The caller checks linear_supported(x, pw)."""
    B, h, w, K = x.shape
    up = pw.Co // 4
    call("unetk_convt2x2_fwd", ptr(x), x.stride(-2), ptr(pw.data), ptr(out), out.stride(-2), B, h, w, K, up, pw.bn,
         dtype_code(x.dtype), ptr(bias), stream())
    return out


def token_wgrad(x: torch.Tensor, dy: torch.Tensor, K: int, N: int, want_bias: bool = True):
    """(dw (N, K), db (N,) or None) fp32 of a linear layer over tokens, bfloat16 operands: x (..., ldx) and
    dy (..., lddy) channels-last with equally spaced token rows."""
    M = x.numel() // x.shape[-1]
    lib = _lib.load()
    need = lib.unetk_token_wgrad_workspace_bytes(M, K, N)
    workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if want_bias else None
    call("unetk_token_wgrad", ptr(x), x.stride(-2), ptr(dy), dy.stride(-2), ptr(dw), ptr(db), M, K, N,
         dtype_code(x.dtype), ptr(workspace), need, 0, stream())
    return dw, db


def token_wgrad_supported(x: torch.Tensor, dy: torch.Tensor, K: int, N: int) -> bool:
    if x.dtype != torch.bfloat16 or x.stride(-1) != 1 or dy.stride(-1) != 1:
        return False
    M = x.numel() // x.shape[-1]
    return bool(_lib.load().unetk_token_wgrad_supported(M, K, N, dtype_code(x.dtype)))


# dw tiles of the token weight-gradient GEMM are 128 x 128: from this many padded elements of dw on it beats
# the sliced 1x1 form of the convolution weight-gradient kernel (measured on the Swin shapes at batch 64)
_TOKEN_WGRAD_MIN_ELEMS = 196608


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor, Ci: int, Co: int, want_bias: bool = True):
    """(dw (Co,Ci), db (Co,) or None) fp32 of a linear layer over tokens: x (..., ldx) and dy (..., lddy)
    channels-last.  bfloat16: the token weight-gradient GEMM for large dw, else the 1x1 form of the convolution
    weight gradient (whose bias sums ride on the same pass); float32: conv_wgrad + channel_sum."""
    dt = x.dtype
    if token_wgrad_supported(x, dy, Ci, Co):
        pad = (-(-Co // 128) * 128) * (-(-Ci // 128) * 128)
        if pad >= _TOKEN_WGRAD_MIN_ELEMS:
            return token_wgrad(x, dy, Ci, Co, want_bias)
    if dt != torch.bfloat16 or pad_channels(Ci, dt) != Ci or pad_channels(Co, dt) != Co or not want_bias:
        return conv_wgrad(x, dy, Ci, Co, 1).view(Co, Ci), (channel_sum(dy, Co) if want_bias else None)
    ntok = x.numel() // x.shape[-1]
    need = _lib.load().unetk_linear_wgrad_workspace_bytes(ntok, Ci, Co, dtype_code(dt))
    workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    dw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    db = torch.empty((Co,), dtype=torch.float32, device=x.device)
    call("unetk_linear_wgrad", ptr(x), x.stride(-2), ptr(dy), dy.stride(-2), ptr(dw), ptr(db), ntok, Ci, Co,
         dtype_code(dt), ptr(workspace), need, 0, stream())
    return dw, db


def conv_wgrad(x, dy: torch.Tensor, Ci: int, Co: int, ks: int,
               dw: Optional[torch.Tensor] = None, accumulate: bool = False,
               workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw (Co,Ci,k,k) fp32 from NHWC activations x (B,H,W,ldx) and output gradient dy (B,H,W,lddy).  x may be a
    pair (x1, x2): the layer's input was their channel concatenation (3x3 only, see conv_fwd_cat)."""
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    B, H, W, _ = x.shape
    ldx = x.stride(2)
    dt = x.dtype
    v = _vec(dt)
    Cip, Cop = pad_channels(Ci, dt), pad_channels(Co, dt)
    need = _lib.load().unetk_conv_wgrad_workspace_bytes(B, H, W, Cip, Cop, ks, dtype_code(dt))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    padded = (Cip != Ci) or (Cop != Co)
    tgt = dw
    if dw is None or padded:
        tgt = torch.empty((Cop, Cip, ks, ks), dtype=torch.float32, device=x.device)
    acc_flag = 1 if (accumulate and not padded and dw is not None) else 0
    if x2 is not None:
        call("unetk_conv_wgrad_cat", ptr(x), ldx, x.shape[-1], ptr(x2), x2.stride(2), ptr(dy), dy.stride(2), ptr(tgt),
             B, H, W, Cip, Cop, dtype_code(dt), ptr(workspace), workspace.numel() * workspace.element_size(), acc_flag,
             stream())
    else:
        call("unetk_conv_wgrad", ptr(x), ldx, ptr(dy), dy.stride(2), ptr(tgt), B, H, W, Cip, Cop, ks,
             dtype_code(dt), ptr(workspace), workspace.numel() * workspace.element_size(), acc_flag, stream())
    if padded:
        res = tgt[:Co, :Ci]
        if dw is not None:
            if accumulate:
                dw += res
            else:
                dw.copy_(res)
            return dw
        return res.contiguous()
    return tgt


def bn_finalize(stats, count, gamma, beta, running_mean=None, running_var=None,
                eps=BN_EPS, momentum=BN_MOMENTUM):
    """Per-tile partial sums -> (mean, invstd, scale, shift); updates the running stats in place."""
    nparts, _, C = stats.shape
    dev = stats.device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)
    call("unetk_bn_finalize", ptr(stats), nparts, C, count, ptr(gamma), ptr(beta), eps, momentum,
         ptr(running_mean), ptr(running_var), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), stream())
    return out[0], out[1], out[2], out[3]


def bn_fold_eval(gamma, beta, running_mean, running_var, conv_bias=None, eps=BN_EPS):
    C = running_mean.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=running_mean.device)
    call("unetk_bn_fold_eval", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps,
         ptr(conv_bias), ptr(out[0]), ptr(out[1]), C, stream())
    return out[0], out[1]


def bn_act_fwd(y, scale, shift, keep=None, relu=True, pool=False, write_a=True, out=None,
               want_argmax=False):
    """a = relu(y*scale+shift)*keep (+ its 2x2 max-pool).  ``out`` may be a channel slice of a wider
    NHWC buffer (the skip half of a decoder's concat input).  With ``want_argmax`` the result is
    (a, pooled, argmax): the uint8 map of first maxima that ``pool_bwd`` takes instead of ``a``."""
    B, H, W, C = y.shape
    a = out if out is not None else (torch.empty_like(y) if write_a else None)
    pooled = torch.empty((B, H // 2, W // 2, C), dtype=y.dtype, device=y.device) if pool else None
    argmax = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=y.device) if (pool and want_argmax) else None
    call("unetk_bn_act_fwd", ptr(y), y.stride(2), ptr(a), a.stride(2) if a is not None else 0, ptr(pooled), ptr(argmax),
         B, H, W, C, ptr(scale), ptr(shift), ptr(keep), 1 if relu else 0, dtype_code(y.dtype), stream())
    if want_argmax:
        return a, pooled, argmax
    return a, pooled


def maxpool2x2(a):
    return bn_act_fwd(a, None, None, None, relu=False, pool=True, write_a=False)[1]


class PooledGrad:
    """Stands for pool_bwd(None, dpooled, dskip, argmax) without storing it (see bn_act_bwd)."""
    __slots__ = ("dpooled", "dskip", "argmax")

    def __init__(self, dpooled, dskip, argmax):
        self.dpooled, self.dskip, self.argmax = dpooled, dskip, argmax


def bn_act_bwd(g, y, scale, shift, mean, invstd, keep=None, relu=True, dgamma=None, dbeta=None,
               accumulate=False, presummed=None):
    """Returns (dy, dgamma, dbeta).  ``g`` is the gradient of the activation, or a pair (gpix (B,H,W) fp32,
    gchan (C,) fp32) standing for the rank-one gradient gpix[b,y,x] * gchan[c] that is never materialised, or a
    PooledGrad: the gradient through a 2x2 max-pool plus a skip gradient, formed inside the passes."""
    B, H, W, C = y.shape
    dev = y.device
    dy = torch.empty_like(y)
    sums = torch.empty((6, C), dtype=torch.float32, device=dev)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        accumulate = False
    if presummed is not None:       # (rows of the two sums left by conv_bwd_data_bnsums, which produced g)
        if keep is not None or not relu or presummed.shape[1:] != (2, C) or not presummed.is_contiguous():
            raise ValueError("bn_act_bwd: presummed partials go with ReLU, no keep factors and shape (n, 2, C)")
        call("unetk_bn_act_bwd_presummed", ptr(g), ptr(y), ptr(dy), B, H, W, C, ptr(scale), ptr(shift), ptr(mean),
             ptr(invstd), ptr(dgamma), ptr(dbeta), 1 if accumulate else 0, ptr(presummed), presummed.shape[0],
             ptr(sums), dtype_code(y.dtype), stream())
        return dy, dgamma, dbeta
    nparts = _lib.load().unetk_reduce_num_partials(B * H * W)
    partials = torch.empty((nparts, 2, C), dtype=torch.float32, device=dev)
    if isinstance(g, PooledGrad):
        assert g.dpooled.shape == (B, H // 2, W // 2, C) and g.dskip.shape == (B, H, W, C) and g.dskip.stride(3) == 1
        assert g.dpooled.is_contiguous() and g.argmax.is_contiguous() and g.dskip.stride(1) == W * g.dskip.stride(2)
        call("unetk_bn_act_bwd_pool", ptr(g.dpooled), ptr(g.argmax), ptr(g.dskip), g.dskip.stride(2), ptr(y), ptr(dy),
             B, H, W, C, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(keep), 1 if relu else 0, ptr(dgamma),
             ptr(dbeta), 1 if accumulate else 0, ptr(partials), ptr(sums), dtype_code(y.dtype), stream())
        return dy, dgamma, dbeta
    if isinstance(g, tuple):
        gpix, gchan = g
        if gchan.numel() < C:       # channel padding of the activation: the padded channels get no gradient
            gchan = torch.cat([gchan, gchan.new_zeros(C - gchan.numel())])
        call("unetk_bn_act_bwd_rank1", ptr(gpix), ptr(gchan), ptr(y), ptr(dy), B, H, W, C, ptr(scale), ptr(shift),
             ptr(mean), ptr(invstd), ptr(keep), 1 if relu else 0, ptr(dgamma), ptr(dbeta), 1 if accumulate else 0,
             ptr(partials), ptr(sums), dtype_code(y.dtype), stream())
        return dy, dgamma, dbeta
    call("unetk_bn_act_bwd", ptr(g), ptr(y), ptr(dy), B, H, W, C, ptr(scale), ptr(shift), ptr(mean),
         ptr(invstd), ptr(keep), 1 if relu else 0, ptr(dgamma), ptr(dbeta), 1 if accumulate else 0,
         ptr(partials), ptr(sums), dtype_code(y.dtype), stream())
    return dy, dgamma, dbeta


def pool_bwd(a, dpooled, dskip=None, argmax=None):
    """da = dskip + dpooled routed to the first maximum of each quad (from ``argmax`` when given,
    else recomputed from ``a``, which may then be None)."""
    B, h2, w2, C = dpooled.shape
    if a is not None:
        _, H, W, _ = a.shape
    else:
        _, H, W, _ = dskip.shape
    da = torch.empty((B, H, W, C), dtype=dpooled.dtype, device=dpooled.device)
    # dskip may be a channel slice of a wider buffer: its pixel stride is what the kernel needs
    call("unetk_pool_bwd", ptr(a) if argmax is None else None, ptr(argmax), ptr(dpooled), ptr(dskip),
         dskip.stride(2) if dskip is not None else 0, ptr(da), B, H, W, C, dtype_code(dpooled.dtype), stream())
    return da


def upsample_concat(low, skip):
    """cat(bilinear_resize(low -> skip's H,W), skip) along channels."""
    B, h, w, C1 = low.shape
    _, H, W, C2 = skip.shape
    out = torch.empty((B, H, W, C1 + C2), dtype=low.dtype, device=low.device)
    call("unetk_upsample_concat_fwd", ptr(low), h, w, C1, ptr(skip), skip.stride(2), C2, ptr(out), B, H, W,
         dtype_code(low.dtype), stream())
    return out


def upsample_into(low, cat):
    """Writes bilinear_resize(low -> cat's H,W) into channels [0,C1) of the dense concat buffer
    ``cat`` (B,H,W,C1+C2) whose skip half is already in place."""
    B, h, w, C1 = low.shape
    _, H, W, C = cat.shape
    assert cat.is_contiguous() and C > C1
    call("unetk_upsample_concat_fwd", ptr(low), h, w, C1, None, 0, C - C1, ptr(cat), B, H, W,
         dtype_code(low.dtype), stream())
    return cat


def upsample(low, H, W):
    """bilinear_resize(low -> H,W) (align_corners=False) as a dense (B,H,W,C) tensor."""
    B, h, w, C1 = low.shape
    out = torch.empty((B, H, W, C1), dtype=low.dtype, device=low.device)
    call("unetk_upsample_concat_fwd", ptr(low), h, w, C1, None, 0, 0, ptr(out), B, H, W, dtype_code(low.dtype), stream())
    return out


def bn_act_upsample_supported(y, *factors):
    """factors: the scale / shift / keep tensors, when known (their 16-byte alignment is part of the answer)."""
    B, h, w, C1 = y.shape
    return bool(_lib.load().unetk_bn_act_upsample2x_supported(h, w, C1, dtype_code(y.dtype))) and \
        all(f is None or f.data_ptr() % 16 == 0 for f in factors)


def bn_act_upsample(y, scale, shift, keep, cat=None):
    """resize2x(relu(y*scale+shift)*keep) without storing the activation: into channels [0,C1) of the concat buffer
    ``cat`` (B,2h,2w,C1+C2), or as a dense (B,2h,2w,C1) tensor.  The same bits as bn_act_fwd + upsample(_into)."""
    B, h, w, C1 = y.shape
    assert y.is_contiguous()
    if cat is None:
        out = torch.empty((B, 2 * h, 2 * w, C1), dtype=y.dtype, device=y.device)
    else:
        assert cat.is_contiguous() and cat.shape[:3] == (B, 2 * h, 2 * w) and cat.shape[3] > C1
        out = cat
    call("unetk_bn_act_upsample2x", ptr(y), B, h, w, C1, ptr(scale), ptr(shift), ptr(keep), ptr(out), out.shape[3],
         dtype_code(y.dtype), stream())
    return out


def upsample_bwd(dcat, C1, h, w):
    """Gradient of the resized channels [0,C1) of dcat (B,H,W,>=C1; may be a channel slice) -> (B,h,w,C1)."""
    B, H, W, _ = dcat.shape
    ld = dcat.stride(2)
    dlow = torch.empty((B, h, w, C1), dtype=dcat.dtype, device=dcat.device)
    call("unetk_upsample_bwd", ptr(dcat), ld, H, W, ptr(dlow), B, h, w, C1, dtype_code(dcat.dtype), stream())
    return dlow


class _ResizePlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        B, C, h, w = x.shape
        xin = x.detach().contiguous().float()
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        call("unetk_resize_planes_f32", ptr(xin), ptr(out), B * C, h, w, H, W, 0, stream())
        ctx.shape = (B, C, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, h, w, H, W = ctx.shape
        g = g.detach().contiguous().float()
        dx = torch.empty((B, C, h, w), dtype=torch.float32, device=g.device)
        call("unetk_resize_planes_f32", ptr(g), ptr(dx), B * C, h, w, H, W, 1, stream())
        return dx, None, None


def interpolate_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """``F.interpolate(x, size=size, mode='bilinear', align_corners=False)`` for an fp32 NCHW tensor on
    the GPU, with autograd: what the Swin trainer applies to the half-resolution prediction before the
    loss (swin_unet_model.py:812-814)."""
    _require_cuda(x, "input")
    if x.dim() != 4:
        raise RuntimeError(f"expected (B,C,h,w), got {tuple(x.shape)}")
    H, W = int(size[0]), int(size[1])
    return _ResizePlanes.apply(x, H, W)


def head1x1_fwd(x, w, bias):
    """x (B,H,W,C) NHWC, w (Co,C,1,1) fp32 -> (B,Co,H,W) fp32 NCHW."""
    B, H, W, C = x.shape
    Co = w.shape[0]
    out = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
    call("unetk_head1x1_fwd", ptr(x), C, ptr(w), ptr(bias), ptr(out), B, H, W, C, Co, dtype_code(x.dtype),
         stream())
    return out


def head1x1_bwd(x, w, dout, need_dx=True):
    """Returns (dx NHWC or None, dw (Co,C,1,1), db (Co,))."""
    B, H, W, C = x.shape
    Co = w.shape[0]
    dev = x.device
    dout = dout.contiguous().float()
    dx = torch.empty_like(x) if need_dx else None
    if Co != 1 and B != 1:
        # general case: one reduction per output channel over a channel-major copy of dout
        dw = torch.empty((Co, C), dtype=torch.float32, device=dev)
        db = torch.empty(Co, dtype=torch.float32, device=dev)
        nparts = _lib.load().unetk_reduce_num_partials(B * H * W)
        partials = torch.empty((nparts, 2, C), dtype=torch.float32, device=dev)
        if need_dx:
            call("unetk_head1x1_bwd", ptr(x), C, ptr(w), ptr(dout), ptr(dx), None, None, B, H, W, C, Co, 0,
                 ptr(partials), dtype_code(x.dtype), stream())
        for o in range(Co):
            d_o = dout[:, o].contiguous()
            call("unetk_head1x1_bwd", ptr(x), C, ptr(w), ptr(d_o), None, ptr(dw[o]), ptr(db[o:o + 1]), B, H,
                 W, C, 1, 0, ptr(partials), dtype_code(x.dtype), stream())
        return dx, dw.view(Co, C, 1, 1), db
    dw = torch.empty((Co, C), dtype=torch.float32, device=dev)
    db = torch.empty(Co, dtype=torch.float32, device=dev)
    nparts = _lib.load().unetk_reduce_num_partials(B * H * W)
    partials = torch.empty((nparts, 2, C), dtype=torch.float32, device=dev)
    call("unetk_head1x1_bwd", ptr(x), C, ptr(w), ptr(dout), ptr(dx), ptr(dw), ptr(db), B, H, W, C, Co, 0,
         ptr(partials), dtype_code(x.dtype), stream())
    return dx, dw.view(Co, C, 1, 1), db


# ---------------------------------------------------------------------------- Swin-UNet pieces
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    """x (..., C) contiguous -> (y, mean, rstd) with per-token statistics in fp32."""
    C = x.shape[-1]
    ntok = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(ntok, dtype=torch.float32, device=x.device)
    rstd = torch.empty(ntok, dtype=torch.float32, device=x.device)
    call("unetk_layernorm_fwd", ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ntok, C, eps,
         dtype_code(x.dtype), stream())
    return y, mean, rstd


class DeferredSums:
    """Column sums of partial tables, taken later by one launch (the dgamma / dbeta of every LayerNorm of a
    backward pass: two dozen 8-microsecond launches otherwise).  The outputs exist from the start; they hold
    their values once flush() has run, in stream order."""

    _TABLES = {}      # (device, kind) -> (key, device table, pinned host copy): the last job table, re-used while nothing moved

    def __init__(self):
        self.jobs = []       # (partials, rows, C, out0, out1): column sums of [rows][2][C] tables
        self.tables = []     # (partials, windows, heads, dtable): relative-position bias table gradients

    def add(self, partials, rows, C, out0, out1):
        self.jobs.append((partials, rows, C, out0, out1))

    def add_table(self, partials, nwin, heads, dtable):
        self.tables.append((partials, nwin, heads, dtable))

    @staticmethod
    def _upload(dev, kind, key):
        hit = DeferredSums._TABLES.get((dev, kind))
        if hit is None or hit[0] != key:
            host = torch.tensor(key, dtype=torch.int64).pin_memory()
            hit = (key, host.to(dev, non_blocking=True), host)
            DeferredSums._TABLES[(dev, kind)] = hit
        return hit[1]

    def flush(self):
        if not self.jobs and not self.tables:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("DeferredSums.flush() under HIP graph capture: its job table is uploaded from host "
                               "memory, which a replayed graph would read again after it has been released")
        if self.jobs:
            dev = self.jobs[0][0].device
            key = tuple(v for p, r, C, o0, o1 in self.jobs
                        for v in (p.data_ptr(), r, C, o0.data_ptr() if o0 is not None else 0,
                                  o1.data_ptr() if o1 is not None else 0))
            call("unetk_partial_sums_batched", ptr(self._upload(dev, "sums", key)), len(self.jobs),
                 max(j[2] for j in self.jobs), stream())
        if self.tables:
            dev = self.tables[0][0].device
            key = tuple(v for p, n, h, t in self.tables for v in (p.data_ptr(), n, h, t.data_ptr()))
            call("unetk_dtable_reduce_batched", ptr(self._upload(dev, "tables", key)), len(self.tables),
                 max(j[2] for j in self.tables), stream())
        self.jobs, self.tables = [], []


def layernorm_bwd(dy, x, gamma, mean, rstd, add=None, need_param_grads=True, defer=None, scale_next=None):
    """Returns (dx [+ add], dgamma, dbeta).  defer: a DeferredSums that takes the two sums later (the returned
    dgamma / dbeta are filled by its flush()).  scale_next = (keep (B,) fp32, tokens per sample), with defer only:
    a fourth result, dx times the per-sample factor (scale_sample(dx, keep) without its launch), or None when the
    shape has no one-pass form."""
    C = x.shape[-1]
    ntok = x.numel() // C
    dev = x.device
    dx = torch.empty_like(x)
    if defer is not None and need_param_grads:
        rows = _lib.load().unetk_layernorm_bwd_rows(ntok, C, dtype_code(x.dtype))
        if rows > 0:
            partials = torch.empty((rows, 2, C), dtype=torch.float32, device=dev)
            dg = torch.empty(C, dtype=torch.float32, device=dev)
            db = torch.empty(C, dtype=torch.float32, device=dev)
            dxs = torch.empty_like(x) if scale_next is not None else None
            call("unetk_layernorm_bwd_partials", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(add), ptr(dx),
                 ptr(partials), ntok, C, dtype_code(x.dtype), ptr(scale_next[0]) if scale_next is not None else None,
                 scale_next[1] if scale_next is not None else 0, ptr(dxs), stream())
            defer.add(partials, rows, C, dg, db)
            return (dx, dg, db) if scale_next is None else (dx, dg, db, dxs)
    nparts = _lib.load().unetk_token_reduce_num_partials(ntok)
    partials = torch.empty((nparts, 2, C), dtype=torch.float32, device=dev)
    dg = torch.empty(C, dtype=torch.float32, device=dev) if need_param_grads else None
    db = torch.empty(C, dtype=torch.float32, device=dev) if need_param_grads else None
    call("unetk_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(add), ptr(dx), ptr(dg),
         ptr(db), ptr(partials), ntok, C, dtype_code(x.dtype), stream())
    return (dx, dg, db) if scale_next is None else (dx, dg, db, None)


def channel_sum(x, C=None):
    """Sum over all tokens of a (..., ld) tensor, first C channels -> (C,) fp32."""
    ld = x.stride(-2) if x.dim() > 1 else x.shape[-1]
    C = C if C is not None else x.shape[-1]
    ntok = x.numel() // x.shape[-1]
    nparts = _lib.load().unetk_token_reduce_num_partials(ntok)
    partials = torch.empty((nparts, 2, C), dtype=torch.float32, device=x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    call("unetk_channel_sum", ptr(x), ld, ntok, C, ptr(out), ptr(partials), dtype_code(x.dtype), stream())
    return out


def swin_mlp_supported(x) -> bool:
    """Whether the one-launch MLP branch takes these tokens (bfloat16, dense, 96 or 192 channels)."""
    C = x.shape[-1]
    return bool(x.dtype == torch.bfloat16 and x.is_contiguous() and
                _lib.load().unetk_swin_mlp_supported(x.numel() // C, C, dtype_code(x.dtype)))


def swin_mlp_fwd(x, gamma, beta, eps, w1_pack64: PackedConv, b1, w2_pack96: PackedConv, b2, keep=None, want_act=True):
    """out = x + keep[sample] * fc2(GELU(fc1(LayerNorm(x)))) in one launch.  Returns (out, ln, mean, rstd, hpre,
    hact or None): everything the backward pass of the branch reads."""
    C = x.shape[-1]
    M = x.numel() // C
    if w1_pack64.bn != 64 or w2_pack96.bn != 96 or (w1_pack64.Co, w1_pack64.Ci) != (4 * C, C) or \
            (w2_pack96.Co, w2_pack96.Ci) != (C, 4 * C):
        raise ValueError("swin_mlp_fwd: fc1 must be packed in 64-channel blocks, fc2 in 96-channel blocks")
    dev = x.device
    ln = torch.empty_like(x)
    out = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=dev)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    hpre = torch.empty(x.shape[:-1] + (4 * C,), dtype=x.dtype, device=dev)
    hact = torch.empty_like(hpre) if want_act else None
    call("unetk_swin_mlp_fwd", ptr(x), ptr(gamma), ptr(beta), float(eps), ptr(w1_pack64.data), ptr(b1),
         ptr(w2_pack96.data), ptr(b2), ptr(keep), M // x.shape[0], ptr(ln), ptr(mean), ptr(rstd), ptr(hpre), ptr(hact),
         ptr(out), M, C, dtype_code(x.dtype), stream())
    return out, ln, mean, rstd, hpre, hact


def swin_mlp_bwd(dy, hpre, w2t_pack64: PackedConv, w1t_pack96: PackedConv, x, mean, rstd, gamma, add, defer,
                 scale_next=None):
    """The data-gradient chain of the MLP branch in one launch: returns (dhp, dx, dgamma, dbeta, dx_scaled or None).
    dgamma / dbeta are filled by ``defer.flush()`` (a DeferredSums).  scale_next = (keep (B,), tokens per sample)."""
    C = x.shape[-1]
    M = x.numel() // C
    if w2t_pack64.bn != 64 or w1t_pack96.bn != 96 or (w2t_pack64.Co, w2t_pack64.Ci) != (4 * C, C) or \
            (w1t_pack96.Co, w1t_pack96.Ci) != (C, 4 * C):
        raise ValueError("swin_mlp_bwd: fc2 must be packed transposed in 64-channel blocks, fc1 transposed in 96-channel blocks")
    dev = x.device
    rows = _lib.load().unetk_swin_mlp_bwd_rows(M, C)
    partials = torch.empty((rows, 2, C), dtype=torch.float32, device=dev)
    dg = torch.empty(C, dtype=torch.float32, device=dev)
    db = torch.empty(C, dtype=torch.float32, device=dev)
    dhp = torch.empty_like(hpre)
    dx = torch.empty_like(x)
    keep, tps = (scale_next if scale_next is not None else (None, 1))
    dxs = torch.empty_like(x) if keep is not None else None
    call("unetk_swin_mlp_bwd", ptr(dy), ptr(hpre), ptr(w2t_pack64.data), ptr(w1t_pack96.data), ptr(x), ptr(mean),
         ptr(rstd), ptr(gamma), ptr(add), ptr(keep), tps, ptr(dhp), ptr(dx), ptr(dxs), ptr(partials), M, C,
         dtype_code(x.dtype), stream())
    defer.add(partials, rows, C, dg, db)
    return dhp, dx, dg, db, dxs


def gelu_fwd(x):
    y = torch.empty_like(x)
    call("unetk_gelu_fwd", ptr(x), ptr(y), x.numel(), dtype_code(x.dtype), stream())
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    call("unetk_gelu_bwd", ptr(dy), ptr(x), ptr(dx), x.numel(), dtype_code(x.dtype), stream())
    return dx


def scale_sample(x, keep=None, add=None):
    """x * keep[b] + add for a (B, ...) tensor."""
    y = torch.empty_like(x)
    B = x.shape[0]
    call("unetk_scale_sample", ptr(x), ptr(keep), ptr(add), ptr(y), B, x.numel() // B, dtype_code(x.dtype),
         stream())
    return y


def patchify(x, ps, dt):
    x = x.contiguous().float()
    B, Cin, H, W = x.shape
    ld = pad_channels(Cin * ps * ps, dt)
    tok = torch.empty((B, H // ps, W // ps, ld), dtype=dt, device=x.device)
    call("unetk_patchify", ptr(x), ptr(tok), B, Cin, H, W, ps, ld, dtype_code(dt), stream())
    return tok


def unpatchify(dtok, Cin, H, W, ps):
    B = dtok.shape[0]
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dtok.device)
    call("unetk_unpatchify", ptr(dtok), ptr(dx), B, Cin, H, W, ps, dtok.shape[-1], dtype_code(dtok.dtype),
         stream())
    return dx


def merge_gather(x):
    B, H, W, C = x.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), dtype=x.dtype, device=x.device)
    call("unetk_merge_gather", ptr(x), ptr(out), B, H, W, C, 0, dtype_code(x.dtype), stream())
    return out


def merge_scatter(dout, H, W):
    B, _, _, C4 = dout.shape
    dx = torch.empty((B, H, W, C4 // 4), dtype=dout.dtype, device=dout.device)
    call("unetk_merge_gather", ptr(dout), ptr(dx), B, H, W, C4 // 4, 1, dtype_code(dout.dtype), stream())
    return dx


def pixel_shuffle(tmp, cat, bias, Co, backward=False):
    """tmp (B,h,w,4*Co) <-> channels [0,Co) of cat (B,2h,2w,ld)."""
    B, h, w, _ = tmp.shape
    call("unetk_pixel_shuffle", ptr(tmp), ptr(cat), cat.stride(2), ptr(bias), B, h, w, Co, 1 if backward else 0,
         dtype_code(tmp.dtype), stream())


def window_attn_is_mfma(C, heads, ws):
    """Window 8 with head dimension 32: the MFMA kernels; anything else goes to the general ones."""
    return ws == 8 and C == heads * 32


def window_attn_fwd(qkv, table, heads, shift, use_mask, ws=8):
    B, He, We, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty((B, He, We, C), dtype=qkv.dtype, device=qkv.device)
    if not window_attn_is_mfma(C, heads, ws):
        call("unetk_window_attn_any_fwd", ptr(qkv), ptr(out), ptr(table), B, He, We, C, heads, ws, shift,
             1 if use_mask else 0, (C // heads) ** -0.5, dtype_code(qkv.dtype), stream())
        return out
    call("unetk_window_attn_fwd", ptr(qkv), ptr(out), ptr(table), B, He, We, C, heads, shift,
         1 if use_mask else 0, (C // heads) ** -0.5, dtype_code(qkv.dtype), stream())
    return out


def window_attn_bwd(qkv, dout, table, heads, shift, use_mask, defer=None, ws=8):
    """Returns (dqkv, dtable ((2*ws-1)**2, heads)).  defer: a DeferredSums that sums the table's partials later
    (window 8 / head dimension 32 only; the general kernels' partials are added here)."""
    B, He, We, C3 = qkv.shape
    C = C3 // 3
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    if not window_attn_is_mfma(C, heads, ws):
        t2 = (2 * ws - 1) ** 2
        nwin = B * (He // ws) * (We // ws)
        partials = torch.empty((nwin, heads, t2), dtype=torch.float32, device=dev)
        call("unetk_window_attn_any_bwd", ptr(qkv), ptr(dout), ptr(dqkv), ptr(table), ptr(partials), B, He, We, C, heads,
             ws, shift, 1 if use_mask else 0, (C // heads) ** -0.5, dtype_code(qkv.dtype), stream())
        return dqkv, partials.sum(0).t().contiguous()
    nblk = B * (He // 8) * (We // 8) * heads
    partials = torch.empty((nblk, 225), dtype=torch.float32, device=dev)
    dtable = torch.empty((225, heads), dtype=torch.float32, device=dev)
    call("unetk_window_attn_bwd", ptr(qkv), ptr(dout), ptr(dqkv), ptr(table), None if defer is not None else ptr(dtable),
         ptr(partials), B, He, We, C, heads, shift, 1 if use_mask else 0, (C // heads) ** -0.5, dtype_code(qkv.dtype),
         stream())
    if defer is not None:
        defer.add_table(partials, nblk // heads, heads, dtable)
    return dqkv, dtable
