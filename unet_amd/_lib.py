"""ctypes binding of libunetk.so (the C ABI declared in include/unetk.h).

There is no CPU fallback: if the library is missing every call raises.  Device memory and
streams come from torch; the kernels take raw device pointers.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunetk.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
OK, EINVAL, ESHAPE, EHIP = 0, -1, -2, -3


class UnetkError(RuntimeError):
    pass


class UnetkMissing(ImportError):
    pass


P, I, L, F, D = c_void_p, c_int, c_int64, c_float, c_double

# name -> (restype, argtypes); mirrors include/unetk.h one to one
_SIGNATURES = {
    "unetk_last_error": (c_char_p, []),
    "unetk_version": (I, []),
    "unetk_stream_fork": (I, [P, P]),
    "unetk_conv_pick_bn": (I, [I]),
    "unetk_conv_packed_elems": (L, [I, I, I, I, I]),
    "unetk_conv_pack_weight": (I, [P, P, I, I, I, I, I, I, P]),
    "unetk_conv_pack_groups": (I, [I, I, I, I, I]),
    "unetk_conv_pack_weights": (I, [P, P, I, P, I, P]),
    "unetk_conv_stats_rows": (I, [I] * 10),
    "unetk_conv_num_tiles": (I, [I, I, I, I]),
    "unetk_conv_fwd": (I, [P, I, P, P, I, I, I, I, I, I, I, I, I, P, P, P, I, P, I, P, P]),
    "unetk_conv_fwd_fused": (I, [P, I, P, P, I, P, I, I, P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, I, P, I, P, P]),
    "unetk_conv_fwd_split": (I, [P, I, P, P, I, P, I, I, I, I, I, I, I, I, I, I, P, P, P, I, P, I, P, P]),
    "unetk_conv_wgrad_workspace_bytes": (L, [I, I, I, I, I, I, I]),
    "unetk_conv_fwd_cat": (I, [P, I, I, P, I, P, P, I, I, I, I, I, I, I, I, P, P, P, I, P]),
    "unetk_conv_bwd_data_bnsums_supported": (I, [I, I, I]),
    "unetk_conv_bwd_data_bnsums": (I, [P, I, P, P, I, I, I, I, I, I, I, P, I, P, P, P, P, P, P]),
    "unetk_conv_wgrad_cat": (I, [P, I, I, P, I, P, I, P, I, I, I, I, I, I, P, L, I, P]),
    "unetk_conv_wgrad": (I, [P, I, P, I, P, I, I, I, I, I, I, I, P, L, I, P]),
    "unetk_linear_wgrad_workspace_bytes": (L, [L, I, I, I]),
    "unetk_linear_wgrad": (I, [P, I, P, I, P, P, L, I, I, I, P, L, I, P]),
    "unetk_linear_supported": (I, [L, I, I, I, I]),
    "unetk_linear_fwd": (I, [P, I, P, P, I, P, I, L, I, I, I, I, P, I, P, I, P, L, P, I, P]),
    "unetk_swin_mlp_supported": (I, [L, I, I]),
    "unetk_swin_mlp_fwd": (I, [P, P, P, F, P, P, P, P, P, L, P, P, P, P, P, P, L, I, I, P]),
    "unetk_swin_mlp_bwd_rows": (I, [L, I]),
    "unetk_swin_mlp_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, L, I, I, P]),
    "unetk_token_wgrad_supported": (I, [L, I, I, I]),
    "unetk_token_wgrad_workspace_bytes": (L, [L, I, I]),
    "unetk_token_wgrad": (I, [P, I, P, I, P, P, L, I, I, I, P, L, I, P]),
    "unetk_optim_chunk_elems": (I, []),
    "unetk_grad_norm": (I, [P, P, P, I, F, P, P, P]),
    "unetk_scale_list": (I, [P, P, P, I, P, P]),
    "unetk_adamw_step": (I, [P, P, P, P, P, P, I, D, D, D, D, D, L, P, I, P]),
    "unetk_adamw_step_device": (I, [P, P, P, P, P, P, I, P, P, D, P, I, D, D, P]),
    "unetk_patch_qc": (I, [P, I, P, P, I, I, I, I, D, P, P]),
    "unetk_patch_gather": (I, [P, I, P, P, I, I, P, P, I, I, I, P, P, P, P]),
    "unetk_metrics_num_partials": (I, [L]),
    "unetk_metrics_update": (I, [P, P, L, P, P, P]),
    "unetk_gauss_rows_max_radius": (I, []),
    "unetk_gauss_rows_f64": (I, [P, P, P, I, I, P, I, I, P]),
    "unetk_transpose_f64": (I, [P, P, I, I, P]),
    "unetk_nchw_to_nhwc": (I, [P, P, I, I, I, I, I, I, P]),
    "unetk_nhwc_to_nchw": (I, [P, I, P, I, I, I, I, I, P]),
    "unetk_bn_finalize": (I, [P, I, I, L, P, P, F, F, P, P, P, P, P, P, P]),
    "unetk_bn_fold_eval": (I, [P, P, P, P, F, P, P, P, I, P]),
    "unetk_bn_act_fwd": (I, [P, I, P, I, P, P, I, I, I, I, P, P, P, I, I, P]),
    "unetk_reduce_num_partials": (I, [L]),
    "unetk_bn_act_bwd": (I, [P, P, P, I, I, I, I, P, P, P, P, P, I, P, P, I, P, P, I, P]),
    "unetk_bn_act_bwd_rank1": (I, [P, P, P, P, I, I, I, I, P, P, P, P, P, I, P, P, I, P, P, I, P]),
    "unetk_bn_act_bwd_presummed": (I, [P, P, P, I, I, I, I, P, P, P, P, P, P, I, P, I, P, I, P]),
    "unetk_bn_act_bwd_pool": (I, [P, P, P, I, P, P, I, I, I, I, P, P, P, P, P, I, P, P, I, P, P, I, P]),
    "unetk_pool_bwd": (I, [P, P, P, P, I, P, I, I, I, I, I, P]),
    "unetk_upsample_concat_fwd": (I, [P, I, I, I, P, I, I, P, I, I, I, I, P]),
    "unetk_upsample_bwd": (I, [P, I, I, I, P, I, I, I, I, I, P]),
    "unetk_bn_act_upsample2x_supported": (I, [I, I, I, I]),
    "unetk_bn_act_upsample2x": (I, [P, I, I, I, I, P, P, P, P, I, I, P]),
    "unetk_resize_planes_f32": (I, [P, P, L, I, I, I, I, I, P]),
    "unetk_head1x1_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "unetk_head1x1_bwd": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "unetk_loss_num_partials": (I, [I, I, I]),
    "unetk_loss_spectral_blocks": (I, [I, I]),
    "unetk_loss_pixgrad_fwd": (I, [P, P, P, I, I, I, I, I, I, F, P, P, P, P, P]),
    "unetk_loss_static_max_pixels": (L, []),
    "unetk_loss_pixgrad_fwd_static": (I, [P, P, P, I, I, I, I, I, I, F, P, P, P, P]),
    "unetk_loss_spectral_fwd": (I, [P, P, P, I, I, I, F, P, P, P, P, P]),
    "unetk_loss_combine": (I, [P, F, F, F, I, I, P]),
    "unetk_loss_pixgrad_bwd": (I, [P, P, P, I, I, I, I, I, P, P, P, P, F, F, I, P, I, P]),
    "unetk_loss_spectral_bwd": (I, [P, P, I, I, I, F, P, F, P, P, P]),
    "unetk_tile_valid": (I, [P, I, I, I, I, I, I, I, P, P]),
    "unetk_tile_gather": (I, [P, I, I, I, P, I, I, P, I, I, P]),
    "unetk_blend_accumulate": (I, [P, P, P, I, I, I, I, I, I, P, P, I, I, I, I, P]),
    "unetk_blend_normalize": (I, [P, P, P, L, P]),
    "unetk_layernorm_fwd": (I, [P, P, P, P, P, P, L, I, F, I, P]),
    "unetk_token_reduce_num_partials": (I, [L]),
    "unetk_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L, I, I, P]),
    "unetk_layernorm_bwd_rows": (I, [L, I, I]),
    "unetk_layernorm_bwd_partials": (I, [P, P, P, P, P, P, P, P, L, I, I, P, L, P, P]),
    "unetk_partial_sums_batched": (I, [P, I, I, P]),
    "unetk_channel_sum": (I, [P, I, L, I, P, P, I, P]),
    "unetk_gelu_fwd": (I, [P, P, L, I, P]),
    "unetk_gelu_bwd": (I, [P, P, P, L, I, P]),
    "unetk_scale_sample": (I, [P, P, P, P, I, L, I, P]),
    "unetk_patchify": (I, [P, P, I, I, I, I, I, I, I, P]),
    "unetk_unpatchify": (I, [P, P, I, I, I, I, I, I, I, P]),
    "unetk_merge_gather": (I, [P, P, I, I, I, I, I, I, P]),
    "unetk_pixel_shuffle": (I, [P, P, I, P, I, I, I, I, I, I, P]),
    "unetk_convt2x2_fwd": (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, P]),
    "unetk_convt2x2_bwd_data": (I, [P, I, P, P, I, I, I, I, I, I, I, I, P]),
    "unetk_window_attn_fwd": (I, [P, P, P, I, I, I, I, I, I, I, F, I, P]),
    "unetk_window_attn_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P]),
    "unetk_dtable_reduce_batched": (I, [P, I, I, P]),
    "unetk_window_attn_any_supported": (I, [I, I]),
    "unetk_window_attn_any_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, F, I, P]),
    "unetk_window_attn_any_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P]),
}

_lib = None


def exported_symbols():
    """Names the header declares; used by the CPU test that the library exports them all."""
    return sorted(_SIGNATURES)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnetkMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().unetk_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc == OK:
        return
    msg = last_error()
    if rc in (EINVAL, ESHAPE):
        raise ValueError(f"unetk: {msg}")
    raise UnetkError(f"unetk (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_SLOW_STREAM = os.environ.get("UNETK_STREAM_VIA_TORCH") == "1"       # A/B switch: the Stream-object route


def stream():
    """The current HIP stream of the current device as a ctypes pointer.  Every kernel launch asks for it: the
    raw accessors take well under a microsecond, torch.cuda.current_stream() builds a Stream object (~9)."""
    if _raw_stream is not None and _raw_device is not None and not _SLOW_STREAM:
        return c_void_p(_raw_stream(_raw_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f"unetk: unsupported compute dtype {dt} (float32 or bfloat16)")


def call(name, *args):
    check(getattr(load(), name)(*args))
