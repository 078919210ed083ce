"""ctypes binding of libwc_hip.so -- the only way the Python host reaches the HIP kernels.

There is deliberately no CPU or PyTorch fallback: if the library is missing or a call fails the
host raises, so a GPU run can never silently pass on something other than the HIP path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwc_hip.so")

WC_OK = 0
ABI_VERSION = 8          # WC_ABI_VERSION of include/wc_hip.h
ERRORS = {-1: "WC_ERR_NULL", -2: "WC_ERR_SHAPE", -3: "WC_ERR_CHANNELS", -4: "WC_ERR_WORKSPACE", -5: "WC_ERR_ARG"}

# name -> (restype, argtypes); mirrors include/wc_hip.h one to one
SIGNATURES = {
    "wc_abi_version": (c_int, []),
    "wc_error_string": (c_char_p, [c_int]),
    "wc_stats_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wc_factor_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_color_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_bwd_reduce_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    "wc_bwd_factor_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_apply_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "wc_apply_plan_bytes": (c_size_t, [c_int, c_int]),
    "wc_bwd_apply_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "wc_stats_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_group_bias_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "wc_factor_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_double, c_double, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_color_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p]),
    "wc_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_apply_act_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_reduce_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_reduce_scaled_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_reduce_relu_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_relu_mask_bytes": (c_size_t, [c_int64, c_int]),
    "wc_relu_mask_apply_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "wc_apply_mask_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_whiten_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wc_factor_error_offset": (c_size_t, [c_int, c_int]),
    "wc_whiten_error_offset": (c_size_t, [c_int64, c_int, c_int]),
    "wc_whiten_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_apply_planes_supported": (c_int, [c_int64, c_int64, c_int]),
    "wc_apply_planes_scale_floats": (c_size_t, []),
    "wc_out_scale_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "wc_apply_planes_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wc_bwd_reduce_mask_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_bits_supported": (c_int, [c_int64, c_int64, c_int, c_int]),
    "wc_bwd_reduce_bits_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_apply_bits_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_apply_scaled_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_factor_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                  c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "wc_bwd_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_split_bytes": (c_size_t, [c_int64, c_int]),
    "wc_split_scales_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wc_split_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "wc_unsplit_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "wc_stats_split_supported": (c_int, [c_int64, c_int, c_int]),
    "wc_stats_split_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wc_stats_split_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_split_bias_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "wc_apply_split_supported": (c_int, [c_int64, c_int64, c_int]),
    "wc_apply_split_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_apply_split_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                     c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_apply_split_ex_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                        c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_whiten_split_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wc_whiten_split_error_offset": (c_size_t, [c_int64, c_int, c_int]),
    "wc_whiten_split_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_double, c_double, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_resadd_split_supported": (c_int, [c_int64, c_int64, c_int64, c_int]),
    "wc_resadd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "wc_resadd_stats_supported": (c_int, [c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "wc_resadd_stats_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int]),
    "wc_resadd_stats_split_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_whiten_presummed_error_offset": (c_size_t, [c_int64, c_int, c_int]),
    "wc_whiten_presummed_f16x2": (c_int, [c_void_p, c_int64, c_int, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_stats_presummed_f16x2": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_resadd_split_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "wc_patch_sum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "wc_fold_channel_scale_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wc_unfold_channel_scale_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wc_color_split_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_factor_mix_supported": (c_int, [c_int, c_int]),
    "wc_factor_mix_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "wc_factor_mix_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_factor_mix_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "wc_group_bias_centered_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "wc_bwd_xsplit_supported": (c_int, [c_int64, c_int64, c_int, c_int]),
    "wc_bwd_reduce_xsplit_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_bwd_apply_xsplit_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_bwd_apply_xsplit_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_stream_copy_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "wc_spectral_norm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wc_spectral_norm_batched_f32": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    "wc_spectral_norm_bwd_batched_f32": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "wc_spectral_norm_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_spectral_norm_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "wc_conv_supported": (c_int, [c_void_p]),
    "wc_conv_split_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wc_conv_weights_bytes": (c_size_t, [c_void_p]),
    "wc_conv_weights_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p]),
    "wc_spectral_norm_amax_offset": (c_size_t, [c_int, c_int]),
    "wc_spectral_norm_error_offset": (c_size_t, [c_int, c_int]),
    "wc_conv_split_hist_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "wc_conv_fwd_narrow_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                       c_int, c_void_p, c_void_p]),
    "wc_conv_wrw_narrow_supported": (c_int, [c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "wc_conv_wrw_narrow_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int, c_int]),
    "wc_conv_wrw_narrow_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int64,
                                       c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_conv_split_colsum_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "wc_conv_wrw_bias_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wc_conv_weights_pair_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                         c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "wc_conv_wrw_workspace_bytes": (c_size_t, [c_void_p]),
    "wc_conv_wrw_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "wc_conv_workspace_bytes": (c_size_t, [c_void_p]),
    "wc_conv_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p, c_size_t, c_void_p]),
}



class SnItem(ctypes.Structure):          # wc_sn_item
    _fields_ = [("W", c_void_p), ("u", c_void_p), ("v", c_void_p), ("w_sn", c_void_p), ("sigma", c_void_p),
                ("u_used", c_void_p), ("v_used", c_void_p), ("ws", c_void_p), ("rows", c_int), ("cols", c_int)]


class SnBwdItem(ctypes.Structure):       # wc_sn_bwd_item
    _fields_ = [("g", c_void_p), ("w_sn", c_void_p), ("u", c_void_p), ("v", c_void_p), ("sigma", c_void_p),
                ("dW", c_void_p), ("ws", c_void_p), ("rows", c_int), ("cols", c_int)]


class ConvGeom(ctypes.Structure):        # wc_conv_geom
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("Hin", c_int), ("Win", c_int), ("Cin", c_int),
                ("Hout", c_int), ("Wout", c_int), ("Cout", c_int), ("in_stride", c_int), ("out_stride", c_int),
                ("ntaps", c_int), ("nphase", c_int),
                ("dy", (ctypes.c_byte * 16) * 4), ("dx", (ctypes.c_byte * 16) * 4),
                ("off_y", ctypes.c_byte * 4), ("off_x", ctypes.c_byte * 4),
                ("nsrc", (ctypes.c_byte * 16) * 4),
                ("wr", ((ctypes.c_byte * 4) * 16) * 4), ("ws", ((ctypes.c_byte * 4) * 16) * 4),
                ("wcoef", c_float)]


_lib = None


class WcHipError(RuntimeError):
    pass


def shared_gpu_guard(device_count=None) -> bool:
    """K2's one-launch forms (the relay, and rounds 2-5's kernel with the inverse inside) wait -- bounded -- for workgroups of their own
    launch: fine when the process has the GPU, wrong when several ranks TIME-SLICE one GPU (a waiting workgroup can be switched out
    with the one it waits for; the wait runs out, W holds NaN and only WC_CHECK_K2=1 would say so).  The two-launch form has no such
    wait.  Until round 6 the user had to know the switch; now a launcher that puts more local ranks on the node than there are visible
    devices (LOCAL_WORLD_SIZE > device count) selects it here, before the library reads its environment.  An explicit
    WC_K2_TWO_LAUNCH (0 or 1) is left alone.  Returns whether the switch was set by this call."""
    if "WC_K2_TWO_LAUNCH" in os.environ:
        if os.environ["WC_K2_TWO_LAUNCH"] == "0":       # (the C side tests for presence: "0" means "not set")
            del os.environ["WC_K2_TWO_LAUNCH"]
        return False
    try:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    except ValueError:
        ranks = 1
    if ranks <= 1:
        return False
    if any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")):
        return False        # (a launcher that masks devices per rank: the count seen here says nothing about sharing)
    if device_count is None:
        try:
            import torch
            device_count = torch.cuda.device_count()      # (does not initialise the GPU)
        except Exception:
            device_count = 0
    if device_count and ranks > device_count:
        os.environ["WC_K2_TWO_LAUNCH"] = "1"
        return True
    return False


def load() -> ctypes.CDLL:
    """Load libwc_hip.so (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libwc_hip.so must bind to the HIP runtime torch has loaded (same streams, same
    # allocations); loading it before torch pulls in a second runtime that sees no device
    import torch  # noqa: F401
    shared_gpu_guard()
    if not os.path.exists(LIB_PATH):
        raise WcHipError(
            f"{LIB_PATH} is missing: build it with `python -m wc_gan_amd.build` "
            "(or __graft_entry__.build()); the WC path has no non-HIP fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    if lib.wc_abi_version() != ABI_VERSION:
        raise WcHipError(f"libwc_hip.so ABI version {lib.wc_abi_version()} != {ABI_VERSION}: rebuild it "
                         "(python -m wc_gan_amd.build)")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code == WC_OK:
        return
    lib = load()
    msg = lib.wc_error_string(code)
    raise WcHipError(f"{what} failed: {ERRORS.get(code, code)} ({msg.decode() if msg else '?'})")
