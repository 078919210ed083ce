"""Tensor-level wrappers over the C ABI (include/wc_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current HIP stream and
torch.distributed.  Every function checks layouts, allocates outputs/workspace with torch.empty and
enqueues the HIP stage on torch's current stream.  The six stages mirror the reference's op groups
(SURVEY.md rows a2/a6/a7/a10):

    stats  -> factor -> color -> apply                       (forward)
    bwd_reduce -> bwd_factor -> bwd_apply                    (backward)
"""
from __future__ import annotations

import ctypes

import os

import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    # the raw stream handle straight from the C side: torch.cuda.current_stream() builds a Stream object (~10 us, and
    # this is called for every launch)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _need(t, dtype, name, ndim=None):
    if not t.is_cuda:
        raise _lib.WcHipError(f"{name} must be a CUDA/HIP tensor (the WC path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-D, got shape {tuple(t.shape)}")


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Measurement hook (bench.py, tests): while TRACE is a list every K3 entry point appends (entry point, kernel it launches, a closure
# that launches the same call again) -- so that a bench can time THE kernel a layer ran instead of one it picked itself.
TRACE = None


def _tf(b):
    return "true" if b else "false"


def _call(fn, args, what, entry=None, kernel=None, keep=()):
    """fn(*args, stream) through the error check; under TRACE also records a closure that repeats exactly this launch -- the raw
    foreign call on the then-current stream, nothing else: a timing loop over it is not paced by Python (the wrappers' checks and
    allocations cost 30-60 us per call, more than a 50-us kernel)."""
    if TRACE is not None and entry is not None:
        TRACE.append((entry, kernel, lambda: fn(*args, _stream()), keep))     # keep: the tensors behind the raw pointers
    _lib.check(fn(*args, _stream()), what)


def stats(x2d, groups=1, flat=False):
    """K1: x2d (M, C) float32 -> (sum (C,) f64, xtx (C, C) f64), the raw additive moments.
    groups > 1: M/groups consecutive rows per statistic group -> sum (G, C), xtx (G, C, C).
    flat=True (groups == 1): both are views of ONE buffer, returned third -- what sync-WC all-reduces in a single call."""
    lib = _lib.load()
    _need(x2d, torch.float32, "x", 2)
    M, C = x2d.shape
    lead = (groups,) if groups > 1 else ()
    buf = None
    if flat and groups == 1:
        buf = torch.empty(C + C * C, dtype=torch.float64, device=x2d.device)
        s, xtx = buf[:C], buf[C:].view(C, C)
    else:
        s = torch.empty(*lead, C, dtype=torch.float64, device=x2d.device)
        xtx = torch.empty(*lead, C, C, dtype=torch.float64, device=x2d.device)
    nb = lib.wc_stats_workspace_bytes(M, C, groups)
    if nb == 0:
        _lib.check(-3 if M % groups == 0 else -2, "wc_stats_f32")
    ws = _workspace(nb, x2d.device)
    _lib.check(lib.wc_stats_f32(_ptr(x2d), M, C, groups, _ptr(s), _ptr(xtx), _ptr(ws), ws.numel(), _stream()), "wc_stats_f32")
    return (s, xtx, buf) if buf is not None else (s, xtx)


def factor(s, xtx, M, C, eps, momentum, ddof, training, moving_mean, moving_cov, device, want_scale=False, groups=1):
    """K2: -> (mu (C,) f32, L (C,C) f64, W (C,C) f64); updates the moving statistics in place when training.
    With want_scale=True also returns chan_scale (C,) f32, the power-of-two 1/sigma the fp16 fast path uses."""
    lib = _lib.load()
    lead = (groups,) if groups > 1 else ()
    mu = torch.empty(*lead, C, dtype=torch.float32, device=device)
    chan_scale = torch.empty(C, dtype=torch.float32, device=device) if want_scale else None
    L = torch.empty(*lead, C, C, dtype=torch.float64, device=device)
    W = torch.empty(*lead, C, C, dtype=torch.float64, device=device)
    if moving_mean is not None:
        _need(moving_mean, torch.float32, "moving_mean")
        _need(moving_cov, torch.float32, "moving_cov", 2)
    ws = _workspace(lib.wc_factor_workspace_bytes(C, groups), device)
    _lib.check(lib.wc_factor_f64(_ptr(s), _ptr(xtx), int(M), C, groups, float(eps), float(momentum), int(ddof), int(bool(training)),
                                 _ptr(moving_mean), _ptr(moving_cov), _ptr(mu), _ptr(chan_scale), _ptr(L), _ptr(W),
                                 _ptr(ws), ws.numel(), _stream()), "wc_factor_f64")
    if CHECK_K2:
        _check_k2(ws, lib.wc_factor_error_offset(C, groups), groups, "wc_factor_f64")
    if want_scale:
        return mu, L, W, chan_scale
    return mu, L, W


def whiten(x2d, eps, momentum, ddof, moving_mean, moving_cov, groups=1):
    """K1 + K2 as one call (wc_whiten_f32; training mode, per-replica statistics): x2d (M, C) float32 -> (mu, L, W, chan_scale)
    exactly as factor(*stats(x2d, groups), M / groups, ..., training=True, want_scale=True, groups=groups) returns them, with one
    launch less and without the moments leaving the workspace.  Updates the moving statistics in place when given."""
    lib = _lib.load()
    _need(x2d, torch.float32, "x", 2)
    M, C = x2d.shape
    dev = x2d.device
    lead = (groups,) if groups > 1 else ()
    mu = torch.empty(*lead, C, dtype=torch.float32, device=dev)
    chan_scale = torch.empty(C, dtype=torch.float32, device=dev)
    L = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    W = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    if moving_mean is not None:
        _need(moving_mean, torch.float32, "moving_mean")
        _need(moving_cov, torch.float32, "moving_cov", 2)
    nb = lib.wc_whiten_workspace_bytes(M, C, groups)
    if nb == 0:
        _lib.check(-3 if M % groups == 0 else -2, "wc_whiten_f32")
    ws = _workspace(nb, dev)
    _lib.check(lib.wc_whiten_f32(_ptr(x2d), M, C, groups, float(eps), float(momentum), int(ddof), _ptr(moving_mean), _ptr(moving_cov),
                                 _ptr(mu), _ptr(chan_scale), _ptr(L), _ptr(W), _ptr(ws), ws.numel(), _stream()), "wc_whiten_f32")
    if CHECK_K2:
        _check_k2(ws, lib.wc_whiten_error_offset(M, C, groups), groups, "wc_whiten_f32")
    return mu, L, W, chan_scale


# WC_CHECK_K2=1: read K2's error words back after every call (a host synchronisation per site: for shared / time-sliced GPUs and for
# debugging -- not under graph capture).  The one-launch K2 waits, with a bounded spin, for a workgroup of its own launch.
CHECK_K2 = os.environ.get("WC_CHECK_K2", "0") == "1"
SPLIT_FLAG_WORDS = 64 + 2 * 1024          # WC_SPLIT_FLAG_WORDS of include/wc_hip.h: the planes producer's status word + its rescaling scratch


def sample_rows(M):
    """The rows of the <= 256-row subsample every fp16 path takes its centre and scales from (wc_sample_row, csrc/wc_common.h)."""
    n = min(int(M), 256)
    stride = int(M) // n
    return [r * stride + (((r * 0x9E3779B1) & 0xFFFFFFFF) >> 8) % stride for r in range(n)]


def _check_k2(ws, offset, groups, what):
    if offset == 0:
        return
    words = ws[offset:offset + 64 * groups].view(torch.int32)[::16]
    if bool((words != 0).any()):
        raise _lib.WcHipError(f"{what}: the inverse's wait for the factorisation ran out (W holds a NaN); "
                              "rerun with WC_K2_TWO_LAUNCH=1 on a shared GPU")


def color(W, gamma, chan_scale=None, groups=1, per_group=False):
    """A_k = W^T Gamma_k and At_k = A_k^T.  gamma (Kc, C, C) float32 or None (whitening only).
    With chan_scale also returns the apply plan (opaque uint8 tensor) -> (A, At, plan).
    per_group: gamma is (groups*Kc, C, C) and group g takes its own run of Kc tables (per-sample tables of a grouped batch)."""
    lib = _lib.load()
    C = W.shape[-1]
    Kc = 1 if gamma is None else gamma.shape[0]
    if gamma is not None:
        _need(gamma, torch.float32, "gamma", 3)
    if per_group:
        if gamma is None or Kc % groups != 0:
            raise ValueError("per_group needs groups*Kc coloring tables")
        Kc //= groups
    A = torch.empty(groups * Kc, C, C, dtype=torch.float32, device=W.device)      # index g*Kc + k
    At = torch.empty(groups * Kc, C, C, dtype=torch.float32, device=W.device)
    ws = _workspace(lib.wc_color_workspace_bytes(C, Kc), W.device)
    plan = None
    if chan_scale is not None and C in (32, 64, 128, 256):
        plan = _workspace(lib.wc_apply_plan_bytes(C, groups * Kc), W.device)
    _lib.check(lib.wc_color_f32(_ptr(W), _ptr(gamma), Kc, C, groups, int(bool(per_group)), _ptr(A), _ptr(At), _ptr(chan_scale), _ptr(plan),
                                _ptr(ws), ws.numel(), _stream()), "wc_color_f32")
    if chan_scale is not None:
        return A, At, plan
    return A, At


def color_split(W, gamma, xs, mu, beta):
    """color() for a site whose input is the SplitTensor xs (one statistic group): -> (A, At, plan, bias_eff) with the tables built for
    the planes' scales and bias_eff = beta + (xs.center - mu) A from the same launch (wc_color_split_f32): apply_split(xs, None, A,
    bias_eff, ..., plan=plan, folded=True) is then K3's single launch."""
    lib = _lib.load()
    C = W.shape[-1]
    Kc = 1 if gamma is None else gamma.shape[0]
    if gamma is not None:
        _need(gamma, torch.float32, "gamma", 3)
    if beta is not None:
        _need(beta, torch.float32, "beta", 2)
    dev = W.device
    A = torch.empty(Kc, C, C, dtype=torch.float32, device=dev)
    At = torch.empty(Kc, C, C, dtype=torch.float32, device=dev)
    plan = _workspace(lib.wc_apply_plan_bytes(C, Kc), dev)
    be = torch.empty(Kc, C, dtype=torch.float32, device=dev)
    _lib.check(lib.wc_color_split_f32(_ptr(W), _ptr(gamma), Kc, C, _ptr(A), _ptr(At), _ptr(xs.scale), _ptr(xs.center), _ptr(mu), _ptr(beta),
                                      _ptr(plan), _ptr(be), None, 0, _stream()), "wc_color_split_f32")
    return A, At, plan, be


def factor_mix_supported(E, C):
    return bool(_lib.load().wc_factor_mix_supported(int(E), int(C)))


def factor_mix(dictionary, alpha, idx=None, base=None):
    """Soft-assignment coloring tables (wc_factor_mix_f32): out[t] = base + sum_e alpha[idx[t], e] dictionary[e].
    dictionary (E, C, C), alpha (K, E), idx (Kc,) int32 or None (one table per class), base (C, C) or None -> (Kc, C, C)."""
    lib = _lib.load()
    _need(dictionary, torch.float32, "dictionary", 3); _need(alpha, torch.float32, "alpha", 2)
    E, C = dictionary.shape[0], dictionary.shape[-1]
    K = alpha.shape[0]
    if idx is not None:
        _need(idx, torch.int32, "idx", 1)
    if base is not None:
        _need(base, torch.float32, "base", 2)
    Kc = K if idx is None else idx.numel()
    out = torch.empty(Kc, C, C, dtype=torch.float32, device=dictionary.device)
    _lib.check(lib.wc_factor_mix_f32(_ptr(dictionary), _ptr(alpha), _ptr(idx), _ptr(base), E, C, K, Kc, _ptr(out), _stream()), "wc_factor_mix_f32")
    return out


def factor_mix_bwd(dictionary, alpha, idx, dout, want_dict=True, want_alpha=True, want_base=False):
    """Gradients of factor_mix from dout (Kc, C, C) -> (ddictionary, dalpha, dbase), None where not wanted."""
    lib = _lib.load()
    _need(dout, torch.float32, "dout", 3)
    E, C = dictionary.shape[0], dictionary.shape[-1]
    K = alpha.shape[0]
    Kc = dout.shape[0]
    dev = dout.device
    dd = torch.empty_like(dictionary) if want_dict else None
    da = torch.empty_like(alpha) if want_alpha else None
    db = torch.empty(C, C, dtype=torch.float32, device=dev) if want_base else None
    ws = _workspace(lib.wc_factor_mix_bwd_workspace_bytes(E, Kc), dev) if want_alpha else None
    _lib.check(lib.wc_factor_mix_bwd_f32(_ptr(dictionary), _ptr(alpha), _ptr(idx), _ptr(dout), E, C, K, Kc, _ptr(dd), _ptr(da), _ptr(db),
                                         _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "wc_factor_mix_bwd_f32")
    return dd, da, db


def group_bias_centered(mu, A, beta, center, groups, Kc, per_group=False):
    """group_bias() with the common centre given (a SplitTensor's centre): -> bias (groups*Kc, C), the grouped planes route's additive
    term beta - (mu_g - center) A directly."""
    lib = _lib.load()
    C = mu.shape[-1]
    bias = torch.empty(groups * Kc, C, dtype=torch.float32, device=mu.device)
    _lib.check(lib.wc_group_bias_centered_f32(_ptr(mu), _ptr(A), _ptr(beta), _ptr(center), groups, Kc, C, int(bool(per_group)), _ptr(bias),
                                              _stream()), "wc_group_bias_centered_f32")
    return bias


def group_bias(mu, A, beta, groups, Kc, per_group=False):
    """Grouped forward glue -> (center (C,), bias (groups*Kc, C)); see wc_group_bias_f32."""
    lib = _lib.load()
    C = mu.shape[-1]
    center = torch.empty(C, dtype=torch.float32, device=mu.device)
    bias = torch.empty(groups * Kc, C, dtype=torch.float32, device=mu.device)
    _lib.check(lib.wc_group_bias_f32(_ptr(mu), _ptr(A), _ptr(beta), groups, Kc, C, int(bool(per_group)), _ptr(center), _ptr(bias), _stream()),
               "wc_group_bias_f32")
    return center, bias


def apply(x, mu, A, bias, slot, out=None, fast=True, plan=None, relu=False, want_mask=False, _mask_out=None):
    """K3: y[n] = (x[n] - mu) A[slot[n]] + bias[slot[n]];  x is (N, ..., C) with C contiguous.
    relu=True folds the ReLU that follows the site into the epilogue (wc_apply_act_f32).
    want_mask=True (with relu, N*HW % 32 == 0): -> (y, mask), mask the ReLU's one-bit gradient mask (int32 (M/32, C),
    wc_apply_mask_f32) for bwd_reduce(relu_mask=...)."""
    lib = _lib.load()
    _need(x, torch.float32, "x")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    Kc = A.shape[0]
    if bias is not None:
        _need(bias, torch.float32, "bias", 2)
    if slot is not None:
        _need(slot, torch.int32, "slot", 1)
    y = torch.empty_like(x) if out is None else out
    ws = _workspace(lib.wc_apply_workspace_bytes(N, HW, C, Kc), x.device) if (fast and plan is None) else None
    traced = fast and plan is not None
    kern = f"affine_ring_kernel<{C}, {_tf(slot is not None)}, {_tf(want_mask)}, false>" if (traced and TRACE is not None) else None
    keep = (x, mu, A, bias, slot, y, plan, ws)
    if want_mask:
        if not relu or (N * HW) % 32 != 0:
            raise ValueError("want_mask needs relu=True and a row count that is a multiple of 32")
        mask = torch.empty((N * HW) // 32, C, dtype=torch.int32, device=x.device) if _mask_out is None else _mask_out
        _call(lib.wc_apply_mask_f32, (_ptr(x), _ptr(mu), _ptr(A), _ptr(bias), _ptr(slot), N, HW, C, Kc, _ptr(y), _ptr(mask),
                                      _ptr(plan) if fast else None, _ptr(ws), ws.numel() if ws is not None else 0),
              "wc_apply_mask_f32", "wc_apply_mask_f32" if traced else None, kern, keep + (mask,))
        return y, mask
    _call(lib.wc_apply_act_f32, (_ptr(x), _ptr(mu), _ptr(A), _ptr(bias), _ptr(slot), N, HW, C, Kc, 1 if relu else 0,
                                 _ptr(y), _ptr(plan) if fast else None, _ptr(ws), ws.numel() if ws is not None else 0),
          "wc_apply_act_f32", "wc_apply_act_f32" if traced else None, kern, keep)
    return y


def apply_planes_supported(shape):
    """Can K3 leave the output of a site of this NHWC shape as the next convolution's planes (apply_planes)?"""
    N, C = shape[0], shape[-1]
    HW = 1
    for d in shape[1:-1]:
        HW *= d
    return bool(_lib.load().wc_apply_planes_supported(N, HW, C))


def out_scale(gamma, beta, C, device):
    """-> the (2 + 1024,) float32 scale record of apply_planes with the predicted scale in [1] (wc_out_scale_f32: from the
    coloring parameters alone -- gamma (K, C, C) | None, beta (K, C) | None -- no pass over data)."""
    lib = _lib.load()
    rec = torch.empty(lib.wc_apply_planes_scale_floats(), dtype=torch.float32, device=device)
    K = gamma.shape[0] if gamma is not None else (beta.shape[0] if beta is not None else 1)
    if gamma is not None:
        _need(gamma, torch.float32, "gamma", 3)
    if beta is not None:
        _need(beta, torch.float32, "beta", 2)
        if beta.shape[0] != K:
            raise ValueError("gamma and beta disagree about the number of tables")
    _lib.check(lib.wc_out_scale_f32(_ptr(gamma), _ptr(beta), K, C, _ptr(rec), _stream()), "wc_out_scale_f32")
    return rec


def apply_planes(x, mu, A, bias, slot, plan, oscale, relu=True, want_mask=False, _mask_out=None, _planes_out=None):
    """K3 whose output leaves as the next convolution's operand (wc_apply_planes_f32): -> (planes (2, *x.shape) float16 = hi | lo,
    oscale) [, mask]; y ~= (hi + lo) / oscale[0].  oscale: the record out_scale() made."""
    lib = _lib.load()
    _need(x, torch.float32, "x")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    Kc = A.shape[0]
    if bias is not None:
        _need(bias, torch.float32, "bias", 2)
    if slot is not None:
        _need(slot, torch.int32, "slot", 1)
    planes = torch.empty((2,) + tuple(x.shape), dtype=torch.float16, device=x.device) if _planes_out is None else _planes_out
    mask = (torch.empty((N * HW) // 32, C, dtype=torch.int32, device=x.device) if _mask_out is None else _mask_out) if want_mask else None
    _call(lib.wc_apply_planes_f32, (_ptr(x), _ptr(mu), _ptr(A), _ptr(bias), _ptr(slot), N, HW, C, Kc, 1 if relu else 0,
                                    _ptr(planes), _ptr(oscale), _ptr(mask), _ptr(plan)), "wc_apply_planes_f32", "wc_apply_planes_f32",
          None if TRACE is None else f"affine_ring_kernel<{C}, {_tf(slot is not None)}, {_tf(want_mask)}, true>",
          (x, mu, A, bias, slot, planes, oscale, mask, plan))
    return (planes, oscale, mask) if want_mask else (planes, oscale)


class SplitTensor:
    """An activation in the pre-split format of include/wc_hip.h (ABI 4): `planes` (2, M, C) float16 = hi | lo,
    x ~= center + (hi + lo) / scale.  `shape` is the NHWC shape of the tensor it stands for; `flag` int32: [0] != 0 after split()
    had to clamp (scales off by more than three decades) -- resadd_split() never clamps: there [0] != 0 says that its gated second
    pass re-ran with the true maxima of the channels whose sampled scale was too tight, and `scale` holds the scales it used."""

    __slots__ = ("planes", "center", "scale", "flag", "shape", "x32", "moments")

    def __init__(self, planes, center, scale, flag, shape, x32=None, moments=None):
        self.planes, self.center, self.scale, self.flag, self.shape = planes, center, scale, flag, tuple(shape)
        self.x32 = x32          # the same tensor in fp32, where the producer also wrote it (a reader without a planes path)
        # (workspace, groups) where the producer also accumulated the covariance partials of the tensor in its own pass
        # (resadd_stats_split): whiten_presummed / stats_presummed finish K1 from there, no pass over the planes
        self.moments = moments

    @property
    def C(self):
        return self.shape[-1]

    @property
    def M(self):
        return self.planes.shape[1]


def split_scales(x):
    """-> (center (C,), scale (C,), flag (64,) int32 zeroed): sampled per-channel centre and power-of-two scale of x (..., C)."""
    lib = _lib.load()
    _need(x, torch.float32, "x")
    C = x.shape[-1]
    M = x.numel() // C
    center = torch.empty(C, dtype=torch.float32, device=x.device)
    scale = torch.empty(C, dtype=torch.float32, device=x.device)
    flag = torch.empty(64, dtype=torch.int32, device=x.device)
    _lib.check(lib.wc_split_scales_f32(_ptr(x), M, C, _ptr(center), _ptr(scale), _ptr(flag), _stream()), "wc_split_scales_f32")
    return center, scale, flag


def split(x, center=None, scale=None, flag=None, relu=False):
    """x (N, ..., C) float32 -> SplitTensor (scales sampled from x when not given)."""
    lib = _lib.load()
    _need(x, torch.float32, "x")
    C = x.shape[-1]
    M = x.numel() // C
    if scale is None:
        center, scale, flag = split_scales(x)
    planes = torch.empty(2, M, C, dtype=torch.float16, device=x.device)
    _lib.check(lib.wc_split_f32(_ptr(x), _ptr(center), _ptr(scale), M, C, 1 if relu else 0, _ptr(planes), _ptr(flag), _stream()),
               "wc_split_f32")
    return SplitTensor(planes, center, scale, flag, x.shape)


def unsplit(xs):
    """SplitTensor -> float32 tensor of xs.shape."""
    lib = _lib.load()
    x = torch.empty(xs.shape, dtype=torch.float32, device=xs.planes.device)
    _lib.check(lib.wc_unsplit_f32(_ptr(xs.planes), _ptr(xs.center), _ptr(xs.scale), xs.M, xs.C, _ptr(x), _stream()), "wc_unsplit_f32")
    return x


def apply_split_supported(shape):
    lib = _lib.load()
    N, C = shape[0], shape[-1]
    HW = 1
    for d in shape[1:-1]:
        HW *= d
    return bool(lib.wc_apply_split_supported(N, HW, C))


def stats_split_supported(M, C, groups=1):
    return bool(_lib.load().wc_stats_split_supported(int(M), int(C), int(groups)))


def stats_split(xs, groups=1, flat=False):
    """K1 on a SplitTensor: as stats(x2d, groups, flat) for the tensor the planes stand for."""
    lib = _lib.load()
    M, C = xs.M, xs.C
    dev = xs.planes.device
    lead = (groups,) if groups > 1 else ()
    buf = None
    if flat and groups == 1:
        buf = torch.empty(C + C * C, dtype=torch.float64, device=dev)
        s, xtx = buf[:C], buf[C:].view(C, C)
    else:
        s = torch.empty(*lead, C, dtype=torch.float64, device=dev)
        xtx = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    nb = lib.wc_stats_split_workspace_bytes(M, C, groups)
    if nb == 0:
        _lib.check(-2, "wc_stats_split_f16x2")
    ws = _workspace(nb, dev)
    _lib.check(lib.wc_stats_split_f16x2(_ptr(xs.planes), _ptr(xs.center), _ptr(xs.scale), M, C, groups, _ptr(s), _ptr(xtx),
                                        _ptr(ws), ws.numel(), _stream()), "wc_stats_split_f16x2")
    return (s, xtx, buf) if buf is not None else (s, xtx)


def split_bias(A, bias, xs, mu):
    """bias_eff (Kc, C) = bias + (xs.center - mu) A: the split apply's additive term, folded once per forward."""
    lib = _lib.load()
    Kc, C = A.shape[0], A.shape[-1]
    out = torch.empty(Kc, C, dtype=torch.float32, device=A.device)
    _lib.check(lib.wc_split_bias_f32(_ptr(A), _ptr(bias), _ptr(xs.center), _ptr(mu), Kc, C, _ptr(out), _stream()), "wc_split_bias_f32")
    return out


def apply_split_workspace(C, Kc, device):
    """The scratch apply_split needs (pass it as ws= to keep a hot loop free of allocations)."""
    return _workspace(_lib.load().wc_apply_split_workspace_bytes(C, Kc), device)


def apply_split(xs, mu, A, bias, slot, plan=None, relu=False, out=None, folded=False, ws=None, want_mask=False, oscale=None,
                _mask_out=None, _planes_out=None):
    """K3 on a pre-split input: y[n] = (x[n] - mu) A[slot[n]] + bias[slot[n]].  `plan` must come from color(W, gamma,
    chan_scale=xs.scale) (None: the tables are built inside the call).  folded=True: `bias` is split_bias(...)'s result
    (mu is ignored) and the call is a single launch.
    want_mask (relu, rows % 32 == 0): also the ReLU's one-bit gradient mask -> (y, mask).
    oscale (the record out_scale() made): the output leaves as the next convolution's fp16 planes instead of y (wc_apply_split_ex_f16x2;
    the protocol of apply_planes) -> (planes (2, *shape) float16, oscale[, mask])."""
    lib = _lib.load()
    N, C = xs.shape[0], xs.shape[-1]
    HW = xs.M // N
    Kc = A.shape[0]
    dev = xs.planes.device
    if bias is not None:
        _need(bias, torch.float32, "bias", 2)
    if slot is not None:
        _need(slot, torch.int32, "slot", 1)
    if ws is None:
        ws = _workspace(lib.wc_apply_split_workspace_bytes(C, Kc), dev)
    mask = None
    if want_mask:
        if not relu or xs.M % 32 != 0:
            raise ValueError("want_mask needs relu=True and a row count that is a multiple of 32")
        mask = torch.empty(xs.M // 32, C, dtype=torch.int32, device=dev) if _mask_out is None else _mask_out
    planes = y = None
    if oscale is not None:
        planes = torch.empty((2,) + tuple(xs.shape), dtype=torch.float16, device=dev) if _planes_out is None else _planes_out
    else:
        y = torch.empty(xs.shape, dtype=torch.float32, device=dev) if out is None else out
    _call(lib.wc_apply_split_ex_f16x2, (_ptr(xs.planes), None if folded else _ptr(xs.center), _ptr(xs.scale), None if folded else _ptr(mu),
                                        _ptr(A), _ptr(bias), _ptr(slot), N, HW, C, Kc, 1 if relu else 0, _ptr(y), _ptr(mask),
                                        _ptr(planes), _ptr(oscale), _ptr(plan), _ptr(ws), ws.numel()),
          "wc_apply_split_ex_f16x2", "wc_apply_split_ex_f16x2",
          None if TRACE is None else f"apply_split_kernel<{C}, {_tf(slot is not None)}, {_tf(want_mask)}, {_tf(oscale is not None)}>",
          (xs.planes, xs.center, xs.scale, mu, A, bias, slot, y, mask, planes, oscale, plan, ws))
    if planes is not None:
        return (planes, oscale, mask) if want_mask else (planes, oscale)
    return (y, mask) if want_mask else y


def whiten_split(xs, eps, momentum, ddof, moving_mean, moving_cov, groups=1):
    """K1 + K2 on a SplitTensor as one call (wc_whiten_split_f16x2; training mode, per-replica statistics): -> (mu, L, W) as
    factor(*stats_split(xs, groups), ...) returns them.  The apply's input scales are xs.scale (give them to color())."""
    lib = _lib.load()
    M, C = xs.M, xs.C
    dev = xs.planes.device
    lead = (groups,) if groups > 1 else ()
    mu = torch.empty(*lead, C, dtype=torch.float32, device=dev)
    L = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    W = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    if moving_mean is not None:
        _need(moving_mean, torch.float32, "moving_mean")
        _need(moving_cov, torch.float32, "moving_cov", 2)
    nb = lib.wc_whiten_split_workspace_bytes(M, C, groups)
    if nb == 0:
        _lib.check(-2, "wc_whiten_split_f16x2")
    ws = _workspace(nb, dev)
    _lib.check(lib.wc_whiten_split_f16x2(_ptr(xs.planes), _ptr(xs.center), _ptr(xs.scale), M, C, groups, float(eps), float(momentum),
                                         int(ddof), _ptr(moving_mean), _ptr(moving_cov), _ptr(mu), _ptr(L), _ptr(W), _ptr(ws), ws.numel(),
                                         _stream()), "wc_whiten_split_f16x2")
    if CHECK_K2:
        _check_k2(ws, lib.wc_whiten_split_error_offset(M, C, groups), groups, "wc_whiten_split_f16x2")
    return mu, L, W


# ---------------------------------------------------------------------------------------------
# the residual add of a generator block (generator.py:142-146) as the producer of the next site's input (csrc/wc_resadd.hip)
# ---------------------------------------------------------------------------------------------
def resadd_split_supported(shape):
    N, H, W, C = shape                      # (wc_resadd_split_supported's rule, without the call: this sits on the layers' hot path)
    return C in (128, 256) and N > 0 and H > 0 and W > 0 and N * H * W < (1 << 31)


def _resadd_args(h, s, up):
    _need(h, torch.float32, "h", 4)
    N, H, W, C = h.shape
    if s is not None:
        _need(s, torch.float32, "s", 4)
        want = (N, H // 2, W // 2, C) if up else (N, H, W, C)
        if tuple(s.shape) != want:
            raise ValueError(f"shortcut shape {tuple(s.shape)} does not fit the sum {tuple(h.shape)} (up={bool(up)})")
    return N, H, W, C


def resadd(h, s, up=False):
    """h + (upsample2x of) s as an fp32 tensor: h (N, H, W, C), s (N, H >> up, W >> up, C) or None."""
    lib = _lib.load()
    N, H, W, C = _resadd_args(h, s, up)
    out = torch.empty_like(h)
    _lib.check(lib.wc_resadd_f32(_ptr(h), _ptr(s), N, H, W, C, 1 if up else 0, _ptr(out), _stream()), "wc_resadd_f32")
    return out


def resadd_split(h, s, up=False, want_x32=False):
    """The same sum written in the pre-split format (one sampling launch + one pass over h and s + the gate of the rescaling pass, which
    runs only when a sampled scale was too tight: nothing saturates): -> SplitTensor, with .x32 = the fp32 sum as well when want_x32
    (a reader without a planes path)."""
    lib = _lib.load()
    N, H, W, C = _resadd_args(h, s, up)
    dev = h.device
    planes = torch.empty(2, N * H * W, C, dtype=torch.float16, device=dev)
    center = torch.empty(C, dtype=torch.float32, device=dev)
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    flag = torch.empty(SPLIT_FLAG_WORDS, dtype=torch.int32, device=dev)
    x32 = torch.empty_like(h) if want_x32 else None
    _lib.check(lib.wc_resadd_split_f32(_ptr(h), _ptr(s), N, H, W, C, 1 if up else 0, _ptr(planes), _ptr(center), _ptr(scale), _ptr(flag),
                                       _ptr(x32), _stream()), "wc_resadd_split_f32")
    return SplitTensor(planes, center, scale, flag, h.shape, x32)


def resadd_stats_supported(shape, up, groups=1):
    """Can the residual add of this NHWC output shape also accumulate the consuming site's covariance partials (resadd_stats_split)?"""
    N, H, W, C = shape
    return bool(_lib.load().wc_resadd_stats_supported(int(N), int(H), int(W), int(C), 1 if up else 0, int(groups)))


def resadd_stats_split(h, s, up=True, groups=1, want_x32=False):
    """resadd_split whose pass ALSO accumulates the covariance partials of the sum for the next WC site (wc_resadd_stats_split_f32:
    the producer feeds K1 literally -- that site's own K1 pass over the planes does not run): -> SplitTensor with .moments = (ws, groups);
    hand it to whiten_presummed (K1 tail + K2) or stats_presummed (the raw moments).  `groups`: the consuming site's statistic groups."""
    lib = _lib.load()
    N, H, W, C = _resadd_args(h, s, up)
    dev = h.device
    nb = lib.wc_resadd_stats_workspace_bytes(N, H, W, C, groups)
    if nb == 0 or not lib.wc_resadd_stats_supported(N, H, W, C, 1 if up else 0, groups):
        _lib.check(-2, "wc_resadd_stats_split_f32")
    planes = torch.empty(2, N * H * W, C, dtype=torch.float16, device=dev)
    center = torch.empty(C, dtype=torch.float32, device=dev)
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    flag = torch.empty(SPLIT_FLAG_WORDS, dtype=torch.int32, device=dev)
    x32 = torch.empty_like(h) if want_x32 else None
    ws = _workspace(nb, dev)
    _lib.check(lib.wc_resadd_stats_split_f32(_ptr(h), _ptr(s), N, H, W, C, 1 if up else 0, int(groups), _ptr(planes), _ptr(center), _ptr(scale),
                                             _ptr(flag), _ptr(x32), _ptr(ws), ws.numel(), _stream()), "wc_resadd_stats_split_f32")
    return SplitTensor(planes, center, scale, flag, h.shape, x32, moments=(ws, int(groups)))


def whiten_presummed(xs, eps, momentum, ddof, moving_mean, moving_cov, groups=1):
    """K1's tail + K2 for a SplitTensor whose producer left the covariance partials (xs.moments): -> (mu, L, W) as whiten_split(xs, ...)
    returns them, without the pass over the planes (wc_whiten_presummed_f16x2)."""
    lib = _lib.load()
    ws, g = xs.moments
    if g != groups:
        raise ValueError(f"the producer accumulated its partials for {g} statistic groups, the site asks for {groups}")
    M, C = xs.M, xs.C
    dev = xs.planes.device
    lead = (groups,) if groups > 1 else ()
    mu = torch.empty(*lead, C, dtype=torch.float32, device=dev)
    L = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    W = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    if moving_mean is not None:
        _need(moving_mean, torch.float32, "moving_mean")
        _need(moving_cov, torch.float32, "moving_cov", 2)
    _lib.check(lib.wc_whiten_presummed_f16x2(_ptr(xs.center), M, C, groups, float(eps), float(momentum), int(ddof), _ptr(moving_mean),
                                             _ptr(moving_cov), _ptr(mu), _ptr(L), _ptr(W), _ptr(ws), ws.numel(), _stream()),
               "wc_whiten_presummed_f16x2")
    if CHECK_K2:
        _check_k2(ws, lib.wc_whiten_presummed_error_offset(M, C, groups), groups, "wc_whiten_presummed_f16x2")
    return mu, L, W


def stats_presummed(xs, groups=1, flat=False):
    """The raw moments (sum, xtx) of a SplitTensor whose producer left the partials: as stats_split(xs, groups, flat), no pass over the planes."""
    lib = _lib.load()
    ws, g = xs.moments
    if g != groups:
        raise ValueError(f"the producer accumulated its partials for {g} statistic groups, the site asks for {groups}")
    M, C = xs.M, xs.C
    dev = xs.planes.device
    lead = (groups,) if groups > 1 else ()
    buf = None
    if flat and groups == 1:
        buf = torch.empty(C + C * C, dtype=torch.float64, device=dev)
        sm, xtx = buf[:C], buf[C:].view(C, C)
    else:
        sm = torch.empty(*lead, C, dtype=torch.float64, device=dev)
        xtx = torch.empty(*lead, C, C, dtype=torch.float64, device=dev)
    _lib.check(lib.wc_stats_presummed_f16x2(_ptr(xs.center), M, C, groups, _ptr(sm), _ptr(xtx), _ptr(ws), ws.numel(), _stream()),
               "wc_stats_presummed_f16x2")
    return (sm, xtx, buf) if buf is not None else (sm, xtx)


def patch_sum(g):
    """(N, 2 Hs, 2 Ws, C) -> (N, Hs, Ws, C): every source pixel collects its 2x2 patch (the gradient of resadd(up=True) w.r.t. s)."""
    lib = _lib.load()
    _need(g, torch.float32, "g", 4)
    N, H, W, C = g.shape
    out = torch.empty(N, H // 2, W // 2, C, dtype=torch.float32, device=g.device)
    _lib.check(lib.wc_patch_sum_f32(_ptr(g), N, H // 2, W // 2, C, _ptr(out), _stream()), "wc_patch_sum_f32")
    return out


def _oc_strides(w):
    # a 1x1 kernel (Cout, Cin, 1, 1) in any dense layout: element (o, c) at o * so + c * sc
    return w.stride(0), w.stride(1)


def fold_channel_scale(w, bias, scale, center):
    """(wf, bf): the 1x1 convolution weight / bias that act on a SplitTensor's planes as (w, bias) act on the tensor itself
    (wc_fold_channel_scale_f32): wf[o, c] = w[o, c] / scale[c], bf[o] = bias[o] + <center, w[o]>."""
    lib = _lib.load()
    so, sc = _oc_strides(w)
    wf = torch.empty_like(w)
    bf = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
    _lib.check(lib.wc_fold_channel_scale_f32(_ptr(w), so, sc, w.shape[0], w.shape[1], _ptr(bias), _ptr(scale), _ptr(center), _ptr(wf),
                                             _ptr(bf), _stream()), "wc_fold_channel_scale_f32")
    return wf, bf


def unfold_channel_scale(D, db, scale, center):
    """dW[o, c] = D[o, c] / scale[c] + center[c] db[o]: the weight gradient of the convolution on the tensor from the one on its planes."""
    lib = _lib.load()
    so, sc = _oc_strides(D)
    dW = torch.empty_like(D)
    _lib.check(lib.wc_unfold_channel_scale_f32(_ptr(D), _ptr(db), so, sc, D.shape[0], D.shape[1], _ptr(scale), _ptr(center), _ptr(dW),
                                               _stream()), "wc_unfold_channel_scale_f32")
    return dW


def bwd_bits_supported(shape, has_slot):
    """Can the ReLU'd backward of a site of this NHWC shape run without a masked copy of the gradient (bwd_reduce(..., relu_mask=,
    write_masked=False) + bwd_apply(..., relu_mask=))?"""
    N, C = shape[0], shape[-1]
    HW = 1
    for d in shape[1:-1]:
        HW *= d
    return bool(_lib.load().wc_bwd_bits_supported(N, HW, C, int(bool(has_slot))))


def bwd_reduce(x, mu, gy, slot, Kc, flat=False, want_scales=False, relu_y=None, relu_mask=None, write_masked=True):
    """K4: -> (R (Kc,C,C) f64, gsum (Kc,C) f64); flat=True: views of one buffer, returned third (sync-WC's single all-reduce);
    want_scales=True: the (2C,) per-channel input scales of (x - mu) and gy, returned last, for bwd_apply(scales=...);
    relu_y: the site's output y when its ReLU rode in K3 -- gy is masked (gy where y > 0) while it is staged, and the masked
    gradient is returned in front of the scales (the gy that bwd_apply then takes)."""
    lib = _lib.load()
    _need(x, torch.float32, "x")
    _need(gy, torch.float32, "gy")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    buf = None
    if flat:
        buf = torch.empty(Kc * (C * C + C), dtype=torch.float64, device=x.device)
        R, gsum = buf[:Kc * C * C].view(Kc, C, C), buf[Kc * C * C:].view(Kc, C)
    else:
        R = torch.empty(Kc, C, C, dtype=torch.float64, device=x.device)
        gsum = torch.empty(Kc, C, dtype=torch.float64, device=x.device)
    ws = _workspace(lib.wc_bwd_reduce_workspace_bytes(N, HW, C, Kc, int(slot is not None)), x.device)
    scales = torch.empty(2 * C, dtype=torch.float32, device=x.device) if want_scales else None
    gm = None
    if relu_mask is not None and not write_masked:
        # K4 with the bits and NO masked copy (wc_bwd_reduce_bits_f32): -> (R, gsum[, buf], scales); bwd_apply(relu_mask=) masks for itself
        _need(relu_mask, torch.int32, "relu_mask", 2)
        if scales is None:
            scales = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        _lib.check(lib.wc_bwd_reduce_bits_f32(_ptr(x), _ptr(mu), _ptr(gy), _ptr(relu_mask), _ptr(slot), N, HW, C, Kc, _ptr(R), _ptr(gsum),
                                              _ptr(scales), _ptr(ws), ws.numel(), _stream()), "wc_bwd_reduce_bits_f32")
        out = (R, gsum, buf) if buf is not None else (R, gsum)
        return out + (scales,)
    if relu_mask is not None:         # the mask in apply(..., want_mask=True)'s one-bit form: same outputs as relu_y
        _need(relu_mask, torch.int32, "relu_mask", 2)
        gm = torch.empty_like(gy)
        _lib.check(lib.wc_bwd_reduce_mask_f32(_ptr(x), _ptr(mu), _ptr(gy), _ptr(relu_mask), _ptr(slot), N, HW, C, Kc, _ptr(R), _ptr(gsum),
                                              _ptr(gm), _ptr(scales), _ptr(ws), ws.numel(), _stream()), "wc_bwd_reduce_mask_f32")
        out = (R, gsum, buf) if buf is not None else (R, gsum)
        out = out + (gm,)
        return out + (scales,) if want_scales else out
    if relu_y is not None:
        _need(relu_y, torch.float32, "relu_y")
        gm = torch.empty_like(gy)
    _lib.check(lib.wc_bwd_reduce_relu_f32(_ptr(x), _ptr(mu), _ptr(gy), _ptr(relu_y), _ptr(slot), N, HW, C, Kc, _ptr(R), _ptr(gsum),
                                          _ptr(gm), _ptr(scales), _ptr(ws), ws.numel(), _stream()), "wc_bwd_reduce_relu_f32")
    out = (R, gsum, buf) if buf is not None else (R, gsum)
    if gm is not None:
        out = out + (gm,)
    return out + (scales,) if want_scales else out


def bwd_xsplit_supported(shape, has_slot):
    """Can K4 and K6 of a site of this NHWC shape read x from pre-split planes (bwd_reduce_xsplit / bwd_apply_xsplit: C = 256, the fast
    reduction and the one-pass K6; C = 128: the plain reduction and K6 as planes pass + accumulating pass)?  Then no fp32 copy of x has to exist for the backward."""
    N, C = shape[0], shape[-1]
    HW = 1
    for d in shape[1:-1]:
        HW *= d
    return bool(_lib.load().wc_bwd_xsplit_supported(N, HW, C, int(bool(has_slot))))


def bwd_reduce_xsplit(xs, mu, gy, slot, Kc, relu_mask=None, flat=False):
    """K4 with x as a SplitTensor (wc_bwd_reduce_xsplit_f32): -> (R, gsum[, buf], scales); relu_mask: the site's one-bit ReLU mask, applied
    while gy is staged (gy is then the gradient before the ReLU; bwd_apply_xsplit(relu_mask=) masks for itself)."""
    lib = _lib.load()
    _need(gy, torch.float32, "gy")
    N, C = gy.shape[0], gy.shape[-1]
    HW = gy.numel() // (N * C)
    dev = gy.device
    buf = None
    if flat:
        buf = torch.empty(Kc * (C * C + C), dtype=torch.float64, device=dev)
        R, gsum = buf[:Kc * C * C].view(Kc, C, C), buf[Kc * C * C:].view(Kc, C)
    else:
        R = torch.empty(Kc, C, C, dtype=torch.float64, device=dev)
        gsum = torch.empty(Kc, C, dtype=torch.float64, device=dev)
    if relu_mask is not None:
        _need(relu_mask, torch.int32, "relu_mask", 2)
    ws = _workspace(lib.wc_bwd_reduce_workspace_bytes(N, HW, C, Kc, int(slot is not None)), dev)
    scales = torch.empty(2 * C, dtype=torch.float32, device=dev)
    _lib.check(lib.wc_bwd_reduce_xsplit_f32(_ptr(xs.planes), _ptr(xs.center), _ptr(xs.scale), _ptr(mu), _ptr(gy), _ptr(relu_mask), _ptr(slot),
                                            N, HW, C, Kc, _ptr(R), _ptr(gsum), _ptr(scales), _ptr(ws), ws.numel(), _stream()),
               "wc_bwd_reduce_xsplit_f32")
    return (R, gsum, buf, scales) if buf is not None else (R, gsum, scales)


def bwd_apply_xsplit(gy, xs, mu, At, S, gmean, slot, scales, relu_mask=None):
    """K6 with x as a SplitTensor (wc_bwd_apply_xsplit_f32): dx[n] = gy[n] At[slot[n]] + (x[n] - mu) S - gmean; scales from
    bwd_reduce_xsplit of the same site."""
    lib = _lib.load()
    _need(gy, torch.float32, "gy")
    N, C = gy.shape[0], gy.shape[-1]
    HW = gy.numel() // (N * C)
    Kc = At.shape[0]
    dx = torch.empty_like(gy)
    ws = _workspace(lib.wc_bwd_apply_xsplit_workspace_bytes(C, Kc), gy.device)
    _lib.check(lib.wc_bwd_apply_xsplit_f32(_ptr(gy), _ptr(relu_mask), _ptr(xs.planes), _ptr(xs.center), _ptr(xs.scale), _ptr(mu), _ptr(At),
                                           _ptr(S), _ptr(gmean), _ptr(slot), N, HW, C, Kc, _ptr(scales), _ptr(dx), _ptr(ws), ws.numel(),
                                           _stream()), "wc_bwd_apply_xsplit_f32")
    return dx


def relu_mask_bits(gy, mask):
    """gy where the one-bit mask says the activation passed, else 0 (the elementwise form; K4 does the same while it stages)."""
    lib = _lib.load()
    _need(gy, torch.float32, "gy")
    C = gy.shape[-1]
    M = gy.numel() // C
    out = torch.empty_like(gy)
    # (no dedicated entry point: the masked copy is what wc_bwd_reduce_mask_f32 writes; here via the same kernel)
    _lib.check(lib.wc_relu_mask_apply_f32(_ptr(gy), _ptr(mask), M, C, _ptr(out), _stream()), "wc_relu_mask_apply_f32")
    return out


def bwd_factor(R, gsum, W, L, gamma, A, M, eps, ddof, training, want_dgamma=True, want_dbeta=True):
    """K5: -> (dgamma (Kc,C,C) f32 | None, dbeta (Kc,C) f32 | None, S (C,C) f32 | None, gmean (C,) f32 | None)."""
    lib = _lib.load()
    Kc, C = R.shape[0], R.shape[1]
    dev = R.device
    dgamma = torch.empty(Kc, C, C, dtype=torch.float32, device=dev) if (gamma is not None and want_dgamma) else None
    dbeta = torch.empty(Kc, C, dtype=torch.float32, device=dev) if want_dbeta else None
    S = torch.empty(C, C, dtype=torch.float32, device=dev) if training else None
    gmean = torch.empty(C, dtype=torch.float32, device=dev) if training else None
    ws = _workspace(lib.wc_bwd_factor_workspace_bytes(C, Kc), dev)
    _lib.check(lib.wc_bwd_factor_f64(_ptr(R), _ptr(gsum), _ptr(W), _ptr(L), _ptr(gamma), _ptr(A), Kc, C, int(M),
                                     float(eps), int(ddof), int(bool(training)), _ptr(dgamma), _ptr(dbeta),
                                     _ptr(S), _ptr(gmean), _ptr(ws), ws.numel(), _stream()), "wc_bwd_factor_f64")
    return dgamma, dbeta, S, gmean


def bwd_apply(gy, x, mu, At, S, gmean, slot, fast=True, scales=None, relu_mask=None):
    """K6: dx[n] = gy[n] At[slot[n]] + (x[n]-mu) S - gmean.  scales: the (2C,) input scales bwd_reduce(..., want_scales=True)
    returned for the same x, mu, gy (three launches instead of six).  relu_mask: gy is the gradient BEFORE the site's ReLU and
    this its one-bit mask -- applied while gy is converted (wc_bwd_apply_bits_f32; bwd_bits_supported shapes, scales required)."""
    lib = _lib.load()
    _need(gy, torch.float32, "gy")
    N, C = gy.shape[0], gy.shape[-1]
    HW = gy.numel() // (N * C)
    Kc = At.shape[0]
    dx = torch.empty_like(gy)
    ws = _workspace(lib.wc_bwd_apply_workspace_bytes(N, HW, C, Kc), gy.device) if fast else None
    if relu_mask is not None:
        _need(relu_mask, torch.int32, "relu_mask", 2)
        _lib.check(lib.wc_bwd_apply_bits_f32(_ptr(gy), _ptr(relu_mask), _ptr(x), _ptr(mu), _ptr(At), _ptr(S), _ptr(gmean), _ptr(slot),
                                             N, HW, C, Kc, _ptr(scales), _ptr(dx), _ptr(ws), ws.numel() if ws is not None else 0,
                                             _stream()), "wc_bwd_apply_bits_f32")
        return dx
    _lib.check(lib.wc_bwd_apply_scaled_f32(_ptr(gy), _ptr(x), _ptr(mu), _ptr(At), _ptr(S), _ptr(gmean), _ptr(slot),
                                           N, HW, C, Kc, _ptr(scales), _ptr(dx), _ptr(ws), ws.numel() if ws is not None else 0,
                                           _stream()), "wc_bwd_apply_scaled_f32")
    return dx


def stream_copy(src, dst):
    lib = _lib.load()
    _lib.check(lib.wc_stream_copy_f32(_ptr(src), _ptr(dst), src.numel(), _stream()), "wc_stream_copy_f32")
    return dst


def spectral_norm_workspace(rows, cols, device):
    """The per-weight scratch of the spectral-norm op: zeroed once here, left zero by every launch."""
    lib = _lib.load()
    return torch.zeros(int(lib.wc_spectral_norm_workspace_bytes(int(rows), int(cols))), dtype=torch.uint8, device=device)


def spectral_norm(weight, u, v, iterations, ws, eps=1e-12, keep_uv=False):
    """N3: one launch of the power iteration + normalisation.  weight: float32, dense (contiguous in its own memory
    format; rows = shape[0]); u (rows,), v (numel/rows,) float32, updated in place when iterations > 0; ws from
    spectral_norm_workspace (one per weight).  Returns (w_sn with the weight's shape/strides, sigma (1,)) and, with
    keep_uv, fresh copies of u and v as used for sigma (written by the same launch)."""
    lib = _lib.load()
    if not weight.is_cuda:
        raise _lib.WcHipError("weight must be a CUDA/HIP tensor (the spectral-norm op has no CPU fallback)")
    if weight.dtype != torch.float32:
        raise TypeError(f"weight must be float32, got {weight.dtype}")
    if not (weight.is_contiguous() or (weight.dim() == 4 and weight.is_contiguous(memory_format=torch.channels_last))):
        raise ValueError("weight must be dense (contiguous or channels_last)")
    _need(u, torch.float32, "u", 1)
    _need(v, torch.float32, "v", 1)
    rows = weight.shape[0]
    cols = weight.numel() // rows
    if u.numel() != rows or v.numel() != cols:
        raise ValueError(f"u/v sizes {u.numel()}/{v.numel()} do not match a ({rows}, {cols}) matrix")
    w_sn = torch.empty_like(weight)            # preserve_format: same memory order as the weight
    sigma = torch.empty(1, dtype=torch.float32, device=weight.device)
    uu = torch.empty_like(u) if keep_uv else None
    vv = torch.empty_like(v) if keep_uv else None
    _lib.check(lib.wc_spectral_norm_f32(_ptr(weight), rows, cols, _ptr(u), _ptr(v), int(iterations), float(eps),
                                        _ptr(w_sn), _ptr(sigma), _ptr(uu), _ptr(vv), _ptr(ws), ws.numel(), _stream()),
               "wc_spectral_norm_f32")
    return (w_sn, sigma, uu, vv) if keep_uv else (w_sn, sigma)


def spectral_norm_bwd(g, w_sn, u, v, sigma, fully_diff, ws):
    """dW = (g - fully_diff <g, w_sn> u v^T) / sigma, in w_sn's memory order."""
    lib = _lib.load()
    rows = w_sn.shape[0]
    cols = w_sn.numel() // rows
    if g.stride() != w_sn.stride():
        g = g.contiguous(memory_format=torch.channels_last) if (w_sn.dim() == 4 and w_sn.is_contiguous(memory_format=torch.channels_last)) else g.contiguous()
    if g.dtype != torch.float32 or not g.is_cuda:
        raise TypeError("g must be a float32 CUDA/HIP tensor")
    dW = torch.empty_like(w_sn)
    _lib.check(lib.wc_spectral_norm_bwd_f32(_ptr(g), _ptr(w_sn), _ptr(u), _ptr(v), _ptr(sigma), rows, cols,
                                            1 if fully_diff else 0, _ptr(dW), _ptr(ws), ws.numel(), _stream()),
               "wc_spectral_norm_bwd_f32")
    return dW


def spectral_norm_batched(weights, us, vs, wss, iterations, eps=1e-12):
    """All layers in one launch: lists of weights / u / v / workspaces -> lists (w_sn, sigma, u_used, v_used)."""
    lib = _lib.load()
    n = len(weights)
    items = (_lib.SnItem * n)()
    outs = []
    for i, (w, u, v, ws) in enumerate(zip(weights, us, vs, wss)):
        if not w.is_cuda or w.dtype != torch.float32:
            raise _lib.WcHipError("spectral_norm_batched needs float32 CUDA/HIP weights")
        if not (w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last))):
            raise ValueError("weight must be dense (contiguous or channels_last)")
        rows = w.shape[0]; cols = w.numel() // rows
        w_sn = torch.empty_like(w); sigma = torch.empty(1, dtype=torch.float32, device=w.device)
        uu = torch.empty_like(u); vv = torch.empty_like(v)
        items[i] = _lib.SnItem(_ptr(w), _ptr(u), _ptr(v), _ptr(w_sn), _ptr(sigma), _ptr(uu), _ptr(vv), _ptr(ws), rows, cols)
        outs.append((w_sn, sigma, uu, vv))
    _lib.check(lib.wc_spectral_norm_batched_f32(ctypes.addressof(items), n, int(iterations), float(eps), _stream()),
               "wc_spectral_norm_batched_f32")
    return outs


def spectral_norm_bwd_batched(gs, w_sns, us, vs, sigmas, wss, fully_diff):
    lib = _lib.load()
    n = len(gs)
    items = (_lib.SnBwdItem * n)()
    dWs = []
    keep = []
    for i in range(n):
        g, w_sn = gs[i], w_sns[i]
        if g.stride() != w_sn.stride():
            g = g.contiguous(memory_format=torch.channels_last) if (w_sn.dim() == 4 and w_sn.is_contiguous(memory_format=torch.channels_last)) else g.contiguous()
        keep.append(g)
        rows = w_sn.shape[0]; cols = w_sn.numel() // rows
        dW = torch.empty_like(w_sn)
        items[i] = _lib.SnBwdItem(_ptr(g), _ptr(w_sn), _ptr(us[i]), _ptr(vs[i]), _ptr(sigmas[i]), _ptr(dW), _ptr(wss[i]), rows, cols)
        dWs.append(dW)
    _lib.check(lib.wc_spectral_norm_bwd_batched_f32(ctypes.addressof(items), n, 1 if fully_diff else 0, _stream()),
               "wc_spectral_norm_bwd_batched_f32")
    return dWs
