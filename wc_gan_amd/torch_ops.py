"""The WC stages as PyTorch custom operators (`torch.ops.wc.*`), the op surface SURVEY.md section 8b names.

    wc::stats  wc::factor  wc::color  wc::apply            forward  (K1, K2, color, K3)
    wc::bwd_reduce  wc::bwd_factor  wc::bwd_apply          backward (K4, K5, K6)
    wc::whiten_color                                       the fused site, with autograd registered on the op

Each is a thin `torch.library.custom_op` over the ctypes wrappers of wc_gan_amd.ops (the C ABI of include/wc_hip.h):
schemas, fake (meta) kernels for shape propagation, declared mutation of the moving statistics -- so that
`torch.compile`, `torch.library.opcheck` and FX tooling see the stages as operators instead of opaque Python.  The
layers themselves (wc_gan_amd.layers / functional) keep calling the wrappers directly: a Python custom op costs tens of
microseconds of dispatch per call and the step makes ~300 of these calls; inside a captured hipGraph neither form
costs anything.  There is no CPU kernel behind these ops: a CPU tensor raises, as everywhere in this package.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import ops

_E = lambda like, *shape, dtype=None: torch.empty(*shape, dtype=dtype or like.dtype, device=like.device)


@custom_op("wc::stats", mutates_args=())
def stats(x: Tensor, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """K1: x (M, C) float32 -> raw moments (sum (C,) | (G, C), xtx (C, C) | (G, C, C)) float64."""
    return ops.stats(x, groups)


@stats.register_fake
def _(x, groups=1):
    C = x.shape[-1]
    lead = (groups,) if groups > 1 else ()
    return _E(x, *lead, C, dtype=torch.float64), _E(x, *lead, C, C, dtype=torch.float64)


@custom_op("wc::factor", mutates_args=("moving_mean", "moving_cov"))
def factor(s: Optional[Tensor], xtx: Optional[Tensor], M: int, C: int, eps: float, momentum: float, ddof: int, training: bool,
           moving_mean: Optional[Tensor], moving_cov: Optional[Tensor], groups: int = 1) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """K2: -> (mu f32, L f64, W f64, chan_scale f32); updates the moving statistics in place when training."""
    dev = (s if s is not None else moving_cov).device
    mm = moving_mean.view(-1) if moving_mean is not None else None
    return ops.factor(s, xtx, M, C, eps, momentum, ddof, training, mm, moving_cov, dev, want_scale=True, groups=groups)


@factor.register_fake
def _(s, xtx, M, C, eps, momentum, ddof, training, moving_mean, moving_cov, groups=1):
    like = s if s is not None else moving_cov
    lead = (groups,) if groups > 1 else ()
    return (_E(like, *lead, C, dtype=torch.float32), _E(like, *lead, C, C, dtype=torch.float64),
            _E(like, *lead, C, C, dtype=torch.float64), _E(like, C, dtype=torch.float32))


@custom_op("wc::color", mutates_args=())
def color(W: Tensor, gamma: Optional[Tensor], chan_scale: Tensor, groups: int = 1, per_group: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """A_k = W^T Gamma_k, At_k = A_k^T and the apply plan (opaque uint8; 1 byte when the width has no fast path)."""
    A, At, plan = ops.color(W, gamma, chan_scale, groups, per_group)
    if plan is None:
        plan = torch.zeros(1, dtype=torch.uint8, device=W.device)
    return A, At, plan


@color.register_fake
def _(W, gamma, chan_scale, groups=1, per_group=False):
    C = W.shape[-1]
    Kc = 1 if gamma is None else (gamma.shape[0] // groups if per_group else gamma.shape[0])
    n = groups * Kc
    from . import _lib
    nbytes = int(_lib.load().wc_apply_plan_bytes(C, n)) if C in (32, 64, 128, 256) else 0      # host-side size query, no GPU
    return (_E(W, n, C, C, dtype=torch.float32), _E(W, n, C, C, dtype=torch.float32), _E(W, max(nbytes, 256) if nbytes else 1, dtype=torch.uint8))


@custom_op("wc::apply", mutates_args=())
def apply(x: Tensor, mu: Optional[Tensor], A: Tensor, bias: Optional[Tensor], slot: Optional[Tensor],
          plan: Optional[Tensor], relu: bool = False) -> Tensor:
    """K3: y[n] = (x[n] - mu) A[slot[n]] + bias[slot[n]] (max(., 0) with relu)."""
    if plan is not None and plan.numel() <= 1:
        plan = None
    return ops.apply(x, mu, A, bias, slot, plan=plan, relu=relu)


@apply.register_fake
def _(x, mu, A, bias, slot, plan, relu=False):
    return torch.empty_like(x)


@custom_op("wc::bwd_reduce", mutates_args=())
def bwd_reduce(x: Tensor, mu: Optional[Tensor], gy: Tensor, slot: Optional[Tensor], Kc: int) -> Tuple[Tensor, Tensor]:
    """K4: -> (R (Kc, C, C), gsum (Kc, C)) float64."""
    return ops.bwd_reduce(x, mu, gy, slot, Kc)


@bwd_reduce.register_fake
def _(x, mu, gy, slot, Kc):
    C = x.shape[-1]
    return _E(x, Kc, C, C, dtype=torch.float64), _E(x, Kc, C, dtype=torch.float64)


@custom_op("wc::bwd_factor", mutates_args=())
def bwd_factor(R: Tensor, gsum: Tensor, W: Tensor, L: Tensor, gamma: Optional[Tensor], A: Tensor, M: int, eps: float,
               ddof: int, training: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """K5: -> (dgamma, dbeta, S, gmean); outputs that do not apply come back empty (no coloring / eval mode)."""
    dg, db, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, eps, ddof, training)
    z = lambda: torch.empty(0, dtype=torch.float32, device=R.device)
    return (dg if dg is not None else z(), db if db is not None else z(), S if S is not None else z(), gm if gm is not None else z())


@bwd_factor.register_fake
def _(R, gsum, W, L, gamma, A, M, eps, ddof, training):
    Kc, C = R.shape[0], R.shape[1]
    f = torch.float32
    return (_E(R, *((Kc, C, C) if gamma is not None else (0,)), dtype=f), _E(R, Kc, C, dtype=f),
            _E(R, *((C, C) if training else (0,)), dtype=f), _E(R, *((C,) if training else (0,)), dtype=f))


@custom_op("wc::bwd_apply", mutates_args=())
def bwd_apply(gy: Tensor, x: Optional[Tensor], mu: Optional[Tensor], At: Tensor, S: Optional[Tensor], gmean: Optional[Tensor],
              slot: Optional[Tensor]) -> Tensor:
    """K6: dx[n] = gy[n] At[slot[n]] + (x[n] - mu) S - gmean."""
    return ops.bwd_apply(gy, x, mu, At, S, gmean, slot)


@bwd_apply.register_fake
def _(gy, x, mu, At, S, gmean, slot):
    return torch.empty_like(gy)


# ---------------------------------------------------------------------------------------------------------------------
# the fused site as ONE operator with autograd registered on it (forward = K1..K3, backward = K4..K6)
# ---------------------------------------------------------------------------------------------------------------------
@custom_op("wc::whiten_color", mutates_args=())
def whiten_color(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], slot: Optional[Tensor],
                 moving_mean: Optional[Tensor], moving_cov: Optional[Tensor], training: bool = True, eps: float = 1e-3,
                 momentum: float = 0.99, ddof: int = 1,
                 relu: bool = False) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (y, mu, L, W, A, At, new_moving_mean, new_moving_cov): y = coloring(whitening(x)); mu..At is what the backward
    needs.  FUNCTIONAL (an operator with an autograd formula may not mutate its inputs): the updated moving statistics
    come back as outputs -- `whiten_color_site` below writes them into the layer's buffers."""
    C = x.shape[-1]
    x = x.contiguous()
    M = x.numel() // C
    s = xtx = None
    if training:
        s, xtx = ops.stats(x.view(M, C))
    z = lambda: torch.empty(0, dtype=torch.float32, device=x.device)
    mm = moving_mean.reshape(-1).clone() if moving_mean is not None else None
    mc = moving_cov.clone() if moving_cov is not None else None
    mu, L, W, cs = ops.factor(s, xtx, M, C, eps, momentum, ddof, training, mm, mc, x.device, want_scale=True)
    A, At, plan = ops.color(W, gamma.contiguous() if gamma is not None else None, cs)
    y = ops.apply(x, mu, A, beta.contiguous() if beta is not None else None, slot, plan=plan, relu=relu)
    return y, mu, L, W, A, At, (mm if mm is not None else z()), (mc if mc is not None else z())


@whiten_color.register_fake
def _(x, gamma, beta, slot, moving_mean, moving_cov, training=True, eps=1e-3, momentum=0.99, ddof=1, relu=False):
    C = x.shape[-1]
    Kc = 1 if gamma is None else gamma.shape[0]
    f64, f32 = torch.float64, torch.float32
    return (torch.empty_like(x), _E(x, C, dtype=f32), _E(x, C, C, dtype=f64), _E(x, C, C, dtype=f64),
            _E(x, Kc, C, C, dtype=f32), _E(x, Kc, C, C, dtype=f32),
            _E(x, *((C,) if moving_mean is not None else (0,)), dtype=f32), _E(x, *((C, C) if moving_cov is not None else (0,)), dtype=f32))


def _wc_setup(ctx, inputs, output):
    x, gamma, beta, slot, moving_mean, moving_cov, training, eps, momentum, ddof, relu = inputs
    y, mu, L, W, A, At = output[:6]
    ctx.save_for_backward(x, gamma, slot, y if relu else None, mu, L, W, A, At)
    ctx.has_beta = beta is not None
    ctx.training, ctx.eps, ctx.ddof, ctx.relu = training, eps, ddof, relu


def _wc_backward(ctx, gy, *_unused):
    x, gamma, slot, y, mu, L, W, A, At = ctx.saved_tensors
    gy = gy.contiguous()
    if ctx.relu:
        gy = torch.ops.aten.threshold_backward(gy, y, 0.0)      # gy where y > 0, else 0: ONE elementwise pass (where(y > 0, ...) took three launches)
    C = x.shape[-1]
    M = x.numel() // C
    R, gsum = ops.bwd_reduce(x.contiguous(), mu, gy, slot, A.shape[0])
    dg, db, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, ctx.eps, ctx.ddof, ctx.training,
                                   want_dgamma=gamma is not None, want_dbeta=ctx.has_beta)
    dx = ops.bwd_apply(gy, x.contiguous(), mu, At, S, gm, slot)
    return dx, dg, (db if ctx.has_beta else None), None, None, None, None, None, None, None, None


whiten_color.register_autograd(_wc_backward, setup_context=_wc_setup)


def whiten_color_site(x, gamma=None, beta=None, slot=None, moving_mean=None, moving_cov=None, training=True, eps=1e-3,
                      momentum=0.99, ddof=1, relu=False):
    """The fused site through the operator: y, with the moving statistics updated in place (training mode)."""
    out = torch.ops.wc.whiten_color(x, gamma, beta, slot, moving_mean, moving_cov, training, eps, momentum, ddof, relu)
    if training and moving_mean is not None:
        with torch.no_grad():
            moving_mean.copy_(out[6].view_as(moving_mean)); moving_cov.copy_(out[7])
    return out[0]
