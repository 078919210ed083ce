"""Spectrally normalised layers on the fused HIP op.

Stands in for gan.spectral_normalized_layers.{SNConv2D, SNDense, SNEmbeding} as the reference builds them at
discriminator.py:26-33 and generator.py:104-113: keyword surface `spectral_iterations`, `fully_diff_spectral`,
`conv_singular` (run.py:265-270).  [UPSTREAM-RECALL] the layers' arithmetic (the submodule is not vendored): the
power-iteration estimate of Miyato et al., one persistent `u` per layer, `iterations` steps per training forward,
sigma = u^T W v, kernel / sigma; `fully_diff_spectral` lets the gradient flow through sigma.
`conv_singular` (singular value of the convolution operator instead of the reshaped kernel) is accepted and ignored:
the reshaped-kernel sigma is used, as in the SN-GAN paper the reference cites.

Every forward is ONE kernel launch (wc_spectral_norm_f32) and every backward one (wc_spectral_norm_bwd_f32); there is
no non-HIP path: the tensors must live on the GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class SpectralNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, u, v, ws, iterations, eps, fully_diff):
        # u, v as used for sigma come back as copies from the same launch: later forwards move the buffers on
        w_sn, sigma, uu, vv = ops.spectral_norm(weight.detach(), u, v, iterations, ws, eps, keep_uv=True)
        ctx.save_for_backward(w_sn, uu, vv, sigma)
        ctx.fully_diff = bool(fully_diff)
        ctx.ws = ws
        ctx.mark_non_differentiable(sigma)
        return w_sn, sigma

    @staticmethod
    def backward(ctx, g, _gs):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None, None
        w_sn, u, v, sigma = ctx.saved_tensors
        return ops.spectral_norm_bwd(g, w_sn, u, v, sigma, ctx.fully_diff, ctx.ws), None, None, None, None, None, None


class _SNMixin:
    """Adds the persistent power-iteration state to a module that owns `self.weight`."""

    def _sn_init(self, spectral_iterations=1, fully_diff_spectral=False, conv_singular=True, eps=1e-12):
        self.spectral_iterations = int(spectral_iterations)
        self.fully_diff_spectral = bool(fully_diff_spectral)
        self.conv_singular = bool(conv_singular)        # accepted, see the module docstring
        self.sn_eps = float(eps)
        w = self.weight.detach()
        rows = w.shape[0]
        wm = self._as_matrix(w)
        g = torch.Generator(device='cpu'); g.manual_seed(rows * 7919 + wm.shape[1])
        u = F.normalize(torch.randn(rows, generator=g), dim=0)
        v = F.normalize(torch.randn(wm.shape[1], generator=g), dim=0)
        for _ in range(15):                              # construction-time warm-up, as torch's parametrisation does
            v = F.normalize(wm.t().mv(u), dim=0, eps=eps)
            u = F.normalize(wm.mv(v), dim=0, eps=eps)
        self.register_buffer('sn_u', u.contiguous())
        self.register_buffer('sn_v', v.contiguous())

    @staticmethod
    def _as_matrix(w):
        """(rows, cols) view in MEMORY order (channels_last kernels: columns run (kh, kw, cin))."""
        if w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous():
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return w.reshape(w.shape[0], -1)

    def normalized_weight(self):
        if not self.weight.is_cuda:
            raise RuntimeError("spectral normalisation runs on the HIP op only: move the module to the GPU")
        it = self.spectral_iterations if self.training else 0
        ws = getattr(self, '_sn_ws', None)
        if ws is None or ws.device != self.weight.device:      # per-weight scratch of the op (zeroed once)
            ws = self._sn_ws = ops.spectral_norm_workspace(self.sn_u.numel(), self.sn_v.numel(), self.weight.device)
        w_sn, _sigma = SpectralNormFunction.apply(self.weight, self.sn_u, self.sn_v, ws, it, self.sn_eps, self.fully_diff_spectral)
        return w_sn


class SNConv2d(nn.Conv2d, _SNMixin):
    def __init__(self, *args, spectral_iterations=1, fully_diff_spectral=False, conv_singular=True, **kw):
        super().__init__(*args, **kw)
        self.to(memory_format=torch.channels_last)
        self._sn_init(spectral_iterations, fully_diff_spectral, conv_singular)

    def forward(self, x):
        return self._conv_forward(x, self.normalized_weight(), self.bias)


class SNLinear(nn.Linear, _SNMixin):
    def __init__(self, *args, spectral_iterations=1, fully_diff_spectral=False, **kw):
        super().__init__(*args, **kw)
        self._sn_init(spectral_iterations, fully_diff_spectral)

    def forward(self, x):
        return F.linear(x, self.normalized_weight(), self.bias)


class SNEmbedding(nn.Embedding, _SNMixin):
    def __init__(self, *args, spectral_iterations=1, fully_diff_spectral=False, **kw):
        super().__init__(*args, **kw)
        self._sn_init(spectral_iterations, fully_diff_spectral)

    def forward(self, idx):
        return F.embedding(idx, self.normalized_weight())
