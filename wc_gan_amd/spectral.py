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

from . import _lib, ops


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


class SpectralNormBatchFunction(torch.autograd.Function):
    """Every spectrally normalised weight of a network in one launch (forward) and one launch (backward)."""

    @staticmethod
    def forward(ctx, mods, iterations, eps, fully_diff, *weights):
        wss = [m._sn_workspace() for m in mods]
        outs = ops.spectral_norm_batched([w.detach() for w in weights], [m.sn_u for m in mods], [m.sn_v for m in mods],
                                         wss, iterations, eps)
        saved = []
        for w_sn, sigma, uu, vv in outs:
            saved += [w_sn, uu, vv, sigma]
        ctx.save_for_backward(*saved)
        ctx.wss, ctx.fully_diff, ctx.n = wss, bool(fully_diff), len(mods)
        return tuple(o[0] for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        sv = ctx.saved_tensors
        idx = [i for i in range(ctx.n) if grads[i] is not None and ctx.needs_input_grad[4 + i]]
        out = [None] * ctx.n
        if idx:
            dWs = ops.spectral_norm_bwd_batched([grads[i] for i in idx], [sv[4 * i] for i in idx], [sv[4 * i + 1] for i in idx],
                                                [sv[4 * i + 2] for i in idx], [sv[4 * i + 3] for i in idx],
                                                [ctx.wss[i] for i in idx], ctx.fully_diff)
            for i, d in zip(idx, dWs):
                out[i] = d
        return (None, None, None, None, *out)


def prepare_spectral(root):
    """Call at the top of a network's forward: normalises the weights of ALL its spectral-norm layers in one launch;
    each layer's forward then picks its weight up (if that weight has not changed since).  No-op with fewer than two
    such layers, on the CPU, or when their settings differ."""
    mods = getattr(root, '_sn_modules', None)
    if mods is None:
        mods = root._sn_modules = [m for m in root.modules() if isinstance(m, _SNMixin)]
    if len(mods) < 2 or not all(m.weight.is_cuda for m in mods):
        return
    m0 = mods[0]
    it = m0.spectral_iterations if root.training else 0
    if any((m.spectral_iterations if m.training else 0) != it or m.fully_diff_spectral != m0.fully_diff_spectral
           or m.sn_eps != m0.sn_eps for m in mods):
        return
    ws = SpectralNormBatchFunction.apply(mods, it, m0.sn_eps, m0.fully_diff_spectral, *[m.weight for m in mods])
    for m, w in zip(mods, ws):
        w._wc_amax = m._sn_amax()                    # max|w_sn| as the op left it: the convolution's weight split needs no sweep
        m._w_ready = (w, m.weight._version, m.training)


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

    def _sn_workspace(self):
        ws = getattr(self, '_sn_ws', None)
        if ws is None or ws.device != self.weight.device:      # per-weight scratch of the op (zeroed once)
            ws = self._sn_ws = ops.spectral_norm_workspace(self.sn_u.numel(), self.sn_v.numel(), self.weight.device)
        return ws

    def _sn_amax(self):
        """The floats of the workspace where the forward op leaves per-workgroup maxima of |w_sn| (between the offset the
        library reports and the four sync words at the end; unused entries stay zero)."""
        a = getattr(self, '_sn_amax_view', None)
        ws = self._sn_workspace()
        if a is None or a.untyped_storage().data_ptr() != ws.untyped_storage().data_ptr():
            off = _lib.load().wc_spectral_norm_amax_offset(self.sn_u.numel(), self.sn_v.numel())
            a = self._sn_amax_view = ws[off:ws.numel() - 16].view(torch.float32)
        return a

    def normalized_weight(self):
        if not self.weight.is_cuda:
            raise RuntimeError("spectral normalisation runs on the HIP op only: move the module to the GPU")
        ready = getattr(self, '_w_ready', None)
        if ready is not None:                                   # normalised by prepare_spectral() for this pass
            self._w_ready = None
            if ready[1] == self.weight._version and ready[2] == self.training:
                return ready[0]
        it = self.spectral_iterations if self.training else 0
        ws = self._sn_workspace()
        w_sn, _sigma = SpectralNormFunction.apply(self.weight, self.sn_u, self.sn_v, ws, it, self.sn_eps, self.fully_diff_spectral)
        w_sn._wc_amax = self._sn_amax()
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
