"""Host-side mirror of the reference's layer surface for the WC path (torch.nn.Module based).

Names, constructor kwargs and NHWC semantics follow the call sites in the reference:

    DecorelationNormalization(name=, renorm=, decomposition=)      generator.py:24,26
    CenterScale(axis=, name=)                                      generator.py:32,37
    ConditionalCenterScale(number_of_classes=, axis=, name=)([x, cls])     generator.py:29-30,36
    ConditionalConv11(filters=, number_of_classes=, name=)([x, cls])       generator.py:42-44,55-57
    FactorizedConv11(number_of_classes=, filters=, filters_emb=, use_bias=, name=)([x, cls])   generator.py:46-48,72-76
    Conv11(filters=, name=)  -- the Keras Conv2D(kernel_size=(1,1)) coloring of generator.py:50-51

The classes themselves live in the reference's un-vendored `gan/` submodule, so weight shapes,
initialisers, epsilon and momentum are [UPSTREAM-RECALL] and exposed as constructor arguments.

Tensors are NHWC (Keras `axis=-1`); `cls` is the int (N, 1) class input of generator.py:102.
Modules build lazily on the first call (Keras `build(input_shape)`) unless `channels=` is given.

`WhiteningColoring` is the fused hot path `create_norm` substitutes whenever norm is 'd'/'dr':
it owns a DecorelationNormalization (`<name>_npart`) and the coloring layers (`<name>_repart*`)
and runs them as ONE statistics pass, one small float64 stage and ONE affine pass over x.
"""
from __future__ import annotations

import os

import math
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _state
from . import functional as WF


_TLS = threading.local()          # per host thread: two trainers on two threads do not see each other's setting


USE_FACTOR_MIX = os.environ.get("WC_FACTOR_MIX", "1") != "0"      # 0: the soft-assignment tables as torch matmul + add + gather (rounds 1-3)


def _stat_groups():
    return getattr(_TLS, 'groups', 1)


# WC_TORCH_OPS=1 (or layers.USE_TORCH_OPS = True): the layers call the fused site through torch.ops.wc.whiten_color instead of
# the ctypes wrappers.  Off by default: a Python custom op costs tens of microseconds of dispatch per call (the eager step
# makes ~300 such calls); inside a captured hipGraph both routes replay the same launches.
USE_TORCH_OPS = os.environ.get('WC_TORCH_OPS', '0') == '1'

class statistic_groups:
    """Context: WC layers treat the batch as `n` independent, equally sized batches stacked along N, each whitened
    with its own statistics (training mode, no autograd).  Equivalent to n separate forward passes -- for the layers
    that have the grouped form (`supports_statistic_groups`); every other normalisation layer RAISES inside the
    context instead of silently pooling the statistics of the n batches."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = _stat_groups()
        _TLS.groups = self.n
        return self

    def __exit__(self, *exc):
        _TLS.groups = self.prev
        return False


def supports_statistic_groups(module):
    """True when every batch-statistics layer under `module` honours statistic_groups(): the fused Cholesky whitening at
    C % 32 == 0 without renorm.  ZCA, renorm ('dr'), zero-padded widths and plain batch norm ('b') do not -- callers run
    separate passes instead (GanTrainer.generate)."""
    for m in module.modules():
        if isinstance(m, DecorelationNormalization):
            if m.renorm or m.decomposition != 'cholesky' or (m.channels is not None and m.channels % 32 != 0):
                return False
            if m.process_group is not None:
                return False                      # sync-WC: the grouped forward has no collective (per-replica statistics only)
            if m.channels is None:
                return False                      # not built yet: width unknown
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)) or type(m).__name__ == '_BatchNormNoAffine':
            return False
    return True


def _glorot_uniform_(t, fan_in, fan_out):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        return t.uniform_(-limit, limit)


def _cls_index(cls):
    return cls.reshape(-1).to(torch.int32).contiguous()


def _pad_channels(x, mult=32):
    C = x.shape[-1]
    Cp = (C + mult - 1) // mult * mult
    if Cp == C:
        return x, C
    return F.pad(x, (0, Cp - C)), C


class _Lazy(nn.Module):
    """Keras-style deferred build: parameters are created on the first call, from the input's channel count."""

    def __init__(self, name=None, channels=None):
        super().__init__()
        self.layer_name = name
        self.channels = None
        if channels is not None:
            self._build(int(channels))

    def _build(self, C, device=None):
        self.channels = C
        self.build(C, device)

    def _ensure(self, x):
        if self.channels is None:
            self._build(x.shape[-1], x.device)
        elif x.shape[-1] != self.channels:
            raise ValueError(f"{self.layer_name}: built for {self.channels} channels, got {x.shape[-1]}")


# ---------------------------------------------------------------------------------------------
# whitening
# ---------------------------------------------------------------------------------------------
class DecorelationNormalization(_Lazy):
    """Batch whitening.  State: moving_mean (C,1), moving_cov (C,C)."""

    def __init__(self, name=None, renorm=False, decomposition='cholesky', momentum=0.99, epsilon=1e-3, axis=-1,
                 channels=None, process_group=None):
        if decomposition not in ('cholesky', 'zca'):
            raise ValueError("decomposition must be 'cholesky' or 'zca'")
        if axis not in (-1, 3):
            raise ValueError("the WC path is NHWC: axis must be -1")
        self.renorm, self.decomposition = bool(renorm), decomposition
        self.momentum, self.epsilon, self.process_group = float(momentum), float(epsilon), process_group
        super().__init__(name, channels)

    def build(self, C, device=None):
        self.register_buffer('moving_mean', torch.zeros(C, 1, device=device))
        self.register_buffer('moving_cov', torch.eye(C, device=device))
        if C % 32 != 0:
            # zero-padded fallback (_padded): its state exists from the build on -- registering buffers inside forward
            # would allocate under a graph capture
            Cp = (C + 31) // 32 * 32
            self.register_buffer('_pad_mean', torch.zeros(Cp, 1, device=device), persistent=False)
            self.register_buffer('_pad_cov', torch.eye(Cp, device=device), persistent=False)
            self.register_buffer('_pad_eye', torch.eye(Cp, device=device), persistent=False)

    def transform(self, x, gamma=None, beta=None, slot=None, gamma_key=None, relu=False, per_sample=False, planes=False):
        """Whitening fused with an optional coloring table (gamma (Kc,C,C), beta (Kc,C), slot (N,)); relu=True also
        folds the ReLU that follows the site into the apply kernel where that path has it (else applied after).
        per_sample: the table holds one entry per sample (slot = arange(N)); only the grouped path needs to know.
        planes: the caller is conv.fast_conv and takes the output as its fp16 planes (functional.whiten_color) where K3 can
        leave it so; the other paths return the plain tensor."""
        self._ensure(x)
        C = self.channels
        groups = _stat_groups() if self.training else 1
        if WF.split_of(x) is not None and not self.takes_split(x.shape):
            # (the producer asks takes_split() before it writes planes: this is a wiring error, not a data-dependent case)
            raise RuntimeError(f"{self.layer_name}: a pre-split handle reached a route without a planes path")
        if groups > 1 and (C % 32 != 0 or self.decomposition != 'cholesky' or self.renorm):
            raise RuntimeError(f"{self.layer_name}: statistic_groups({groups}) has no grouped form for this layer "
                               "(zca / renorm / a width that is not a multiple of 32): run separate passes")
        if groups > 1 and self.process_group is not None:
            # the grouped forward whitens with per-replica moments and updates the moving statistics from the local batch only:
            # under sync-WC that would silently differ from g_step's all-reduced statistics and let the replicas' moving
            # statistics drift apart
            raise RuntimeError(f"{self.layer_name}: statistic_groups({groups}) has no sync-WC form (process_group is set): "
                               "run separate passes")
        if C % 32 != 0:
            y = self._padded(x, gamma, beta, slot)
            return F.relu(y) if relu else y
        if self.decomposition == 'zca':
            if self.renorm:
                raise NotImplementedError("renorm is defined for decomposition='cholesky' only")
            y = WF.whiten_color_modular(x, gamma, beta, slot, self.moving_mean, self.moving_cov, self.training,
                                        self.epsilon, self.momentum, 1, 'zca')
            return F.relu(y) if relu else y
        if groups > 1:
            if torch.is_grad_enabled() and (x.requires_grad or (gamma is not None and gamma.requires_grad)):
                raise RuntimeError("statistic_groups() is a forward-only path: wrap the call in torch.no_grad()")
            return WF.whiten_color_grouped(x, groups, gamma, beta, slot, self.moving_mean, self.moving_cov,
                                           self.epsilon, self.momentum, 1, relu=relu, per_sample=per_sample, planes=planes)
        if not self.training and not torch.is_grad_enabled():
            # inference (scorer.py:60,72): moving statistics are constants -> cached factorisation, one K3 launch
            if not hasattr(self, '_eval_plan'):
                self._eval_plan = WF.EvalPlan()
            return WF.whiten_color_eval_cached(x, self._eval_plan, gamma, beta, slot, self.moving_mean, self.moving_cov,
                                               self.epsilon, gamma_key, relu=relu, planes=planes)
        if self.renorm and self.training:
            gamma = self._renorm_gamma(x, gamma)
        if USE_TORCH_OPS and self.process_group is None:
            # the same site through the registered operator torch.ops.wc.whiten_color (torch_ops.py: schema, fake kernel,
            # autograd formula on the op) -- what torch.compile / FX tooling see; no hand-off, no bit mask on this route
            from . import torch_ops
            return torch_ops.whiten_color_site(x, gamma, beta, slot, self.moving_mean, self.moving_cov, self.training,
                                               self.epsilon, self.momentum, 1, relu)
        return WF.whiten_color(x, gamma, beta, slot, self.moving_mean, self.moving_cov, self.training,
                               self.epsilon, self.momentum, 1, self.process_group, relu=relu, planes=planes)

    def takes_split(self, shape):
        """Can this layer, in its present mode, read an input of this NHWC shape as pre-split planes (the residual add in front then
        writes those instead of fp32: functional.residual_add)?  The fused Cholesky route only, C in {128, 256}, the shapes of
        functional.split_route_supported."""
        if self.channels is None or self.channels != shape[-1] or self.decomposition != 'cholesky' or USE_TORCH_OPS:
            return False
        if self.renorm and self.training:
            return False
        groups = _stat_groups() if self.training else 1
        if groups > 1 and self.process_group is not None:
            return False
        if shape[0] % groups != 0:
            return False
        return WF.split_route_supported(tuple(shape), self.training, groups)

    def _renorm_gamma(self, x, gamma):
        # W_eff = L_mov^-1 . stop_grad(L_batch) . L_batch^-1 (row a4): the batch factor carries the gradient,
        # the value is the moving-statistics whitening.  Folded into the coloring: Gamma' = (L_mov^-1 L_batch)^T Gamma.
        C = self.channels
        with torch.no_grad():
            from . import ops
            s, xtx = ops.stats(x.contiguous().view(-1, C))
            M = x.numel() // C
            _, Lb, _ = ops.factor(s, xtx, M, C, self.epsilon, self.momentum, 1, True, None, None, x.device)
            _, _, Wm = ops.factor(None, None, M, C, self.epsilon, self.momentum, 1, False,
                                  self.moving_mean.view(-1), self.moving_cov, x.device)
            C0t = (Wm @ Lb).t().to(torch.float32)
        if gamma is None:
            return C0t.unsqueeze(0).contiguous()
        return torch.matmul(C0t.unsqueeze(0), gamma)

    def _padded(self, x, gamma, beta, slot):
        # zero channels whiten to zero and leave the real channels' Cholesky rows untouched
        if self.decomposition != 'cholesky' or self.renorm:
            raise NotImplementedError(f"{self.layer_name}: widths that are not a multiple of 32 are built for "
                                      "decomposition='cholesky' without renorm only")
        C = self.channels
        xp, _ = _pad_channels(x)
        Cp = xp.shape[-1]
        with torch.no_grad():
            self._pad_mean.zero_(); self._pad_mean[:C] = self.moving_mean
            self._pad_cov.copy_(self._pad_eye); self._pad_cov[:C, :C] = self.moving_cov
        Kc = 1 if gamma is None else gamma.shape[0]
        g = self._pad_eye.repeat(Kc, 1, 1)
        g[:, :C, :C] = gamma if gamma is not None else self._pad_eye[:C, :C]
        b = None if beta is None else F.pad(beta, (0, Cp - C))
        y = WF.whiten_color(xp.contiguous(), g, b, slot, self._pad_mean, self._pad_cov, self.training,
                            self.epsilon, self.momentum, 1, self.process_group)
        with torch.no_grad():
            self.moving_mean.copy_(self._pad_mean[:C]); self.moving_cov.copy_(self._pad_cov[:C, :C])
        return y[..., :C]

    def forward(self, x):
        return self.transform(x)


# ---------------------------------------------------------------------------------------------
# coloring layers.  Each exposes table(cls) -> (gamma (Kc,C,C), beta (Kc,C) | None, slot (N,) | None),
# which is what the fused path consumes; forward() applies the layer on its own (unfused use).
# ---------------------------------------------------------------------------------------------
class _Coloring(_Lazy):
    conditional = False

    def table(self, cls=None):
        raise NotImplementedError

    def forward(self, inputs):
        x, cls = (inputs if isinstance(inputs, (list, tuple)) else (inputs, None))
        self._ensure(x)
        gamma, beta, slot = self.table(cls)
        xp, C = _pad_channels(x)
        if xp.shape[-1] != C:
            Cp = xp.shape[-1]
            g = torch.zeros(gamma.shape[0], Cp, Cp, device=x.device); g[:, :C, :C] = gamma
            beta = None if beta is None else F.pad(beta, (0, Cp - C))
            gamma = g
        y = WF.AffineRowsFunction.apply(xp.contiguous(), None, gamma, beta, slot)
        return y[..., :C] if xp.shape[-1] != C else y


class Conv11(_Coloring):
    """Keras Conv2D(kernel_size=(1,1), filters=C): kernel (1,1,C_in,C_out), bias (C).  generator.py:50-51."""

    def __init__(self, filters=None, name=None, use_bias=True, kernel_size=(1, 1), channels=None):
        assert tuple(kernel_size) == (1, 1)
        self.use_bias = use_bias
        super().__init__(name, channels if channels is not None else filters)

    def build(self, C, device=None):
        self.kernel = nn.Parameter(_glorot_uniform_(torch.empty(1, 1, C, C, device=device), C, C))
        self.bias = nn.Parameter(torch.zeros(C, device=device)) if self.use_bias else None

    def table(self, cls=None):
        C = self.channels
        return self.kernel.view(1, C, C), (self.bias.view(1, C) if self.bias is not None else None), None


class ConditionalConv11(_Coloring):
    """Per-class 1x1 convolution: kernel (K, C_in, C_out), bias (K, C).  generator.py:42-44,55-57."""
    conditional = True

    def __init__(self, filters=None, number_of_classes=10, name=None, use_bias=True, channels=None):
        self.number_of_classes, self.use_bias = int(number_of_classes), use_bias
        super().__init__(name, channels if channels is not None else filters)

    def build(self, C, device=None):
        K = self.number_of_classes
        self.kernel = nn.Parameter(_glorot_uniform_(torch.empty(K, C, C, device=device), C, C))
        self.bias = nn.Parameter(torch.zeros(K, C, device=device)) if self.use_bias else None

    def table(self, cls=None):
        return self.kernel, self.bias, _cls_index(cls)


class FactorizedConv11(_Coloring):
    """Soft-assignment coloring (cWC_sa): Gamma_y = sum_e alpha[y,e] Gamma_e.  generator.py:46-48,72-76."""
    conditional = True

    def __init__(self, number_of_classes=10, filters=None, filters_emb=10, use_bias=False, name=None, channels=None):
        self.number_of_classes, self.filters_emb, self.use_bias = int(number_of_classes), int(filters_emb), use_bias
        super().__init__(name, channels if channels is not None else filters)

    def build(self, C, device=None):
        K, E = self.number_of_classes, self.filters_emb
        self.kernel = nn.Parameter(_glorot_uniform_(torch.empty(E, C, C, device=device), C, C))
        self.class_matrix = nn.Parameter(_glorot_uniform_(torch.empty(K, E, device=device), K, E))
        self.bias = nn.Parameter(torch.zeros(K, C, device=device)) if self.use_bias else None

    def table(self, cls=None):
        E, C = self.filters_emb, self.channels
        gamma = (self.class_matrix @ self.kernel.view(E, C * C)).view(self.number_of_classes, C, C)
        return gamma, self.bias, _cls_index(cls)


class CenterScale(_Coloring):
    """Per-channel gamma * x + beta (diagonal coloring, "WC-diag").  generator.py:32,37."""

    def __init__(self, axis=-1, name=None, channels=None):
        super().__init__(name, channels)

    def build(self, C, device=None):
        self.gamma = nn.Parameter(torch.ones(C, device=device))
        self.beta = nn.Parameter(torch.zeros(C, device=device))

    def table(self, cls=None):
        return torch.diag(self.gamma).unsqueeze(0), self.beta.view(1, -1), None

    def forward(self, inputs):
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        self._ensure(x)
        return x * self.gamma + self.beta


class ConditionalCenterScale(_Coloring):
    """Per-class, per-channel gamma_y * x + beta_y.  generator.py:29-30,36."""
    conditional = True

    def __init__(self, number_of_classes=10, axis=-1, name=None, channels=None):
        self.number_of_classes = int(number_of_classes)
        super().__init__(name, channels)

    def build(self, C, device=None):
        K = self.number_of_classes
        self.gamma = nn.Parameter(torch.ones(K, C, device=device))
        self.beta = nn.Parameter(torch.zeros(K, C, device=device))

    def table(self, cls=None):
        return torch.diag_embed(self.gamma), self.beta, _cls_index(cls)

    def forward(self, inputs):
        x, cls = inputs
        self._ensure(x)
        idx = cls.reshape(-1).long()
        shape = (x.shape[0],) + (1,) * (x.dim() - 2) + (x.shape[-1],)
        return x * self.gamma[idx].view(shape) + self.beta[idx].view(shape)


# ---------------------------------------------------------------------------------------------
# the fused hot path
# ---------------------------------------------------------------------------------------------
class WhiteningColoring(nn.Module):
    """`stack(inp)` of generator.py:83-87 as one fused op: y = coloring(whitening(x)).

    `branches` are coloring layers whose outputs the reference adds (`Add`, generator.py:39,58,66,77);
    their tables add too, so any after_norm value is a single (Gamma_eff, beta_eff, slot) for K3/K5.
    """

    def __init__(self, npart: DecorelationNormalization, branches):
        super().__init__()
        self.npart = npart
        self.branches = nn.ModuleList(branches)

    def _mixed_table(self, x, cls):
        """cWC_sa (after_norm ufconv / fconv: one FactorizedConv11 beside at most unconditional 1x1 branches) on a HIP tensor: the tables
        the batch uses straight from the dictionary (wc_factor_mix_f32) -- no (K, C, C) table of all classes, no broadcast add, no gather."""
        fact = [br for br in self.branches if isinstance(br, FactorizedConv11)]
        rest = [br for br in self.branches if not isinstance(br, FactorizedConv11)]
        if not USE_FACTOR_MIX or len(fact) != 1 or fact[0].use_bias or not all(type(br) is Conv11 for br in rest) or not x.is_cuda or cls is None:
            return None
        for br in self.branches:
            br._ensure(x)
        f = fact[0]
        C, K, N = f.channels, f.number_of_classes, x.shape[0]
        if not WF.ops.factor_mix_supported(f.filters_emb, C):
            return None
        base = beta = None
        for br in rest:
            base = br.kernel.view(C, C) if base is None else base + br.kernel.view(C, C)
            if br.bias is not None:
                beta = br.bias.view(1, C) if beta is None else beta + br.bias.view(1, C)
        slot = _cls_index(cls)
        groups = _stat_groups() if self.npart.training else 1
        per_sample = K > N // max(groups, 1)
        gamma = WF.factor_mix(f.kernel, f.class_matrix, slot if per_sample else None, base)
        if per_sample:
            slot = torch.arange(N, dtype=torch.int32, device=x.device)
        if beta is not None:
            beta = beta.expand(gamma.shape[0], -1)
        return gamma, beta, slot, per_sample

    def coloring_table(self, x, cls):
        mixed = self._mixed_table(x, cls)
        if mixed is not None:
            return mixed
        gamma = beta = slot = None
        for br in self.branches:
            br._ensure(x)
            g, b, s = br.table(cls)
            if s is not None:
                slot = s
            gamma = g if gamma is None else gamma + g          # (1,C,C) broadcasts against (K,C,C)
            if b is not None:
                beta = b if beta is None else beta + b
        per_sample = False
        if gamma is not None and slot is not None:
            K, N = gamma.shape[0], x.shape[0]
            groups = _stat_groups() if self.npart.training else 1
            # more classes than samples (per statistic group: the table of a grouped batch is groups x Kc entries):
            # one table per SAMPLE instead of one per class (run.py:172-173: 200 / 1000 classes at batch 64)
            if K > N // max(groups, 1):
                idx = slot.long()
                gamma = gamma.expand(K, -1, -1)[idx] if gamma.shape[0] == K else gamma
                if beta is not None:
                    beta = beta.expand(K, -1)[idx]
                slot = torch.arange(N, dtype=torch.int32, device=x.device)
                per_sample = True
        if gamma is not None and beta is not None and beta.shape[0] != gamma.shape[0]:
            beta = beta.expand(gamma.shape[0], -1)
        return gamma, beta, slot, per_sample

    def takes_split(self, shape):
        """See DecorelationNormalization.takes_split (every coloring variant reduces to one table: no further condition)."""
        return self.npart.takes_split(shape)

    def wants_moments(self, shape):
        """The statistic groups for which a producer may accumulate this site's covariance partials in its own pass (functional.residual_add's
        stat_groups), 0 when the site will not take them: training mode on the fused Cholesky route only."""
        n = self.npart
        if not n.training or not self.takes_split(shape):
            return 0
        return _stat_groups()

    def backward_takes_split(self, shape):
        """Will the backward of this site read x from the planes as well (functional.USE_BWD_XSPLIT: K4 / K6 on planes)?  Then the
        producer need not write an fp32 copy of the sum beside them."""
        has_slot = any(br.conditional for br in self.branches)
        return WF.USE_BWD_XSPLIT and self.npart.training and WF.ops.bwd_xsplit_supported(tuple(shape), has_slot)

    def forward(self, x, cls=None, relu=False, planes=False):
        if isinstance(x, (list, tuple)):
            x, cls = x
        gamma, beta, slot, per_sample = self.coloring_table(x, cls)
        if gamma is not None:
            gamma = gamma.contiguous()
        if beta is not None:
            beta = beta.contiguous()
        # identity of the coloring weights (for the eval-mode plan cache); per-sample tables depend on cls -> no key
        key = None if per_sample else (_state.replays,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        return self.npart.transform(x, gamma, beta, slot, gamma_key=key, relu=relu, per_sample=per_sample, planes=planes)
