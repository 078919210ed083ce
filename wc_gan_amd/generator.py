"""`create_norm` and `make_generator` with the reference's signatures (generator.py:13-17, 93-98).

`create_norm(norm, after_norm, ...)` returns `result_norm(axis, name)` exactly as in the reference
(generator.py:82-90); calling that gives the `stack` callable.  Differences forced by define-by-run
PyTorch: `stack(inp, cls)` receives the class tensor at call time (the reference closes over the
symbolic `cls` Input, generator.py:102,131-139), and a stack is an nn.Module so its weights register.

Whenever norm is 'd' / 'dr' the stack is the fused `WhiteningColoring` (one statistics pass, one
float64 C x C stage, one affine pass).  Sub-layer naming follows generator.py:85-86 and 36-38, 55-58:
`<name>_npart`, `<name>_repart`, `<name>_repart_c`, `<name>_repart_u`.

The ResNet block body (`gan.layer_utils.resblock`) is in the un-vendored submodule; the block here is
the SN-GAN generator block it implements [UPSTREAM-RECALL]: norm -> relu -> upsample -> conv3x3 ->
norm -> relu -> conv3x3, plus an upsample -> conv1x1 shortcut (SURVEY.md row a2 site list).
"""
from __future__ import annotations

from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import conv as fast_conv_mod
from .functional import residual_add, split_of
from .layers import (CenterScale, ConditionalCenterScale, ConditionalConv11, Conv11, DecorelationNormalization,
                     FactorizedConv11, WhiteningColoring)

# the block convolutions on the split-fp16 MFMA kernel where it takes the shape (WC_FAST_CONV=0: MIOpen everywhere)
FAST_CONV = os.environ.get('WC_FAST_CONV', '1') != '0'
# a WC site whose only reader is such a convolution writes that convolution's fp16 operand planes from its apply kernel
# (WC_HANDOFF=0: fp32 out of the site, absmax + split in front of the convolution)
HANDOFF = os.environ.get('WC_HANDOFF', '1') != '0'
# the residual add of a block writes the next site's input as pre-split fp16 planes where every reader has a planes path
# (SURVEY.md section 8f row N2 "residual Add feeding K1"; WC_SPLIT_PRODUCER=0: the fp32 sum, from the same HIP kernel)
SPLIT_PRODUCER = os.environ.get('WC_SPLIT_PRODUCER', '1') != '0'

NORMS = ['n', 'b', 'd', 'dr']
AFTER_NORMS = ['ucs', 'ccs', 'uccs', 'uconv', 'fconv', 'ufconv', 'cconv', 'ucconv', 'ccsuconv', 'n']


# ---------------------------------------------------------------------------------------------
# NHWC plumbing around the convolutions (wc_gan_amd/conv.py where the kernel takes the shape, torch / MIOpen otherwise)
# ---------------------------------------------------------------------------------------------
def to_nchw_view(x):
    return x.permute(0, 3, 1, 2)          # NHWC-contiguous -> channels_last NCHW view, zero copy


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()   # no copy when x is channels_last


class _NarrowConv3x3(torch.autograd.Function):
    """3x3 'same' convolution to a handful of output channels (the generator's last layer, generator.py:155-157:
    256 -> 3) as ONE GEMM + col2im: y[p] = sum_taps Z[p + offset][tap] with Z = x @ W_all  (M x Cin @ Cin x 9 Cout).
    MIOpen's implicit-GEMM kernels waste their 16-wide output tile on 3 channels (0.57 ms at 320 x 32 x 32 x 256
    against 0.16 ms here, measured); the backward keeps MIOpen's kernels, which are the faster ones there."""

    @staticmethod
    def forward(ctx, x, w, b):                      # x (N,H,W,Cin) contiguous NHWC; w (Cout,Cin,3,3); b (Cout,) | None
        N, H, W, C = x.shape
        O = w.shape[0]
        wall = w.flip(2, 3).permute(1, 0, 2, 3).reshape(C, O * 9)        # [c, (o, a, b)] = w[o, c, 2-a, 2-b]
        z = x.reshape(N * H * W, C) @ wall
        y = F.fold(z.view(N, H * W, O * 9).transpose(1, 2), (H, W), kernel_size=3, padding=1)     # (N, O, H, W)
        if b is not None:
            y = y + b.view(1, O, 1, 1)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y.permute(0, 2, 3, 1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        # round 5: the weight gradient in one pass over x on the fp32 matrix pipe (conv.narrow_out_weight_gradient: 256 -> 3 is the
        # narrow-INPUT gradient with the operands exchanged); MIOpen keeps the data gradient and the 3-element bias gradient
        own = fast_conv_mod.narrow_out_wrw_supported(x, w) and x.is_contiguous()
        own_w = ctx.needs_input_grad[1] and own
        own_x = ctx.needs_input_grad[0] and own
        gx, gw, gb = torch.ops.aten.convolution_backward(
            g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, [w.shape[0]] if ctx.has_bias else None,
            [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
            [ctx.needs_input_grad[0] and not own_x, ctx.needs_input_grad[1] and not own_w, ctx.has_bias and ctx.needs_input_grad[2]])
        gx = gx.permute(0, 2, 3, 1) if gx is not None else None
        gc = g.contiguous() if (own_w or own_x) else None
        if own_w:
            gw = fast_conv_mod.narrow_out_weight_gradient(x, gc, w)
        if own_x:       # the data gradient = the narrow-input FORWARD of gy through the mirrored, transposed weight
            gx = fast_conv_mod.narrow_forward(gc, w, None, mirrored=True)
        return gx, gw, (gb if ctx.has_bias else None)


class Conv2D(nn.Module):
    """Keras-style Conv2D on NHWC tensors (padding='same'); glorot-uniform kernel, zero bias."""

    def __init__(self, in_channels, filters, kernel_size=(3, 3), use_bias=True, name=None, spectral=False,
                 spectral_iterations=1, fully_diff_spectral=False, conv_singular=True):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        if spectral:        # SNConv2D (generator.py:105-106, discriminator.py:27-28): the fused HIP op
            from .spectral import SNConv2d
            self.conv = SNConv2d(in_channels, filters, k, padding=k // 2, bias=use_bias,
                                 spectral_iterations=spectral_iterations, fully_diff_spectral=fully_diff_spectral,
                                 conv_singular=conv_singular)
            with torch.no_grad():
                nn.init.xavier_uniform_(self.conv.weight)
                self.conv._sn_init(spectral_iterations, fully_diff_spectral, conv_singular)     # u, v for the new kernel
        else:
            self.conv = nn.Conv2d(in_channels, filters, k, padding=k // 2, bias=use_bias)
            nn.init.xavier_uniform_(self.conv.weight)
        if use_bias:
            nn.init.zeros_(self.conv.bias)
        self.conv = self.conv.to(memory_format=torch.channels_last)
        self.layer_name = name

    def _weight(self):
        conv = self.conv
        return conv.normalized_weight() if hasattr(conv, 'normalized_weight') else conv.weight

    def takes_planes(self, shape, kind='same'):
        """Will forward / forward_upsampled read an input of this shape as planes handed over by the WC site in front of it?"""
        c = self.conv
        if not (FAST_CONV and HANDOFF) or (c.out_channels <= 4 and not hasattr(c, 'normalized_weight')):
            return False
        if kind == 'up3' and tuple(c.weight.shape[2:]) != (3, 3):
            return False
        return c.weight.dtype == torch.float32 and fast_conv_mod.takes_planes(shape, c.weight.shape, kind)

    def takes_split(self, shape):
        """Will forward() read an input of this NHWC shape as the pre-split planes the residual add wrote (functional.residual_add)?
        The 1x1 shortcut of a block: 1 / scale and centre fold into its weight and bias (conv.split_conv)."""
        c = self.conv
        if not (FAST_CONV and SPLIT_PRODUCER) or tuple(c.kernel_size) != (1, 1) or c.weight.dtype != torch.float32:
            return False
        return fast_conv_mod.takes_planes(shape, c.weight.shape, 'same')

    def forward_relu(self, x):
        """conv(relu(x)): on the fast path the ReLU happens while the activation is split (one kernel and one pass less)"""
        w = self._weight()
        if FAST_CONV and x.is_cuda:
            y = fast_conv_mod.fast_conv_or_none(x, w, self.conv.bias, 'same', relu_input=True, site=self)
            if y is not None:
                return y
        return self.forward(F.relu(x), _w=w)

    def forward(self, x, _w=None):
        c = self.conv
        st = split_of(x)
        if st is not None:      # the block input as pre-split planes: the convolution reads those (weight and bias folded)
            return fast_conv_mod.split_conv(x, st, self._weight() if _w is None else _w, c.bias, site=self)
        if (c.out_channels <= 4 and tuple(c.kernel_size) == (3, 3) and not hasattr(c, 'normalized_weight')
                and x.is_cuda and x.is_contiguous()):
            return _NarrowConv3x3.apply(x, c.weight, c.bias)
        # the weight ONCE per forward: a spectrally normalised layer advances its power iteration in normalized_weight()
        w = self._weight() if _w is None else _w
        if FAST_CONV and x.is_cuda:
            y = fast_conv_mod.fast_conv_or_none(x, w, c.bias, 'same', site=self)      # split-fp16 MFMA implicit GEMM (csrc/wc_conv.hip)
            if y is not None:
                return y
        if x.is_cuda and fast_conv_mod.narrow_wrw_supported(x, w):
            return fast_conv_mod.narrow_in_conv(x, w, c.bias)       # an image-like input: forward and weight / bias gradient on the fp32 matrix pipe (csrc/wc_conv.hip)
        return to_nhwc(c._conv_forward(to_nchw_view(x), w, c.bias))

    def forward_upsampled(self, x):
        """conv(upsample2x(x)) for a 3x3 'same' convolution, without the upsampled tensor: a nearest-neighbour 2x
        upsample followed by a 3x3 convolution is ONE transposed convolution with a 4x4 kernel, stride 2, padding 1 --
        source pixel y feeds output rows 2y-1 .. 2y+2, each through the sum of the 3x3 taps that read an upsampled
        copy of y there ([w2, w1+w2, w0+w1, w0] along each axis; the zero padding carries over).  16 instead of 36
        tap products per source pixel and no 4x intermediate.  Same linear map as generator.py:144-151
        (UpSampling2D then Conv2D); the results differ by fp32 summation order only (measured 3e-6)."""
        conv = self.conv
        w = conv.normalized_weight() if hasattr(conv, 'normalized_weight') else conv.weight
        if tuple(w.shape[2:]) != (3, 3):
            return self.forward(upsample2x(x), _w=w)
        if FAST_CONV and x.is_cuda:                 # the 4x4 kernel is formed inside the weight image (csrc/wc_conv.hip)
            y = fast_conv_mod.fast_conv_or_none(x, w, conv.bias, 'up3', site=self)
            if y is not None:
                return y
        rows = torch.stack([w[:, :, 2], w[:, :, 1] + w[:, :, 2], w[:, :, 0] + w[:, :, 1], w[:, :, 0]], dim=2)
        k = torch.stack([rows[..., 2], rows[..., 1] + rows[..., 2], rows[..., 0] + rows[..., 1], rows[..., 0]], dim=3)
        k = k.transpose(0, 1).contiguous(memory_format=torch.channels_last)             # (Cin, Cout, 4, 4)
        return to_nhwc(F.conv_transpose2d(to_nchw_view(x), k, conv.bias, stride=2, padding=1))


def _conv2d_forward_pooled(self, x, relu_input=False):
    """avg_pool2x2(conv(x)) for a 3x3 'same' convolution as ONE 4x4 stride-2 convolution: the average of the four
    3x3 windows under an output pixel is a 4x4 window whose taps are quarter-sums of the 3x3 taps (zero padding carries
    over, the bias is unchanged) -- 16 instead of 36 tap products per output and no full-resolution intermediate.
    Same linear map as discriminator.py:41-54's Conv2D then AveragePooling2D; differs by fp32 summation order only."""
    conv = self.conv
    w = conv.normalized_weight() if hasattr(conv, 'normalized_weight') else conv.weight
    if FAST_CONV and x.is_cuda and tuple(w.shape[2:]) == (3, 3):
        y = fast_conv_mod.fast_conv_or_none(x, w, conv.bias, 'down3', relu_input=relu_input, site=self)
        if y is not None:
            return y
    if relu_input:
        x = F.relu(x)
    if tuple(w.shape[2:]) != (3, 3):
        return to_nhwc(F.avg_pool2d(to_nchw_view(self.forward(x, _w=w)), 2))
    k = (F.pad(w, (0, 1, 0, 1)) + F.pad(w, (1, 0, 0, 1)) + F.pad(w, (0, 1, 1, 0)) + F.pad(w, (1, 0, 1, 0))) * 0.25
    return to_nhwc(F.conv2d(to_nchw_view(x), k.contiguous(memory_format=torch.channels_last), conv.bias, stride=2, padding=1))


Conv2D.forward_pooled = _conv2d_forward_pooled


def upsample2x(x):
    return to_nhwc(F.interpolate(to_nchw_view(x), scale_factor=2, mode='nearest'))


class _BatchNormNoAffine(nn.Module):
    """norm == 'b': BatchNormalization(center=False, scale=False) (generator.py:22) on NHWC."""

    def __init__(self, name=None):
        super().__init__()
        self.bn = None
        self.layer_name = name

    def forward(self, x):
        if self.bn is None:
            self.bn = nn.BatchNorm2d(x.shape[-1], eps=1e-3, momentum=0.01, affine=False).to(x.device)
        return to_nhwc(self.bn(to_nchw_view(x)))


class _UnfusedStack(nn.Module):
    """norm in {'n','b'}: normalisation then the coloring branches applied on their own and added."""

    def __init__(self, norm_layer, branches):
        super().__init__()
        self.norm_layer = norm_layer
        self.branches = nn.ModuleList(branches)

    def forward(self, x, cls=None):
        if isinstance(x, (list, tuple)):
            x, cls = x
        out = self.norm_layer(x) if self.norm_layer is not None else x
        if len(self.branches) == 0:
            return out
        total = None
        for br in self.branches:
            y = br([out, cls]) if br.conditional else br(out)
            total = y if total is None else total + y
        return total


def create_norm(norm, after_norm, cls=None, number_of_classes=None, filters_emb=10,
                uncoditional_conv_layer=Conv11, conditional_conv_layer=ConditionalConv11,
                factor_conv_layer=FactorizedConv11, process_group=None):
    """Factory of generator.py:13-90: returns result_norm(axis, name) -> stack(inp, cls)."""
    assert norm in NORMS
    assert after_norm in AFTER_NORMS
    K = number_of_classes

    def branches(axis, name, ch):
        if after_norm == 'ccs':
            return [ConditionalCenterScale(number_of_classes=K, axis=axis, name=name, channels=ch)]
        if after_norm == 'ucs':
            return [CenterScale(axis=axis, name=name, channels=ch)]
        if after_norm == 'uccs':
            return [ConditionalCenterScale(number_of_classes=K, axis=axis, name=name + '_c', channels=ch),
                    CenterScale(axis=axis, name=name + '_u', channels=ch)]
        if after_norm == 'cconv':
            return [conditional_conv_layer(number_of_classes=K, name=name, channels=ch)]
        if after_norm == 'fconv':
            return [factor_conv_layer(number_of_classes=K, name=name + '_c', filters_emb=filters_emb, use_bias=False, channels=ch)]
        if after_norm == 'uconv':
            return [uncoditional_conv_layer(kernel_size=(1, 1), name=name, channels=ch)]
        if after_norm == 'ucconv':
            return [conditional_conv_layer(number_of_classes=K, name=name + '_c', channels=ch),
                    uncoditional_conv_layer(kernel_size=(1, 1), name=name + '_u', channels=ch)]
        if after_norm == 'ccsuconv':
            return [ConditionalCenterScale(number_of_classes=K, axis=axis, name=name + '_c', channels=ch),
                    uncoditional_conv_layer(kernel_size=(1, 1), name=name + '_u', channels=ch)]
        if after_norm == 'ufconv':
            return [factor_conv_layer(number_of_classes=K, name=name + '_c', filters_emb=filters_emb, use_bias=False, channels=ch),
                    uncoditional_conv_layer(kernel_size=(1, 1), name=name + '_u', channels=ch)]
        return []                                           # 'n'

    def result_norm(axis, name, channels=None):
        # channels=None keeps the Keras behaviour (build on first call); giving it builds the weights now
        br = branches(axis, name + '_repart', channels)
        if norm in ('d', 'dr'):
            npart = DecorelationNormalization(name=name + '_npart', renorm=(norm == 'dr'), channels=channels,
                                              process_group=process_group)
            return WhiteningColoring(npart, br)
        norm_layer = _BatchNormNoAffine(name=name + '_npart') if norm == 'b' else None
        return _UnfusedStack(norm_layer, br)

    return result_norm


# ---------------------------------------------------------------------------------------------
# generator
# ---------------------------------------------------------------------------------------------
# WC_OVERLAP_SHORTCUT=0: the block's shortcut convolution on the main stream also in forward-only passes
OVERLAP_SHORTCUT = os.environ.get('WC_OVERLAP_SHORTCUT', '1') != '0'
_SIDE_STREAMS = {}


def _side_stream(device):
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _norm_relu(norm, x, cls, consumer=None, kind='same'):
    """relu(norm(x, cls)) (generator.py:144-151, 154); the fused WC stack takes the activation into its apply kernel.
    consumer: the Conv2D that reads the result (and nothing else does) -- where it can, the site's apply kernel then writes that
    convolution's fp16 operand planes directly (no fp32 tensor, no absmax + split passes in front of the convolution)."""
    from .layers import WhiteningColoring
    if isinstance(norm, WhiteningColoring):
        planes = consumer is not None and x.is_cuda and consumer.takes_planes(x.shape, kind)
        return norm(x, cls, relu=True, planes=planes)
    return F.relu(norm(x, cls))


class ResBlockUp(nn.Module):
    def __init__(self, in_ch, nfilters, resample, name, norm, conv_layer):
        super().__init__()
        assert resample in ('UP', 'SAME')
        self.resample = resample
        self.bn1 = norm(axis=-1, name=name + '.bn1', channels=in_ch)
        self.conv1 = conv_layer(in_ch, nfilters, (3, 3), name=name + '.conv1')
        self.bn2 = norm(axis=-1, name=name + '.bn2', channels=nfilters)
        self.conv2 = conv_layer(nfilters, nfilters, (3, 3), name=name + '.conv2')
        self.shortcut = conv_layer(in_ch, nfilters, (1, 1), name=name + '.shortcut')
        # Per-module opt-out of the planes hand-over: with True (default) the block's output may be a HANDLE -- a NaN-valued
        # stride-0 tensor that carries the autograd edge while the data travels as pre-split planes in `_wc_split` -- which is safe only
        # for the readers Generator.forward names (the next block's bn1 and shortcut, the last norm).  Set False on a block whose output
        # something else reads (a forward hook, feature extraction, torch.utils.checkpoint, a custom loop): it then returns the fp32
        # sum.  functional.materialize(handle) converts a handle after the fact.  INTEGRATION.md, "Handles".
        self.split_output = True

    def forward(self, x, cls, readers=()):
        """readers: the modules that read this block's output (the next block's bn1 and shortcut, or the generator's last norm) --
        when each of them has a planes path for the output's shape, the residual add writes pre-split planes instead of fp32."""
        up = self.resample == 'UP' and x.shape[1] * x.shape[2] >= 64
        h = _norm_relu(self.bn1, x, cls, self.conv1 if (up or self.resample != 'UP') else None, 'up3' if up else 'same')
        # the 1x1 shortcut commutes with nearest-neighbour upsampling (every output pixel is the same per-pixel affine
        # map of its source pixel): it runs at the input resolution, a quarter of the work, and is added per 2x2 patch below
        side = None
        if OVERLAP_SHORTCUT and x.is_cuda and not torch.is_grad_enabled():
            # forward-only passes (the generator passes inside the critic updates): the shortcut needs x only, so it runs on a
            # second stream beside the site's narrow stage (K2: a handful of workgroups for 70-80 us) and the block's convolutions
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                s = self.shortcut(x)
        else:
            s = self.shortcut(x)
        if self.resample == 'UP':
            # from 8x8 inputs on, upsample + 3x3 as one 4x4 stride-2 transposed convolution is faster than MIOpen on
            # the 4x tensor (measured forward+backward at N = 128: 3.91 -> 2.02 ms from 16x16, 1.06 -> 0.72 from 8x8,
            # even at 4x4)
            h = self.conv1.forward_upsampled(h) if h.shape[1] * h.shape[2] >= 64 else self.conv1(upsample2x(h))
        else:
            h = self.conv1(h)
        h = _norm_relu(self.bn2, h, cls, self.conv2)
        h = self.conv2(h)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            s.record_stream(torch.cuda.current_stream())
        # the Add that ends the block (generator.py:142-146).  UP: h + upsample2x(s) without the upsampled tensor -- every 2x2
        # output patch adds its one source pixel (csrc/wc_resadd.hip; rounds 1-3: a torch broadcast add)
        if h.is_cuda and h.dtype == torch.float32 and h.shape[-1] % 32 == 0:
            planes = (SPLIT_PRODUCER and self.split_output and len(readers) > 0 and all(r.takes_split(h.shape) for r in readers))
            # the fp32 sum beside the planes: only while a backward will read it (a site whose K4 / K6 have no planes form)
            x32 = planes and torch.is_grad_enabled() and not all(r.backward_takes_split(h.shape) for r in readers
                                                                  if hasattr(r, 'backward_takes_split'))
            # the WC site among the readers, in training mode: the add's pass accumulates its covariance partials too (no K1 launch there)
            sg = max([r.wants_moments(h.shape) for r in readers if hasattr(r, 'wants_moments')] + [0]) if planes else 0
            return residual_add(h, s, self.resample == 'UP', planes=planes, x32=x32, stat_groups=sg)
        if self.resample == 'UP':
            N, H, W, C = s.shape
            return (h.view(N, H, 2, W, 2, C) + s.view(N, H, 1, W, 1, C)).view(N, 2 * H, 2 * W, C)
        return h + s


class DCBlockUp(nn.Module):
    def __init__(self, in_ch, nfilters, resample, name, norm):
        super().__init__()
        self.deconv = nn.ConvTranspose2d(in_ch, nfilters, 4, stride=2 if resample == 'UP' else 1,
                                         padding=1 if resample == 'UP' else 0).to(memory_format=torch.channels_last)
        self.bn = norm(axis=-1, name=name + '.bn', channels=nfilters)

    def forward(self, x, cls):
        h = to_nhwc(self.deconv(to_nchw_view(x)))
        return _norm_relu(self.bn, h, cls)


class Generator(nn.Module):
    def __init__(self, input_noise_shape, output_channels, first_block_shape, block_sizes, resamples,
                 block_norm_layer, last_norm_layer, conv_layer, dense_spectral, concat_cls, number_of_classes, arch,
                 conditional):
        super().__init__()
        self.first_block_shape = tuple(int(v) for v in first_block_shape)
        self.conditional = conditional
        in_dim = int(np.prod(input_noise_shape))
        self.emb = None
        if concat_cls:
            self.emb = nn.Embedding(number_of_classes, self.first_block_shape[-1])
            in_dim += self.first_block_shape[-1]
        self.dense = nn.Linear(in_dim, int(np.prod(self.first_block_shape)))
        nn.init.xavier_uniform_(self.dense.weight); nn.init.zeros_(self.dense.bias)
        if dense_spectral:      # SNDense (generator.py:107-108)
            from .spectral import SNLinear
            sn = SNLinear(self.dense.in_features, self.dense.out_features)
            with torch.no_grad():
                sn.weight.copy_(self.dense.weight); sn.bias.copy_(self.dense.bias)
                sn._sn_init()
            self.dense = sn
        blocks = []
        ch = self.first_block_shape[-1]
        for i, (bs, rs) in enumerate(zip(block_sizes, resamples)):
            bs = int(bs)
            name = 'Generator.' + str(i)
            if arch == 'res':
                blocks.append(ResBlockUp(ch, bs, rs, name, block_norm_layer, conv_layer))
            else:
                blocks.append(DCBlockUp(ch, bs, rs, name, block_norm_layer))
            ch = bs
        self.blocks = nn.ModuleList(blocks)
        self.final_norm = last_norm_layer(axis=-1, name='Generator.BN.Final', channels=ch)
        self.final_conv = conv_layer(ch, output_channels, (3, 3), name='Generator.Final')

    def forward(self, z, cls=None):
        y = z
        if self.emb is not None:
            y = torch.cat([self.emb(cls.reshape(-1).long()), z], dim=-1)
        y = self.dense(y).view(-1, *self.first_block_shape)
        nb = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            if isinstance(blk, ResBlockUp):
                nxt = self.blocks[i + 1] if i + 1 < nb else None
                readers = (self.final_norm,) if nxt is None else ((nxt.bn1, nxt.shortcut) if isinstance(nxt, ResBlockUp) else ())
                readers = tuple(r for r in readers if hasattr(r, 'takes_split'))
                y = blk(y, cls, readers if len(readers) == (1 if nxt is None else 2) else ())
            else:
                y = blk(y, cls)
        y = _norm_relu(self.final_norm, y, cls)
        return torch.tanh(self.final_conv(y))


def make_generator(input_noise_shape=(128,), output_channels=3, input_cls_shape=(1,),
                   block_sizes=(128, 128, 128), resamples=("UP", "UP", "UP"),
                   first_block_shape=(4, 4, 128), number_of_classes=10, concat_cls=False,
                   block_norm='u', block_after_norm='cs', filters_emb=10,
                   last_norm='u', last_after_norm='cs', gan_type=None, arch='res',
                   spectral=False, fully_diff_spectral=False, spectral_iterations=1, conv_singular=True,
                   process_group=None):
    """Same keyword surface as generator.py:93-98; returns an nn.Module called as G(z) or G(z, cls)."""
    assert arch in ['res', 'dcgan']
    if spectral and (block_after_norm not in ('uconv', 'ucs', 'n') or last_after_norm not in ('uconv', 'ucs', 'n')):
        raise NotImplementedError("spectral-normalised conditional coloring (SNConditionalConv11/SNFactorizedConv11) "
                                  "is outside the WC hot path; no shipped recipe sets --generator_spectral")
    conv_layer = partial(Conv2D, spectral=bool(spectral))
    mk = partial(create_norm, number_of_classes=number_of_classes, filters_emb=filters_emb, process_group=process_group)
    block_norm_layer = mk(block_norm, block_after_norm)
    last_norm_layer = mk(last_norm, last_after_norm)
    return Generator(input_noise_shape, output_channels, first_block_shape, block_sizes, resamples,
                     block_norm_layer, last_norm_layer, conv_layer, bool(spectral), concat_cls, number_of_classes, arch,
                     conditional=gan_type is not None)
