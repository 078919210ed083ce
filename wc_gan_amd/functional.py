"""Autograd function for the fused whitening + coloring transform.

Forward  (SURVEY.md rows a2/a3/a6-a9):  y_n = (x_n - mu) A_{slot(n)} + beta_{slot(n)},  A_k = W^T Gamma_k
Backward (row a10): one reduction (K4), the small float64 stage (K5), one two-stream apply (K6).

`process_group` turns on sync-WC: the additive moments (K1 output) and the backward reductions
(K4 output) are all-reduced over RCCL, so every replica whitens with the global-batch statistics.
Default (None) keeps per-replica statistics, which is the reference's behaviour on each GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

import os

from . import _state, ops

# K1 + K2 through the one-call entry wc_whiten_f32 (one launch less; identical results).  WC_WHITEN=0: the two separate calls.
USE_WHITEN = os.environ.get('WC_WHITEN', '1') != '0'
# the ReLU'd backward without a masked copy of the gradient (K4 and K6 both apply the bit mask; WC_BWD_BITS=0: K4 writes the copy)
USE_BWD_BITS = os.environ.get('WC_BWD_BITS', '1') != '0'
# the backward of a site on pre-split planes reads x from the planes too (wc_bwd_reduce_xsplit_f32 / wc_bwd_apply_xsplit_f32); WC_BWD_XSPLIT=0:
# from the fp32 sum the producer then writes beside the planes
USE_BWD_XSPLIT = os.environ.get('WC_BWD_XSPLIT', '1') != '0'
# the residual add accumulates the next site's covariance partials in its own pass (wc_resadd_stats_split_f32; WC_FUSED_STATS=0: the
# site runs its K1 on the planes the add wrote, as in round 4)
USE_FUSED_STATS = os.environ.get('WC_FUSED_STATS', '1') != '0'
# Test hook (tests/test_producer_gpu.py): {'record': []} collects the one-bit ReLU masks the sites of a pass produce, {'replay': [...]} makes
# the sites of the next pass SAVE those instead of their own -- two routes whose K3 outputs differ in the last bit then run their backward
# on identical masks, and their gradients can be compared at rounding level instead of at the level of a few flipped ReLUs.
MASK_TAP = None


def _tap_mask(mask):
    if 'record' in MASK_TAP:
        MASK_TAP['record'].append(mask.clone())
        return mask
    return MASK_TAP['replay'].pop(0)


def _allreduce_(tensors, group):
    """Sum over the replicas, in place.  (General form: pack, reduce, unpack.  The WC path itself hands over tensors that are
    already views of one buffer -- ops.stats / ops.bwd_reduce with flat=True -- and all-reduces that buffer directly.)"""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class WhitenColorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, slot, moving_mean, moving_cov, training, eps, momentum, ddof, process_group, relu=False,
                planes_box=None, st=None):
        # x: (N, ..., C) float32 contiguous (NHWC); gamma (Kc,C,C)|None; beta (Kc,C)|None; slot int32 (N,)|None
        # planes_box: a list -> the output leaves as the next convolution's fp16 planes (conv_handoff below); the box receives them
        # st: x is a HANDLE whose data is this ops.SplitTensor (the residual add wrote the pre-split planes, split_handle below):
        #     K1 and K3 run on the planes (wc_whiten_split_f16x2, wc_apply_split_ex_f16x2), no conversion, no fp32 read
        C = x.shape[-1]
        M_local = x.numel() // C
        dev = x.device
        M = M_local
        mm = moving_mean.view(-1) if moving_mean is not None else None
        if st is None:
            x = x.contiguous()
        if st is not None:
            pre = training and st.moments is not None and st.moments[1] == 1     # the producer accumulated K1's partials in its own pass
            if training and process_group is None:
                mu, L, W = (ops.whiten_presummed if pre else ops.whiten_split)(st, eps, momentum, ddof, mm, moving_cov)
            else:
                if training:       # sync-WC: the additive moments of all replicas, one collective on K1's own buffer
                    s, xtx, buf = (ops.stats_presummed if pre else ops.stats_split)(st, flat=True)
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
                    M = M_local * dist.get_world_size(process_group)
                else:
                    s = xtx = None
                mu, L, W = ops.factor(s, xtx, M, C, eps, momentum, ddof, training, mm, moving_cov, dev)
            chan_scale = st.scale          # the planes' own scales are the apply's input scales
        elif training and process_group is None and USE_WHITEN:
            # per-replica statistics (the reference's behaviour): K1 and K2 as one call -- the moments never leave the workspace
            # and the K1 tail / K2 head run as one launch (wc_whiten_f32; identical results to the two calls below)
            mu, L, W, chan_scale = ops.whiten(x.view(M_local, C), eps, momentum, ddof, mm, moving_cov)
        else:
            if training and process_group is not None:
                # sync-WC: the additive moments of all replicas, ONE collective on the buffer K1 wrote them into (no pack /
                # unpack launches); every replica holds the same number of rows (fixed per-GPU batch): no host sync for the count
                s, xtx, buf = ops.stats(x.view(M_local, C), flat=True)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
                M = M_local * dist.get_world_size(process_group)
            elif training:
                s, xtx = ops.stats(x.view(M_local, C))
            else:
                s = xtx = None
            mu, L, W, chan_scale = ops.factor(s, xtx, M, C, eps, momentum, ddof, training, mm, moving_cov, dev, want_scale=True)
        if training:
            _touched(moving_mean, moving_cov)
        g = gamma.contiguous() if gamma is not None else None
        b = beta.contiguous() if beta is not None else None
        if st is not None:      # ... and, on planes, the additive term beta + (center - mu) A from the same launch as the tables
            A, At, plan, be = ops.color_split(W, g, st, mu, b)
        else:
            A, At, plan = ops.color(W, g, chan_scale)      # plan: the apply's fp16 tables, so K3 is one launch
        # relu: folded into K3's epilogue (row N2).  Its gradient mask is kept as ONE BIT per element (K3 writes it): the
        # backward neither re-reads y (K4: 134 MB at the headline site) nor keeps y alive for it
        bits = bool(relu) and M_local % 32 == 0
        if st is not None:
            if planes_box is not None:
                rec = ops.out_scale(g, b, C, dev)
                out = ops.apply_split(st, None, A, be, slot, plan=plan, relu=relu, folded=True, want_mask=bits, oscale=rec)
                planes_box.append((out[0], out[1], out[2] if bits else None))
                y, mask = _nan_handle(x.shape, dev), planes_box[0][2]
            elif bits:
                y, mask = ops.apply_split(st, None, A, be, slot, plan=plan, relu=True, folded=True, want_mask=True)
            else:
                y, mask = ops.apply_split(st, None, A, be, slot, plan=plan, relu=relu, folded=True), None
            # the backward: K4 / K6 read x from the same planes where they can (wc_bwd_*_xsplit_f32: C = 256 and 128 fast paths); elsewhere
            # from the fp32 sum the producer wrote beside the planes (st.x32), or -- no such copy -- from one made here
            ctx.xsplit = None
            if any(ctx.needs_input_grad[:3]):
                if USE_BWD_XSPLIT and ops.bwd_xsplit_supported(x.shape, slot is not None) and (not relu or bits):
                    ctx.xsplit = (st.shape,)
                    ctx.xs_tensors = (st.planes, st.center, st.scale)
                    x = torch.empty(0, device=dev)
                else:
                    x = st.x32 if st.x32 is not None else ops.unsplit(st)
            else:
                x = torch.empty(0, device=dev)
        elif planes_box is not None:
            y = _apply_planes(x, mu, A, b, slot, plan, g, relu, bits, planes_box)
            mask = planes_box[0][2]
        elif bits:
            y, mask = ops.apply(x, mu, A, b, slot, plan=plan, relu=True, want_mask=True)
        else:
            y, mask = ops.apply(x, mu, A, b, slot, plan=plan, relu=relu), None
        if MASK_TAP is not None and bits:
            mask = _tap_mask(mask)
        xs_t = getattr(ctx, 'xs_tensors', None) or ()
        ctx.xs_tensors = None
        ctx.save_for_backward(x, mu, L, W, A, At, g if g is not None else torch.empty(0, device=dev),
                              slot if slot is not None else torch.empty(0, dtype=torch.int32, device=dev),
                              mask if bits else (y if relu else torch.empty(0, device=dev)), *xs_t)
        if st is None:
            ctx.xsplit = None
        ctx.relu = bool(relu)
        ctx.mask_bits = bits
        ctx.has_gamma = g is not None
        ctx.has_beta = b is not None
        ctx.has_slot = slot is not None
        ctx.training = bool(training)
        ctx.eps, ctx.ddof, ctx.M, ctx.group = eps, ddof, M, process_group
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mu, L, W, A, At, g, slot, y = ctx.saved_tensors[:9]
        xs = None
        if ctx.xsplit is not None:          # x lives in the producer's planes: K4 / K6 read those
            pl, cen, sc = ctx.saved_tensors[9:12]
            xs = ops.SplitTensor(pl, cen, sc, None, ctx.xsplit[0])
        g = g if ctx.has_gamma else None
        slot = slot if ctx.has_slot else None
        gy = gy.contiguous()
        need_x, need_g, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        Kc = A.shape[0]
        dgamma = dbeta = dx = S = gmean = None
        stats_path = ctx.training and need_x
        want_g = ctx.has_gamma and need_g
        want_b = ctx.has_beta and need_b
        reduce_runs = want_g or want_b or stats_path
        if ctx.relu and not reduce_runs:  # the fused activation's gradient: the mask in front of the unchanged backward
            if ctx.mask_bits:
                gy = ops.relu_mask_bits(gy, y)
            else:
                gy = torch.ops.aten.threshold_backward(gy, y, 0.0)      # gy where y > 0, else 0: ONE elementwise pass (where(y > 0, ...) took three launches)
        scales = None          # K4 samples the fp16 scales of (x - mu) and gy; K6 reuses them (three launches instead of six)
        k6_mask = None
        if reduce_runs:
            share = bool(stats_path)
            # K4 applies the mask while it stages gy and hands the masked gradient on (no pass of its own)
            ry = y if (ctx.relu and not ctx.mask_bits) else None
            rm = y if (ctx.relu and ctx.mask_bits) else None          # (the saved tensor is the bit mask then)
            # with the bits and a K6 that masks for itself (C = 256 fast paths) K4 writes no masked copy of the gradient at all
            bits_only = rm is not None and share and need_x and USE_BWD_BITS and ops.bwd_bits_supported(x.shape, slot is not None)
            if xs is not None:
                if rm is not None and gy.shape[-1] != 256:      # the planes forms of K4 / K6 apply the bits themselves at C = 256 only: one masking pass in front
                    gy, rm = ops.relu_mask_bits(gy, rm), None
                out = ops.bwd_reduce_xsplit(xs, mu, gy, slot, Kc, relu_mask=rm, flat=ctx.group is not None)
                R, gsum = out[0], out[1]
                rbuf = out[2] if ctx.group is not None else None
                scales, k6_mask = out[-1], rm
            else:
                if ctx.group is None:
                    out = ops.bwd_reduce(x, mu, gy, slot, Kc, want_scales=share, relu_y=ry, relu_mask=rm, write_masked=not bits_only)
                    R, gsum = out[0], out[1]
                else:
                    out = ops.bwd_reduce(x, mu, gy, slot, Kc, flat=True, want_scales=share, relu_y=ry, relu_mask=rm, write_masked=not bits_only)
                    R, gsum, rbuf = out[0], out[1], out[2]
                if share:
                    scales = out[-1]
                if bits_only:
                    k6_mask = rm
                elif ry is not None or rm is not None:
                    gy = out[-2] if share else out[-1]
            if ctx.group is None:
                dgamma, dbeta, S, gmean = ops.bwd_factor(R, gsum, W, L, g, A, ctx.M, ctx.eps, ctx.ddof, stats_path,
                                                         want_dgamma=want_g, want_dbeta=want_b)
            else:
                # parameter gradients stay per-replica (the DDP-style average happens outside);
                # the statistics path needs the global reductions under sync-WC
                if want_g or want_b:
                    dgamma, dbeta, _, _ = ops.bwd_factor(R, gsum, W, L, g, A, ctx.M, ctx.eps, ctx.ddof, False,
                                                         want_dgamma=want_g, want_dbeta=want_b)
                if stats_path:
                    dist.all_reduce(rbuf, op=dist.ReduceOp.SUM, group=ctx.group)
                    _, _, S, gmean = ops.bwd_factor(R, gsum, W, L, g, A, ctx.M, ctx.eps, ctx.ddof, True,
                                                    want_dgamma=False, want_dbeta=False)
        if need_x:
            if xs is not None and S is not None and scales is not None:
                dx = ops.bwd_apply_xsplit(gy, xs, mu, At, S, gmean, slot, scales, relu_mask=k6_mask)
            else:
                if xs is not None and k6_mask is not None:          # (evaluation-mode site with a gradient: no statistics path, dx = masked gy At)
                    gy, k6_mask = ops.relu_mask_bits(gy, k6_mask), None
                dx = ops.bwd_apply(gy, x if xs is None else None, mu if xs is None else None, At, S if xs is None else None,
                                   gmean if xs is None else None, slot, scales=scales if xs is None else None, relu_mask=k6_mask)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# K3 -> convolution hand-off: the site's output as the next convolution's operand
# ---------------------------------------------------------------------------------------------
_NAN = {}


def _nan_handle(shape, dev):
    """The tensor that stands for y in the autograd graph when y itself leaves as fp16 planes: the right shape and dtype, no
    memory (one NaN element, stride 0) -- anything that reads it as data fails loudly instead of computing on zeros."""
    t = _NAN.get(str(dev))
    if t is None:
        t = torch.full((1,), float('nan'), dtype=torch.float32, device=dev)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _NAN[str(dev)] = t
    return t.expand(tuple(shape))


def conv_handoff_supported(shape, relu, Ktables=1):
    """May a site of this NHWC output shape hand its output to the next convolution as planes? (relu'd sites only: that is what
    every convolution behind a WC site reads, generator.py:144-151)"""
    return bool(relu) and Ktables <= 1024 and ops.apply_planes_supported(tuple(shape))


def _apply_planes(x, mu, A, b, slot, plan, gamma, relu, want_mask, box, beta=False):
    # beta: the coloring's own bias where `b` is an effective one (grouped batches: b also carries the groups' mean offsets)
    rec = ops.out_scale(gamma, b if beta is False else beta, x.shape[-1], x.device)
    out = ops.apply_planes(x, mu, A, b, slot, plan, rec, relu=relu, want_mask=want_mask)
    box.append((out[0], out[1], out[2] if want_mask else None))
    return _nan_handle(x.shape, x.device)


def attach_planes(handle, box):
    """handle._wc_planes = (hi, lo, scale record): what conv.fast_conv_or_none looks for on its input."""
    both, rec, _ = box[0]
    handle._wc_planes = (both[0], both[1], rec)
    return handle


def split_of(x):
    """The ops.SplitTensor a handle carries (the residual add of the block in front wrote the tensor as pre-split planes), or None."""
    return getattr(x, '_wc_split', None)


def materialize(x):
    """The fp32 tensor behind a handle (no autograd): a block output that travels as pre-split planes (`_wc_split`: ops.unsplit) or a site
    output that travels as the next convolution's planes (`_wc_planes`: (hi + lo) / scale); any other tensor is returned as it is.  For
    readers outside the generator's own wiring -- hooks, feature extraction, debugging -- which would otherwise compute on the handle's NaN."""
    st = getattr(x, '_wc_split', None)
    if st is not None:
        return ops.unsplit(st)
    pl = getattr(x, '_wc_planes', None)
    if pl is not None:
        hi, lo, rec = pl
        return ((hi.float() + lo.float()) / rec[0]).view(x.shape)
    return x


def split_handle(st, shape, dev):
    """A tensor that stands for a pre-split activation wherever a tensor object is needed (shape, device, autograd edge): the NaN
    handle of the K3 -> convolution hand-off, with the data attached as `_wc_split`."""
    h = _nan_handle(shape, dev)
    h._wc_split = st
    return h


class ResidualAddFunction(torch.autograd.Function):
    """out = h + upsample2x(s) (up) or h + s: the Add that ends a `resblock` (generator.py:142-146), csrc/wc_resadd.hip.
    box is None: the fp32 sum.  box a list: the sum leaves in the pre-split format for the next WC site (and the next shortcut
    convolution) -- the result is a handle and the SplitTensor lands in the box (with .x32 when a backward will read fp32)."""

    @staticmethod
    def forward(ctx, h, s, up, box, x32, stat_groups):
        h = h.contiguous(); s = s.contiguous()
        ctx.up = bool(up)
        if box is None:
            return ops.resadd(h, s, up)
        want32 = bool(x32) and any(ctx.needs_input_grad[:2])
        if stat_groups and USE_FUSED_STATS and ops.resadd_stats_supported(h.shape, up, stat_groups):
            st = ops.resadd_stats_split(h, s, up, stat_groups, want_x32=want32)      # ... and K1's partials from the same pass
        else:
            st = ops.resadd_split(h, s, up, want_x32=want32)
        box.append(st)
        return _nan_handle(h.shape, h.device)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g, (ops.patch_sum(g) if ctx.up else g), None, None, None, None


def residual_add(h, s, up, planes=False, x32=True, stat_groups=0):
    """h + (upsample2x of) s.  planes=True (the readers of the sum all have a planes path: layers.WhiteningColoring.takes_split,
    generator.Conv2D.takes_split): a handle carrying the sum as pre-split planes (`split_of(handle)`).  x32=False: no reader's backward
    needs the fp32 sum either (layers.WhiteningColoring.backward_takes_split) -- else it is written beside the planes while a gradient
    is wanted.  stat_groups > 0: the WC site that reads the sum is in training mode with that many statistic groups -- the add's pass
    then accumulates that site's covariance partials as well (ops.resadd_stats_split: the site's K1 launch does not exist)."""
    if planes and ops.resadd_split_supported(h.shape):
        box = []
        out = ResidualAddFunction.apply(h, s, bool(up), box, bool(x32), int(stat_groups))
        out._wc_split = box[0]
        return out
    return ResidualAddFunction.apply(h, s, bool(up), None, False, 0)


_SLOT_BASE = {}


def _group_slot_base(N, groups, Kc, dev):
    """int32 (N,): group(n) * Kc, the table index of sample n before its class slot is added.  It depends on the shapes
    only, and building it took four elementwise launches per site and step (arange, //, *, cast: ~20 us of the grouped
    forward site), so it is kept per (N, groups, Kc, device) -- except while a graph is being recorded, whose private pool
    must not hand memory to later eager calls."""
    key = (N, groups, Kc, str(dev))
    t = _SLOT_BASE.get(key)
    if t is None:
        t = ((torch.arange(N, device=dev, dtype=torch.int32) // (N // groups)) * Kc).to(torch.int32).contiguous()
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _SLOT_BASE[key] = t
    return t


def whiten_color_grouped(x, groups, gamma=None, beta=None, slot=None, moving_mean=None, moving_cov=None,
                         eps=1e-3, momentum=0.99, ddof=1, relu=False, per_sample=False, planes=False):
    """Training-mode forward of `groups` INDEPENDENT batches stacked along N (no autograd): each run of N/groups
    samples is whitened with its own batch statistics, exactly as `groups` separate calls would be, but the
    covariance / Cholesky / inverse problems of the groups are solved side by side in one set of launches.
    Used for the generator passes inside the critic updates (fixed generator weights, no graph).
    per_sample: gamma (N, C, C) / beta (N, C) hold one coloring table per SAMPLE (more classes than samples per batch,
    layers.WhiteningColoring.coloring_table); sample n of group g is coloured by W_g^T gamma[n]."""
    N, C = x.shape[0], x.shape[-1]
    if N % groups != 0:
        raise ValueError("N must be a multiple of groups")
    st = split_of(x)                    # the residual add in front wrote pre-split planes: K1 / K3 read those
    x = x.detach() if st is not None else x.detach().contiguous()
    M = x.numel() // C
    Mg = M // groups
    dev = x.device
    mm = moving_mean.view(-1) if moving_mean is not None else None
    if st is not None:
        # K1 + K2 (wc_whiten_split_f16x2) -- or, where the residual add accumulated K1's partials itself, the tail + K2 only
        pre = st.moments is not None and st.moments[1] == groups
        mu, L, W = (ops.whiten_presummed if pre else ops.whiten_split)(st, eps, momentum, ddof, mm, moving_cov, groups)
        cs = st.scale
    elif USE_WHITEN:
        mu, L, W, cs = ops.whiten(x.view(M, C), eps, momentum, ddof, mm, moving_cov, groups)      # K1 + K2 (wc_whiten_f32)
    else:
        s, xtx = ops.stats(x.view(M, C), groups)
        mu, L, W, cs = ops.factor(s, xtx, Mg, C, eps, momentum, ddof, True, mm, moving_cov, dev, want_scale=True, groups=groups)
    _touched(moving_mean, moving_cov)
    g = gamma.detach().contiguous() if gamma is not None else None
    b = beta.detach().contiguous() if beta is not None else None
    def finish(center, A, bias, slots, plan):
        # (the predicted scale follows from the coloring tables as given: every group's whitened batch has unit covariance)
        handoff = planes and conv_handoff_supported(x.shape, relu, 1 if g is None else g.shape[0])
        if st is not None:
            be = bias            # (already beta - (mu_g - st.center) A: group_bias_centered)
            if handoff:
                rec = ops.out_scale(g, b, C, dev)
                both, rec = ops.apply_split(st, None, A, be, slots, plan=plan, relu=relu, folded=True, oscale=rec)
                return attach_planes(_nan_handle(x.shape, dev), [(both, rec, None)])
            return ops.apply_split(st, None, A, be, slots, plan=plan, relu=relu, folded=True)
        if handoff:
            box = []
            return attach_planes(_apply_planes(x, center, A, bias, slots, plan, g, relu, False, box, beta=b), box)
        return ops.apply(x, center, A, bias, slots, plan=plan, relu=relu)

    def gbias(A, Kc, per_group):
        # on planes the common centre is the planes' own: the biases are then the additive terms of the split apply directly
        if st is not None:
            return st.center, ops.group_bias_centered(mu.view(groups, C), A, b, st.center, groups, Kc, per_group=per_group)
        return ops.group_bias(mu.view(groups, C), A, b, groups, Kc, per_group=per_group)

    if per_sample:
        if g is None or g.shape[0] != N:
            raise ValueError("per_sample needs one coloring table per sample")
        Kc = N // groups
        A, At, plan = ops.color(W, g, cs, groups, per_group=True)
        center, bias = gbias(A, Kc, True)
        return finish(center, A, bias, _group_slot_base(N, N, 1, dev), plan)
    Kc = 1 if g is None else g.shape[0]
    A, At, plan = ops.color(W, g, cs, groups)
    center, bias = gbias(A, Kc, False)
    full_slot = _group_slot_base(N, groups, Kc, dev)
    if slot is not None:
        full_slot = (full_slot + slot.view(-1)).to(torch.int32).contiguous()
    return finish(center, A, bias, full_slot, plan)


def _touched(*tensors):
    """The HIP stages update the moving statistics through raw pointers: tell torch (version counters), so that whatever is
    cached per version -- the eval-mode plan below -- sees the update."""
    inc = getattr(torch.autograd.graph, 'increment_version', None)
    if inc is not None:
        for t in tensors:
            if t is not None:
                inc(t)


class EvalPlan:
    """Eval-mode cache: the moving statistics are constants between weight updates, so
    mu, W = chol((1-eps) moving_cov + eps I)^-1, A = W^T Gamma and the apply plan are computed once and reused --
    the reference redoes the Cholesky on every scorer.py call.  Invalidated by any in-place change of the inputs."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, C, gamma, moving_mean, moving_cov, eps, dev, gamma_key=None):
        # gamma is usually rebuilt from the coloring weights on every call: key on those weights (gamma_key) when given
        gk = gamma_key if gamma_key is not None else (None if gamma is None else (gamma.data_ptr(), gamma._version))
        key = (_state.replays, C, eps, moving_mean._version, moving_cov._version, moving_mean.data_ptr(), moving_cov.data_ptr(),
               None if gamma is None else tuple(gamma.shape), gk)
        if key != self.key:
            with torch.no_grad():
                mu, L, W, cs = ops.factor(None, None, 1, C, eps, 0.0, 1, False, moving_mean.view(-1), moving_cov, dev,
                                          want_scale=True)
                g = gamma.detach().contiguous() if gamma is not None else None
                A, At, plan = ops.color(W, g, cs)
            self.key, self.val = key, (mu, A, At, plan)
        return self.val


def whiten_color_eval_cached(x, cache, gamma=None, beta=None, slot=None, moving_mean=None, moving_cov=None, eps=1e-3,
                             gamma_key=None, relu=False, planes=False):
    """Inference forward (no autograd) through an EvalPlan: one K3 launch per call once the plan is warm."""
    C = x.shape[-1]
    mu, A, At, plan = cache.get(C, gamma, moving_mean, moving_cov, eps, x.device, gamma_key)
    b = beta.detach().contiguous() if beta is not None else None
    st = split_of(x)
    handoff = planes and conv_handoff_supported(x.shape, relu, A.shape[0])
    g = gamma.detach().contiguous() if (gamma is not None and handoff) else None
    if st is not None:
        # on planes: the cached A, with the tables for THIS tensor's scales and the additive term beta + (center - mu) A built
        # inside the call (three launches; the planes' scales come from a sample of the data, not from the cached statistics)
        if handoff:
            rec = ops.out_scale(g, b, C, x.device)
            both, rec = ops.apply_split(st, mu, A, b, slot, relu=relu, oscale=rec)
            return attach_planes(_nan_handle(x.shape, x.device), [(both, rec, None)])
        return ops.apply_split(st, mu, A, b, slot, relu=relu)
    x = x.detach().contiguous()
    if handoff:
        box = []
        return attach_planes(_apply_planes(x, mu, A, b, slot, plan, g, relu, False, box), box)
    return ops.apply(x, mu, A, b, slot, plan=plan, relu=relu)


def whiten_color(x, gamma=None, beta=None, slot=None, moving_mean=None, moving_cov=None, training=True,
                 eps=1e-3, momentum=0.99, ddof=1, process_group=None, relu=False, planes=False):
    """y = coloring(whitening(x)) (relu=True: max(y, 0) from the same kernel).  x: (N, H, W, C) float32 on the GPU,
    C % 32 == 0 (see layers for padding).  planes=True (relu'd sites whose consumer is conv.fast_conv): where K3 can, the
    result is a HANDLE -- a NaN tensor of y's shape without memory that carries the autograd edge -- with the output itself
    attached as the convolution's fp16 planes (handle._wc_planes); else the plain tensor."""
    Kt = 1 if gamma is None else gamma.shape[0]
    st = split_of(x)
    if planes and conv_handoff_supported(x.shape, relu, Kt):
        box = []
        h = WhitenColorFunction.apply(x, gamma, beta, slot, moving_mean, moving_cov, bool(training),
                                      float(eps), float(momentum), int(ddof), process_group, True, box, st)
        return attach_planes(h, box)
    return WhitenColorFunction.apply(x, gamma, beta, slot, moving_mean, moving_cov, bool(training),
                                     float(eps), float(momentum), int(ddof), process_group, bool(relu), None, st)


_ROUTE = {}


def split_route_supported(shape, training, groups=1):
    """Can a WC site of this NHWC input shape read its input as pre-split planes (K1: wc_whiten_split_f16x2 in training mode, K3:
    wc_apply_split_ex_f16x2)?  Shapes only (cached: the producer asks on every pass)."""
    key = (tuple(shape), bool(training), int(groups))
    r = _ROUTE.get(key)
    if r is None:
        C = shape[-1]
        M = 1
        for d in shape[:-1]:
            M *= d
        r = ops.apply_split_supported(tuple(shape)) and ((not training) or ops.stats_split_supported(M, C, groups))
        _ROUTE[key] = r
    return r


# ---------------------------------------------------------------------------------------------
# Modular pieces: the same HIP kernels exposed as two differentiable ops, so that a C x C stage
# written in torch (ZCA's eigendecomposition, renorm's constant factor) can sit between them.
# ---------------------------------------------------------------------------------------------
class FactorMixFunction(torch.autograd.Function):
    """Tables of the soft-assignment coloring (generator.py:69-78): out[t] = base + sum_e alpha[idx[t], e] dictionary[e] through
    wc_factor_mix_f32 / wc_factor_mix_bwd_f32 -- only the tables the batch uses, one launch forward, two to three backward."""

    @staticmethod
    def forward(ctx, dictionary, alpha, idx, base):
        dictionary, alpha = dictionary.contiguous(), alpha.contiguous()
        base = None if base is None else base.contiguous()
        ctx.save_for_backward(dictionary, alpha, idx)
        ctx.has_base = base is not None
        return ops.factor_mix(dictionary, alpha, idx, base)

    @staticmethod
    def backward(ctx, dout):
        dictionary, alpha, idx = ctx.saved_tensors
        dd, da, db = ops.factor_mix_bwd(dictionary, alpha, idx, dout.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                        ctx.has_base and ctx.needs_input_grad[3])
        return dd, da, None, db


def factor_mix(dictionary, alpha, idx=None, base=None):
    return FactorMixFunction.apply(dictionary, alpha, idx, base)


class MomentsFunction(torch.autograd.Function):
    """(sum, xtx) = K1(x).  backward: dx[m] = gsum + x[m] (gxtx + gxtx^T)  -- one K3 launch."""

    @staticmethod
    def forward(ctx, x):
        C = x.shape[-1]
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.stats(x.view(-1, C))

    @staticmethod
    def backward(ctx, gs, gxtx):
        (x,) = ctx.saved_tensors
        C = x.shape[-1]
        A = (gxtx + gxtx.t()).to(torch.float32).reshape(1, C, C).contiguous()
        b = gs.to(torch.float32).reshape(1, C).contiguous()
        return ops.apply(x, None, A, b, None)


class AffineRowsFunction(torch.autograd.Function):
    """y[n] = (x[n] - mu) A[slot[n]] + b[slot[n]] with gradients to x, mu, A and b (K3 / K4 / K6)."""

    @staticmethod
    def forward(ctx, x, mu, A, b, slot):
        x = x.contiguous()
        A = A.contiguous()
        mu_c = mu.contiguous() if mu is not None else None
        b_c = b.contiguous() if b is not None else None
        ctx.save_for_backward(x, A, mu_c if mu_c is not None else torch.empty(0, device=x.device),
                              slot if slot is not None else torch.empty(0, dtype=torch.int32, device=x.device))
        ctx.has_mu, ctx.has_b, ctx.has_slot = mu is not None, b is not None, slot is not None
        return ops.apply(x, mu_c, A, b_c, slot)

    @staticmethod
    def backward(ctx, gy):
        x, A, mu, slot = ctx.saved_tensors
        mu = mu if ctx.has_mu else None
        slot = slot if ctx.has_slot else None
        gy = gy.contiguous()
        Kc = A.shape[0]
        need_x, need_mu, need_A, need_b = ctx.needs_input_grad[:4]
        dx = dmu = dA = db = None
        At = A.transpose(1, 2).contiguous()
        if need_x:
            dx = ops.bwd_apply(gy, None, None, At, None, None, slot)
        if need_A or need_b or need_mu:
            R, gsum = ops.bwd_reduce(x, mu, gy, slot, Kc)
            if need_A:
                dA = R.to(torch.float32)
            if need_b and ctx.has_b:
                db = gsum.to(torch.float32)
            if need_mu and ctx.has_mu:
                dmu = -torch.einsum('kj,kcj->c', gsum, A.to(torch.float64)).to(torch.float32)
        return dx, dmu, dA, db, None


def whiten_color_modular(x, gamma=None, beta=None, slot=None, moving_mean=None, moving_cov=None, training=True,
                         eps=1e-3, momentum=0.99, ddof=1, decomposition='zca'):
    """Unfused composition moments -> torch C x C stage -> affine, for decompositions without a fused kernel.

    decomposition='zca' (generator.py:24, commented alternative): W = U diag(S^-1/2) U^T of Sigma + eps I;
    torch.linalg.eigh supplies the (reportedly unstable) gradient, as tf.svd did upstream.
    """
    C = x.shape[-1]
    M = x.numel() // C
    if training:
        s, xtx = MomentsFunction.apply(x)
        mu64 = s / M
        sigma = (xtx - torch.outer(s, s) / M) / (M - ddof)
        sigma = 0.5 * (sigma + sigma.t())
        if moving_mean is not None:
            with torch.no_grad():
                moving_mean.mul_(momentum).add_((1 - momentum) * mu64.to(torch.float32).view_as(moving_mean))
                moving_cov.mul_(momentum).add_((1 - momentum) * sigma.to(torch.float32))
    else:
        mu64 = moving_mean.view(-1).to(torch.float64)
        sigma = moving_cov.to(torch.float64)
    eye = torch.eye(C, dtype=torch.float64, device=x.device)
    if decomposition == 'zca':
        S, U = torch.linalg.eigh(sigma + eps * eye)
        W = (U * S.rsqrt()) @ U.t()
    elif decomposition == 'cholesky':
        L = torch.linalg.cholesky((1 - eps) * sigma + eps * eye)
        W = torch.linalg.solve_triangular(L, eye, upper=False)
    else:
        raise ValueError(f"unknown decomposition {decomposition!r}")
    if gamma is None:
        A = W.t().unsqueeze(0)
    else:
        A = torch.matmul(W.t().unsqueeze(0), gamma.to(torch.float64))
    return AffineRowsFunction.apply(x, mu64.to(torch.float32), A.to(torch.float32), beta, slot)
