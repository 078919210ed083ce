"""Convolutions of the residual blocks either side of the WC sites on the split-fp16 MFMA path (csrc/wc_conv.hip).

`fast_conv(x, w, bias, kind)` on NHWC tensors:

    kind 'same'  Keras `Conv2D(k x k, padding='same')`, w (Cout, Cin, k, k)          generator.py:142-158
    kind 'down'  4x4 stride-2 padding-1 convolution, w (Cout, Cin, 4, 4)              (= Conv2D 3x3 -> AveragePooling2D, discriminator.py:41-54)
    kind 'up'    4x4 stride-2 padding-1 TRANSPOSED convolution, w (Cin, Cout, 4, 4)   (= UpSampling2D -> Conv2D 3x3, generator.py:144-151)
    kind 'down3' / 'up3'  the same two layers given the 3x3 weight (Cout, Cin, 3, 3) itself: the 4x4 kernels are sums of 3x3
                 taps, formed while the weight image is built and folded back in the weight gradient

Forward and the data gradient run on the HIP kernel (the data gradient of 'same' is a 'same', of 'down' an 'up' and of
'up' a 'down', with the channel roles swapped in the weight image), and so does the weight gradient (pixel-major tiles
read back through gfx950's transposing LDS read).  No fallback inside: `supported(...)` tells the caller whether the kernel takes a shape.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, _state
from .ops import _ptr, _stream

_zero_lines = {}


def _zero_line(device):
    z = _zero_lines.get(device)
    if z is None:
        z = _zero_lines[device] = torch.zeros(64, dtype=torch.uint8, device=device)
    return z


def _plain(r, s):
    return [(r, s)], 1.0


def _pooled3(r, s):
    """4x4 stride-2 kernel of Conv2D 3x3 -> AveragePooling2D: tap (r, s) = 1/4 of the 3x3 taps (r - a, s - b), a, b in {0, 1}"""
    return [(r - a, s - b) for a in (0, 1) for b in (0, 1) if 0 <= r - a <= 2 and 0 <= s - b <= 2], 0.25


_UP_ROWS = ((2,), (1, 2), (0, 1), (0,))


def _upsampled3(r, s):
    """4x4 stride-2 transposed kernel of UpSampling2D -> Conv2D 3x3: rows [w2, w1 + w2, w0 + w1, w0] along each axis"""
    return [(r3, s3) for r3 in _UP_ROWS[r] for s3 in _UP_ROWS[s]], 1.0


def _set_sources(g, p, t, r, s, source):
    taps, coef = source(r, s)
    g.nsrc[p][t] = len(taps)
    for m, (r3, s3) in enumerate(taps):
        g.wr[p][t][m], g.ws[p][t][m] = r3, s3
    g.wcoef = coef


def _dense_geom(N, Hin, Win, Cin, Cout, k, stride, source=_plain):
    """taps (r, s) read input (y*stride + r - pad, x*stride + s - pad); pad = k//2 for 'same', 1 for the 4x4 stride-2 form"""
    pad = k // 2 if stride == 1 else 1
    g = _lib.ConvGeom()
    g.N, g.Hin, g.Win, g.Cin, g.Cout = N, Hin, Win, Cin, Cout
    g.H, g.W = Hin // stride, Win // stride
    g.Hout, g.Wout = g.H, g.W
    g.in_stride, g.out_stride, g.ntaps, g.nphase = stride, 1, k * k, 1
    for r in range(k):
        for s in range(k):
            t = r * k + s
            g.dy[0][t], g.dx[0][t] = r - pad, s - pad
            _set_sources(g, 0, t, r, s, source)
    return g


def _phase_geom(N, Hin, Win, Cin, Cout, source=_plain):
    """4x4 stride-2 padding-1 transposed convolution as four 2x2 sub-pixel convolutions: output (2y+py, 2x+px) reads
    input (y+dy, x+dx) through weight tap r = py + 1 - 2 dy"""
    g = _lib.ConvGeom()
    g.N, g.Hin, g.Win, g.Cin, g.Cout = N, Hin, Win, Cin, Cout
    g.H, g.W, g.Hout, g.Wout = Hin, Win, 2 * Hin, 2 * Win
    g.in_stride, g.out_stride, g.ntaps, g.nphase = 1, 2, 4, 4
    for py in range(2):
        for px in range(2):
            p = py * 2 + px
            g.off_y[p], g.off_x[p] = py, px
            t = 0
            for dy in ((-1, 0) if py == 0 else (0, 1)):
                for dx in ((-1, 0) if px == 0 else (0, 1)):
                    g.dy[p][t], g.dx[p][t] = dy, dx
                    _set_sources(g, p, t, py + 1 - 2 * dy, px + 1 - 2 * dx, source)
                    t += 1
    return g


def _supported(g):
    return bool(_lib.load().wc_conv_supported(ctypes.addressof(g)))


def supported(x, w, kind):
    """Does the kernel take this call?  (channels multiples of 128, N*H*W of the virtual grid a multiple of 128)"""
    handed = getattr(x, '_wc_planes', None) is not None       # a K3 handle: the data is in the planes it carries
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (handed or x.is_contiguous()) and w.dtype == torch.float32):
        return False
    p = _plan(kind, x, w)
    return bool(p) and p.ok


class _Plan:
    """Everything about a call that depends on the shapes only, built once: the two geometries, the weight axes, the
    support verdict and the workspace sizes (the per-call host work is what is left: a few allocations and launches)."""
    __slots__ = ('ok', 'fwd', 'bwd', 'fwd_ws', 'bwd_ws', 'wrw_ws', 'fwd_ptr', 'bwd_ptr')

    def __init__(self, kind, N, H, W, wshape):
        class _W:                                   # shape-only stand-in for the weight
            shape = wshape
        lib = _lib.load()
        (gf, kf, nf), (gb, kb, nb) = _geoms(kind, N, H, W, _W)
        self.fwd, self.bwd = (gf, kf, nf), (gb, kb, nb)
        self.fwd_ptr, self.bwd_ptr = ctypes.addressof(gf), ctypes.addressof(gb)
        self.ok = bool(lib.wc_conv_supported(self.fwd_ptr)) and bool(lib.wc_conv_supported(self.bwd_ptr))
        self.fwd_ws = lib.wc_conv_workspace_bytes(self.fwd_ptr) if self.ok else 0
        self.bwd_ws = lib.wc_conv_workspace_bytes(self.bwd_ptr) if self.ok else 0
        self.wrw_ws = lib.wc_conv_wrw_workspace_bytes(self.fwd_ptr) if self.ok else 0


_plans = {}


def _plan(kind, x, w):
    key = (kind, tuple(x.shape), tuple(w.shape))
    p = _plans.get(key)
    if p is None:
        N, H, W, C = x.shape
        good = (kind == 'same' and w.shape[1] == C and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3)) or \
               (kind == 'down' and w.shape[1] == C and tuple(w.shape[2:]) == (4, 4) and H % 2 == 0 and W % 2 == 0) or \
               (kind == 'down3' and w.shape[1] == C and tuple(w.shape[2:]) == (3, 3) and H % 2 == 0 and W % 2 == 0) or \
               (kind == 'up3' and w.shape[1] == C and tuple(w.shape[2:]) == (3, 3)) or \
               (kind == 'up' and w.shape[0] == C and tuple(w.shape[2:]) == (4, 4))
        if kind not in ('same', 'down', 'up', 'down3', 'up3'):
            raise ValueError(kind)
        p = _plans[key] = _Plan(kind, N, H, W, tuple(w.shape)) if good else False
    return p


def _colsum_ok(C):
    return C % 4 == 0 and 256 % (C // 4) == 0


# The split's scale from the call before at the same call site (wc_conv_split_hist_f32: one launch instead of absmax + split; nothing clamps,
# an element that does not fit goes inf = loud).  Training-mode layers only: an eval-mode pass measures every tensor, so that the images of a
# loaded checkpoint do not depend on what the process ran before.  WC_SPLIT_HIST=0: the measured maximum everywhere, two launches (rounds 1-4).
SPLIT_HIST = os.environ.get('WC_SPLIT_HIST', '1') != '0'
HIST_FLOATS = 4 * 512 + 16     # include/wc_hip.h WC_CONV_HIST_FLOATS
HIST_REDO = 4 * 512            # WC_CONV_HIST_REDO: uint32 count of the gated second passes a site has taken


# The gated second pass of the history-scaled split (include/wc_hip.h, wc_conv_split_hist_f32): WC_SPLIT_HIST_REDO = 1 (default) for the
# output gradients only, 2 for every split, 0 for none.  (Every split of the step: +0.56 ms of 18 -- what the history saves.)
_REDO_MODE = os.environ.get('WC_SPLIT_HIST_REDO', '1')


def _guarded(role):
    return _REDO_MODE == '2' or (_REDO_MODE != '0' and role == 'g')


def _site_hist(site, role, device):
    """[record (HIST_FLOATS floats on the device: two arrays of per-workgroup (maximum, tag) pairs), seeded?] of a call site:
    `site` is the layer object that owns the convolution (state lives in its __dict__, not in a parameter or buffer: no checkpoint entry --
    a resumed run measures once), role 'x' (its input) or 'g' (its output gradient).  None for a layer in eval mode, and while a hipGraph is
    being recorded for a site that has no record yet (no allocation into a graph's private pool; the trainers warm up eagerly first)."""
    if not getattr(site, 'training', True):
        return None
    book = site.__dict__.setdefault('_wc_split_hist', {})
    h = book.get(role)
    if h is not None and h[0].device != device:     # the layer moved to another device: a fresh record there (its first call measures)
        h = None
    if h is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        h = book[role] = [torch.zeros(HIST_FLOATS, dtype=torch.float32, device=device), False]
    return h


def split_planes(x, relu=False, colsum=False, site=None, role='x'):
    """fp32 tensor -> (hi, lo, scale): fp16 planes of s*x and the device scalar s.  colsum: also the 512 partial rows of the
    column sums over the last axis (-> the bias gradient, finished by weight_gradient), returned as a 4th element.
    site (+ role): the layer object this tensor belongs to -- the scale then comes from the previous call of that site (one launch)."""
    lib = _lib.load()
    both = torch.empty((2,) + tuple(x.shape), dtype=torch.float16, device=x.device)
    hi, lo = both[0], both[1]
    scale = torch.empty(1 + 512, dtype=torch.float32, device=x.device)    # [scale | per-workgroup maxima scratch]
    h = _site_hist(site, role, x.device) if (SPLIT_HIST and site is not None) else None
    if h is not None:
        C = x.shape[-1]
        part = torch.empty((512, C), dtype=torch.float32, device=x.device) if colsum else None
        _lib.check(lib.wc_conv_split_hist_f32(_ptr(x), x.numel(), 1 if relu else 0, _ptr(hi), _ptr(lo), _ptr(scale), _ptr(part), C if colsum else 0,
                                              _ptr(h[0]), (0 if h[1] else 1) | (0 if _guarded(role) else 2), _stream()), "wc_conv_split_hist_f32")
        h[1] = True
        return (hi, lo, scale, part) if colsum else (hi, lo, scale)
    if not colsum:
        _lib.check(lib.wc_conv_split_f32(_ptr(x), x.numel(), 1 if relu else 0, _ptr(hi), _ptr(lo), _ptr(scale),
                                         scale.data_ptr() + 4, _stream()), "wc_conv_split_f32")
        return hi, lo, scale
    C = x.shape[-1]
    part = torch.empty((512, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.wc_conv_split_colsum_f32(_ptr(x), x.numel(), 1 if relu else 0, _ptr(hi), _ptr(lo), _ptr(scale),
                                            scale.data_ptr() + 4, _ptr(part), C, _stream()), "wc_conv_split_colsum_f32")
    return hi, lo, scale, part


def _storage_extent(w):
    return 1 + sum((n - 1) * s for n, s in zip(w.shape, w.stride()))


def weight_image(w, geom, k_axis, n_axis):
    """Fragment image of a (dense) 4-d weight for a geometry; k_axis / n_axis = which axis is reduced / produced."""
    lib = _lib.load()
    nbytes = lib.wc_conv_weights_bytes(ctypes.addressof(geom))
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    scale = torch.empty(1 + 512, dtype=torch.float32, device=w.device)
    ext = _storage_extent(w)
    if ext != w.numel():
        raise ValueError("weight must be dense")
    known = getattr(w, '_wc_amax', None)        # floats whose maximum is max|w| (left by the spectral-norm op): no sweep here
    _lib.check(lib.wc_conv_weights_f32(_ptr(w), w.stride(k_axis), w.stride(n_axis), w.stride(2), w.stride(3), ext,
                                       ctypes.addressof(geom), _ptr(img), _ptr(scale), scale.data_ptr() + 4,
                                       _ptr(known), 0 if known is None else known.numel(), _stream()),
               "wc_conv_weights_f32")
    return img, scale


def weight_image_pair(w, fwd, bwd):
    """The forward and the data-gradient image of one weight in one launch; fwd / bwd = (geometry, k_axis, n_axis)."""
    lib = _lib.load()
    (ga, ka, na), (gb, kb, nb) = fwd, bwd
    img_a = torch.empty(lib.wc_conv_weights_bytes(ctypes.addressof(ga)), dtype=torch.uint8, device=w.device)
    img_b = torch.empty(lib.wc_conv_weights_bytes(ctypes.addressof(gb)), dtype=torch.uint8, device=w.device)
    scale = torch.empty(2 + 512, dtype=torch.float32, device=w.device)
    ext = _storage_extent(w)
    if ext != w.numel():
        raise ValueError("weight must be dense")
    known = getattr(w, '_wc_amax', None)
    _lib.check(lib.wc_conv_weights_pair_f32(_ptr(w), w.stride(2), w.stride(3), ext,
                                            w.stride(ka), w.stride(na), ctypes.addressof(ga), _ptr(img_a),
                                            w.stride(kb), w.stride(nb), ctypes.addressof(gb), _ptr(img_b),
                                            _ptr(scale), scale.data_ptr() + 8, _ptr(known), 0 if known is None else known.numel(),
                                            _stream()), "wc_conv_weights_pair_f32")
    return (img_a, scale[0:1]), (img_b, scale[1:2])


def _cached_image(w, key, geom, k_axis, n_axis):
    """The image is rebuilt on every call (two small launches).  It used to be cached on the weight tensor per
    `w._version` -- but fused optimizers (torch's fused Adam) and replayed hipGraphs update weights WITHOUT moving that
    counter, and a stale image is a silently wrong convolution; the reuse it bought (the generator's 7 weights, twice per
    step) was ~0.1 ms."""
    return weight_image(w, geom, k_axis, n_axis)


def run(planes, image, geom, bias=None, relu=False, nbytes=None):
    hi, lo, xs = planes
    img, ws = image
    lib = _lib.load()
    y = torch.empty((geom.N, geom.Hout, geom.Wout, geom.Cout), dtype=torch.float32, device=hi.device)
    if nbytes is None:
        nbytes = lib.wc_conv_workspace_bytes(ctypes.addressof(geom))
    work = torch.empty(nbytes, dtype=torch.uint8, device=hi.device) if nbytes else None
    _lib.check(lib.wc_conv_f16x3(_ptr(hi), _ptr(lo), _ptr(xs), _ptr(img), _ptr(ws), _ptr(bias) if bias is not None else None,
                                 _ptr(_zero_line(hi.device)), ctypes.addressof(geom), 1 if relu else 0, _ptr(y),
                                 _ptr(work), nbytes, _stream()), "wc_conv_f16x3")
    return y


def _geoms(kind, N, H, W, w):
    """(forward geometry, weight axes (k, n)), (data-gradient geometry, axes) for x of shape (N, H, W, Cin)"""
    if kind == 'same':
        co, ci, k = w.shape[0], w.shape[1], w.shape[2]
        fwd = _dense_geom(N, H, W, ci, co, k, 1)
        bwd = _dense_geom(N, H, W, co, ci, k, 1)
        for t in range(k * k):                       # dx[y] = sum_r g[y + pad - r] w[r]
            bwd.dy[0][t], bwd.dx[0][t] = -bwd.dy[0][t], -bwd.dx[0][t]
        return (fwd, 1, 0), (bwd, 0, 1)
    if kind in ('down', 'down3'):                    # 'down3': the 3x3 weight itself, the pooled 4x4 kernel is formed in the image
        co, ci = w.shape[0], w.shape[1]
        src = _pooled3 if kind == 'down3' else _plain
        return (_dense_geom(N, H, W, ci, co, 4, 2, src), 1, 0), (_phase_geom(N, H // 2, W // 2, co, ci, src), 0, 1)
    if kind == 'up3':                                # the 3x3 weight (Cout, Cin, 3, 3) of UpSampling2D -> Conv2D
        co, ci = w.shape[0], w.shape[1]
        return (_phase_geom(N, H, W, ci, co, _upsampled3), 1, 0), (_dense_geom(N, 2 * H, 2 * W, co, ci, 4, 2, _upsampled3), 0, 1)
    ci, co = w.shape[0], w.shape[1]                  # 'up'
    return (_phase_geom(N, H, W, ci, co), 0, 1), (_dense_geom(N, 2 * H, 2 * W, co, ci, 4, 2), 1, 0)


def weight_gradient(x_planes, g_planes, geom, w, k_axis, n_axis, nbytes=None, colsum=None):
    """dW (in w's own layout) from the split planes of the layer input and of the output gradient; `geom` = the forward
    geometry.  colsum: the partial rows split_planes(gy, colsum=True) left -> returns (dW, db)."""
    lib = _lib.load()
    xh, xl, xs = x_planes
    gh, gl, gs = g_planes[:3]
    dw = torch.empty_like(w)
    if dw.stride() != w.stride():
        raise ValueError("weight must be dense")
    if nbytes is None:
        nbytes = lib.wc_conv_wrw_workspace_bytes(ctypes.addressof(geom))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    if colsum is not None:
        db = torch.empty(geom.Cout, dtype=torch.float32, device=w.device)
        _lib.check(lib.wc_conv_wrw_bias_f16x3(_ptr(xh), _ptr(xl), _ptr(xs), _ptr(gh), _ptr(gl), _ptr(gs), _ptr(_zero_line(w.device)),
                                              ctypes.addressof(geom), _ptr(dw), w.stride(k_axis), w.stride(n_axis), w.stride(2),
                                              w.stride(3), _ptr(colsum), _ptr(db), _ptr(ws), nbytes, _stream()),
                   "wc_conv_wrw_bias_f16x3")
        return dw, db
    _lib.check(lib.wc_conv_wrw_f16x3(_ptr(xh), _ptr(xl), _ptr(xs), _ptr(gh), _ptr(gl), _ptr(gs), _ptr(_zero_line(w.device)),
                                     ctypes.addressof(geom), _ptr(dw), w.stride(k_axis), w.stride(n_axis), w.stride(2),
                                     w.stride(3), _ptr(ws), nbytes, _stream()), "wc_conv_wrw_f16x3")
    return dw


def takes_planes(shape, wshape, kind):
    """Would fast_conv_or_none take an input of this NHWC shape as planes handed over by K3 (functional.whiten_color(planes=True))?
    Shapes only -- the caller asks before it runs the site."""
    class _S:
        pass
    xs, ws = _S(), _S()
    xs.shape, ws.shape = tuple(shape), tuple(wshape)
    p = _plan(kind, xs, ws)
    return bool(p) and p.ok


class _FastConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, kind, plan, relu_input=False, handed=None, site=None):
        gf, kf, nf = plan.fwd
        # handed: x is a K3 handle and these are its planes (already ReLU'd and split by K3's epilogue: no pass here)
        # relu_input: the layer is conv(relu(x)); the ReLU happens in the split
        # site: the layer object (generator.Conv2D) -- its splits take their scale from the site's previous call (split_planes)
        planes = handed if handed is not None else split_planes(x, relu=relu_input, site=site, role='x')
        ctx.site = site
        if ctx.needs_input_grad[0]:                 # the data gradient will want its image too: both in one launch
            img, ctx.bwd_image = weight_image_pair(w, plan.fwd, plan.bwd)
        else:
            img, ctx.bwd_image = weight_image(w, gf, kf, nf), None
        y = run(planes, img, gf, bias, nbytes=plan.fwd_ws)
        # the planes stand in for (relu of) x (same bytes) in the weight gradient; x itself only for the ReLU mask
        ctx.save_for_backward(w, *planes, *((x,) if relu_input else ()))
        ctx.kind, ctx.has_bias, ctx.plan, ctx.relu_input = kind, bias is not None, plan, relu_input
        return y

    @staticmethod
    def backward(ctx, gy):
        w, xh, xl, xs = ctx.saved_tensors[:4]
        kind, plan = ctx.kind, ctx.plan
        gy = gy.contiguous()
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        fused_db = want_db and ctx.needs_input_grad[1] and _colsum_ok(gy.shape[-1])    # db rides on the split + the dW reduction
        g_planes = split_planes(gy, colsum=fused_db, site=ctx.site, role='g')
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            gb, kb, nb = plan.bwd
            image = ctx.bwd_image if ctx.bwd_image is not None else weight_image(w, gb, kb, nb)
            dx = run(g_planes[:3], image, gb, nbytes=plan.bwd_ws)
            if ctx.relu_input:
                dx = torch.ops.aten.threshold_backward(dx, ctx.saved_tensors[4], 0)
        if ctx.needs_input_grad[1]:
            gf, kf, nf = plan.fwd
            if fused_db:
                dw, db = weight_gradient((xh, xl, xs), g_planes, gf, w, kf, nf, nbytes=plan.wrw_ws, colsum=g_planes[3])
            else:
                dw = weight_gradient((xh, xl, xs), g_planes, gf, w, kf, nf, nbytes=plan.wrw_ws)
        if want_db and not fused_db:
            db = gy.sum((0, 1, 2))
        return dx, dw, db, None, None, None, None, None


_ones = {}


def _one(device):
    """the device scalar 1.0: the `x scale` of planes that carry their scales per channel (folded into the weight instead)"""
    t = _ones.get(device)
    if t is None:
        t = torch.ones(1, dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _ones[device] = t
    return t


class _SplitConv(torch.autograd.Function):
    """A 1x1 convolution (the block's shortcut, generator.py:142-146) whose input arrives as the pre-split planes of the residual add
    in front (ops.SplitTensor: x[c] = center[c] + (hi + lo)[c] / scale[c]): the kernel runs on the planes themselves with
    wf[o][c] = w[o][c] / scale[c] and bf = bias + <center, w[o]> (wc_fold_channel_scale_f32), so x is neither converted back to fp32
    nor split again (rounds 1-3: one pass for max |x| and one to split, per shortcut and step).  Backward: the data gradient with the
    weight itself; the weight gradient of wf on the same planes, unfolded (wc_unfold_channel_scale_f32)."""

    @staticmethod
    def forward(ctx, x, w, bias, plan, st, site=None):
        from . import ops
        ctx.site = site
        gf, kf, nf = plan.fwd
        wf, bf = ops.fold_channel_scale(w, bias, st.scale, st.center)
        img = weight_image(wf, gf, kf, nf)
        ctx.bwd_image = None
        if ctx.needs_input_grad[0]:
            gb, kb, nb = plan.bwd
            ctx.bwd_image = weight_image(w, gb, kb, nb)
        hi, lo = st.planes[0].view(st.shape), st.planes[1].view(st.shape)
        one = _one(hi.device)
        y = run((hi, lo, one), img, gf, bf, nbytes=plan.fwd_ws)
        ctx.save_for_backward(w, hi, lo, one, st.scale, st.center)
        ctx.plan, ctx.has_bias = plan, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        w, xh, xl, one, scale, center = ctx.saved_tensors
        plan = ctx.plan
        gy = gy.contiguous()
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        fused_db = need_w and _colsum_ok(gy.shape[-1])       # db rides on the split + the dW reduction (the unfolding needs it anyway)
        g_planes = split_planes(gy, colsum=fused_db, site=ctx.site, role='g')
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            gb, kb, nb = plan.bwd
            image = ctx.bwd_image if ctx.bwd_image is not None else weight_image(w, gb, kb, nb)
            dx = run(g_planes[:3], image, gb, nbytes=plan.bwd_ws)
        if need_w:
            gf, kf, nf = plan.fwd
            if fused_db:
                D, db = weight_gradient((xh, xl, one), g_planes, gf, w, kf, nf, nbytes=plan.wrw_ws, colsum=g_planes[3])
            else:
                D = weight_gradient((xh, xl, one), g_planes, gf, w, kf, nf, nbytes=plan.wrw_ws)
                db = gy.sum((0, 1, 2))
            dw = ops.unfold_channel_scale(D, db, scale, center)
        elif need_b:
            db = gy.sum((0, 1, 2))
        return dx, dw, (db if need_b else None), None, None, None


def split_conv(x, st, w, bias=None, site=None):
    """conv1x1(x) for a handle x whose data is the ops.SplitTensor st (see _SplitConv).  Raises when the kernel does not take the
    shape -- the producer asks takes_planes() before it writes planes."""
    p = _plan('same', x, w)
    if not (p and p.ok and tuple(w.shape[2:]) == (1, 1) and w.dtype == torch.float32):
        raise _lib.WcHipError(f"split_conv: a pre-split handle reached a convolution without a planes path {tuple(x.shape)} x {tuple(w.shape)}")
    return _SplitConv.apply(x, w, bias, p, st, site)


def fast_conv_or_none(x, w, bias=None, kind='same', relu_input=False, site=None):
    """fast_conv when the kernel takes the call, else None (the caller's other path).  relu_input: conv(relu(x)).
    site: the layer object that owns this convolution (split_planes keeps the splits' scale history there)."""
    handed = getattr(x, '_wc_planes', None)
    if not supported(x, w, kind):
        if handed is not None:
            raise _lib.WcHipError(f"fast_conv: a K3 handle reached a convolution that cannot take planes {tuple(x.shape)} x {tuple(w.shape)} ({kind})")
        return None
    if handed is not None:
        if relu_input:
            raise ValueError("a K3 handle is already ReLU'd")
        return _FastConv.apply(x, w, bias, kind, _plan(kind, x, w), False, handed, site)
    return _FastConv.apply(x, w, bias, kind, _plan(kind, x, w), relu_input, None, site)


# The critic's first block reads images (Conv2D 3 -> 128 and the 1x1 shortcut 3 -> 128): the forward stays with MIOpen, the weight and bias
# gradients come from ONE pass over gy on the fp32 matrix pipe (wc_conv_wrw_narrow_f32; WC_NARROW_WRW=0: MIOpen's, rounds 1-4)
NARROW_WRW = os.environ.get('WC_NARROW_WRW', '1') != '0'


def narrow_wrw_supported(x, w):
    """x NHWC fp32 on the GPU, w (Cout, Cin, k, k) with k in (1, 3), k*k*Cin < 32, Cout a multiple of 128"""
    if not (NARROW_WRW and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 4 and w.dim() == 4):
        return False
    k = w.shape[2]
    if w.shape[3] != k or k not in (1, 3) or w.shape[1] != x.shape[3]:
        return False
    N, H, W, C = x.shape
    return bool(_lib.load().wc_conv_wrw_narrow_supported(N, H, W, C, w.shape[0], k))


def narrow_forward(x, w, bias=None, relu=False, mirrored=False):
    """wc_conv_fwd_narrow_f32: y = conv(x, w) + bias for NHWC x with a handful of channels, w (Cout, Cin, k, k) in any dense layout.
    mirrored: w is (Cin, Cout, k, k) of the TRANSPOSED map and its taps are read back to front -- y is then the data gradient of the
    convolution w belongs to (x = its output gradient)."""
    lib = _lib.load()
    N, H, W, C = x.shape
    k = w.shape[2]
    if mirrored:
        O = w.shape[1]
        ptr = ctypes.c_void_p(w.data_ptr() + 4 * ((k - 1) * w.stride(2) + (k - 1) * w.stride(3)))
        strides = (w.stride(0), w.stride(1), -w.stride(2), -w.stride(3))
    else:
        O = w.shape[0]
        ptr = _ptr(w)
        strides = (w.stride(1), w.stride(0), w.stride(2), w.stride(3))
    y = torch.empty((N, H, W, O), dtype=torch.float32, device=x.device)
    xc = x if x.is_contiguous() else x.contiguous()
    _lib.check(lib.wc_conv_fwd_narrow_f32(_ptr(xc), ptr, *strides, _ptr(bias), N, H, W, C, O, k, 1 if relu else 0, _ptr(y), _stream()),
               "wc_conv_fwd_narrow_f32")
    return y


class _NarrowInConv(torch.autograd.Function):
    """'same' convolution of an image-like input (k*k*Cin < 32): forward and weight / bias gradient on the fp32 matrix pipe
    (wc_conv_fwd_narrow_f32, wc_conv_wrw_narrow_f32); the data gradient -- wanted in the generator update only -- by MIOpen."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return narrow_forward(x, w, bias)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        N, H, W, C = x.shape
        O, k = w.shape[0], w.shape[2]
        lib = _lib.load()
        dx = dw = db = None
        g = gy if gy.is_contiguous() else gy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, [1, 1], [k // 2, k // 2], [1, 1],
                                                     False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_strided(w.shape, w.stride(), dtype=torch.float32, device=w.device)
            if _storage_extent(dw) != dw.numel():
                raise ValueError("weight must be dense")
            db = torch.empty(O, dtype=torch.float32, device=w.device) if ctx.has_bias else None
            nb = lib.wc_conv_wrw_narrow_workspace_bytes(N, H, W, C, O, k)
            ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
            xc = x if x.is_contiguous() else x.contiguous()
            _lib.check(lib.wc_conv_wrw_narrow_f32(_ptr(xc), _ptr(g), N, H, W, C, O, k, _ptr(dw), dw.stride(1), dw.stride(0), dw.stride(2),
                                                  dw.stride(3), _ptr(db), _ptr(ws), nb, _stream()), "wc_conv_wrw_narrow_f32")
        return dx, dw, db


def narrow_in_conv(x, w, bias=None):
    """y = conv(x, w) + bias for NHWC x with a handful of channels (narrow_wrw_supported)"""
    if _storage_extent(w) != w.numel():
        w = w.contiguous()
    return _NarrowInConv.apply(x, w, bias)


def narrow_out_weight_gradient(x, gy, w):
    """dW of a 3x3 'same' convolution with a handful of OUTPUT channels (the generator's last layer, 256 -> 3) by the same one-pass kernel with
    the operands' roles exchanged: dW[o][c][r][s] = sum_q gy[q - (r-1, s-1)][o] x[q][c] is the narrow-input gradient of a convolution that reads
    gy (3 channels) and whose output gradient is x (256 channels), with the taps mirrored -- written through negative tap strides.  x, gy NHWC."""
    lib = _lib.load()
    N, H, W, C = x.shape
    O, k = w.shape[0], w.shape[2]
    dw = torch.empty_strided(w.shape, w.stride(), dtype=torch.float32, device=w.device)
    if _storage_extent(dw) != dw.numel():
        raise ValueError("weight must be dense")
    nb = lib.wc_conv_wrw_narrow_workspace_bytes(N, H, W, O, C, k)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    last = dw.data_ptr() + 4 * ((k - 1) * dw.stride(2) + (k - 1) * dw.stride(3))        # tap (k-1, k-1): the mirrored (0, 0)
    _lib.check(lib.wc_conv_wrw_narrow_f32(_ptr(gy), _ptr(x), N, H, W, O, C, k, ctypes.c_void_p(last), dw.stride(0), dw.stride(1), -dw.stride(2),
                                          -dw.stride(3), None, _ptr(ws), nb, _stream()), "wc_conv_wrw_narrow_f32")
    return dw


def narrow_out_wrw_supported(x, w):
    if not (NARROW_WRW and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 4 and w.dim() == 4):
        return False
    k = w.shape[2]
    if w.shape[3] != k or k not in (1, 3) or w.shape[1] != x.shape[3]:
        return False
    N, H, W, C = x.shape
    return bool(_lib.load().wc_conv_wrw_narrow_supported(N, H, W, w.shape[0], C, k))


def fast_conv(x, w, bias=None, kind='same'):
    """NHWC convolution (see the module docstring) -- raises if the shape is not one the kernel takes."""
    if not supported(x, w, kind):
        raise _lib.WcHipError(f"fast_conv: unsupported call {tuple(x.shape)} x {tuple(w.shape)} ({kind})")
    return _FastConv.apply(x, w, bias, kind, _plan(kind, x, w))
