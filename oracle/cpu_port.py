"""TEST INFRASTRUCTURE -- ctypes loader of oracle/libwc_cpu.so (oracle/wc_cpu.cpp): the CPU restatement of the WC path behind
the `_cpu`-suffixed C ABI of include/wc_hip.h, on numpy arrays.  Only tests/, __graft_entry__ and bench.py's cpu legs use it;
nothing under wc_gan_amd/ may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_P, _I, _L, _D, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
_SIGS = {
    "wc_cpu_threads": (_I, []),
    "wc_stats_f32_cpu": (_I, [_P, _L, _I, _I, _P, _P, _P, _Z, _P]),
    "wc_factor_f64_cpu": (_I, [_P, _P, _L, _I, _I, _D, _D, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "wc_color_f32_cpu": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "wc_group_bias_f32_cpu": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "wc_apply_f32_cpu": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P, _Z, _P]),
    "wc_apply_act_f32_cpu": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "wc_bwd_reduce_f32_cpu": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P, _Z, _P]),
    "wc_bwd_factor_f64_cpu": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _D, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "wc_bwd_apply_f32_cpu": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _Z, _P]),
}


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return os.path.join(_HERE, "libwc_cpu.so")


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libwc_cpu.so")
        src = os.path.join(_HERE, "wc_cpu.cpp")
        alt = os.environ.get("WC_CPU_PORT_LIB")          # another build of the same source (`make -C oracle asan`)
        if alt:
            path = alt
        elif not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
            build()
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> {rc}")


def stats(x2d, groups=1):
    x2d = _f32(x2d); M, Cc = x2d.shape
    s = np.empty((groups, Cc)); xtx = np.empty((groups, Cc, Cc))
    _chk(load().wc_stats_f32_cpu(_p(x2d), M, Cc, groups, _p(s), _p(xtx), None, 0, None), "wc_stats_f32_cpu")
    return (s[0], xtx[0]) if groups == 1 else (s, xtx)


def factor(s, xtx, M, Cc, eps=1e-3, momentum=0.99, ddof=1, training=True, moving_mean=None, moving_cov=None, groups=1):
    """-> mu f32, L f64, W f64, chan_scale f32; moving_mean / moving_cov (float32 arrays) are updated in place."""
    mu = np.empty((groups, Cc), np.float32); cs = np.empty(Cc, np.float32)
    L = np.empty((groups, Cc, Cc)); W = np.empty((groups, Cc, Cc))
    s = None if s is None else np.ascontiguousarray(s, np.float64); xtx = None if xtx is None else np.ascontiguousarray(xtx, np.float64)
    _chk(load().wc_factor_f64_cpu(_p(s), _p(xtx), int(M), Cc, groups, eps, momentum, ddof, int(training), _p(moving_mean),
                                  _p(moving_cov), _p(mu), _p(cs), _p(L), _p(W), None, 0, None), "wc_factor_f64_cpu")
    return (mu[0], L[0], W[0], cs) if groups == 1 else (mu, L, W, cs)


def color(W, gamma, groups=1, per_group=False):
    W = np.ascontiguousarray(W, np.float64); Cc = W.shape[-1]
    gamma = _f32(gamma)
    Kc = 1 if gamma is None else (gamma.shape[0] // groups if per_group else gamma.shape[0])
    A = np.empty((groups * Kc, Cc, Cc), np.float32); At = np.empty_like(A)
    _chk(load().wc_color_f32_cpu(_p(W), _p(gamma), Kc, Cc, groups, int(per_group), _p(A), _p(At), None, None, None, 0, None),
         "wc_color_f32_cpu")
    return A, At


def apply(x, mu, A, bias, slot, relu=False):
    x = _f32(x); N, Cc = x.shape[0], x.shape[-1]; HW = x.size // (N * Cc)
    y = np.empty_like(x)
    slot = None if slot is None else np.ascontiguousarray(slot, np.int32)
    _chk(load().wc_apply_act_f32_cpu(_p(x), _p(_f32(mu)), _p(_f32(A)), _p(_f32(bias)), _p(slot), N, HW, Cc, A.shape[0], int(relu),
                                     _p(y), None, None, 0, None), "wc_apply_act_f32_cpu")
    return y


def bwd_reduce(x, mu, gy, slot, Kc):
    x = _f32(x); gy = _f32(gy); N, Cc = x.shape[0], x.shape[-1]; HW = x.size // (N * Cc)
    R = np.empty((Kc, Cc, Cc)); gs = np.empty((Kc, Cc))
    slot = None if slot is None else np.ascontiguousarray(slot, np.int32)
    _chk(load().wc_bwd_reduce_f32_cpu(_p(x), _p(_f32(mu)), _p(gy), _p(slot), N, HW, Cc, Kc, _p(R), _p(gs), None, 0, None),
         "wc_bwd_reduce_f32_cpu")
    return R, gs


def bwd_factor(R, gsum, W, L, gamma, A, M, eps=1e-3, ddof=1, training=True):
    Kc, Cc = R.shape[0], R.shape[1]
    gamma = _f32(gamma)
    dg = np.empty((Kc, Cc, Cc), np.float32) if gamma is not None else None
    db = np.empty((Kc, Cc), np.float32)
    S = np.empty((Cc, Cc), np.float32) if training else None
    gm = np.empty(Cc, np.float32) if training else None
    _chk(load().wc_bwd_factor_f64_cpu(_p(np.ascontiguousarray(R)), _p(np.ascontiguousarray(gsum)), _p(np.ascontiguousarray(W)),
                                      _p(np.ascontiguousarray(L)), _p(gamma), _p(_f32(A)), Kc, Cc, int(M), eps, ddof, int(training),
                                      _p(dg), _p(db), _p(S), _p(gm), None, 0, None), "wc_bwd_factor_f64_cpu")
    return dg, db, S, gm


def bwd_apply(gy, x, mu, At, S, gmean, slot):
    gy = _f32(gy); N, Cc = gy.shape[0], gy.shape[-1]; HW = gy.size // (N * Cc)
    dx = np.empty_like(gy)
    slot = None if slot is None else np.ascontiguousarray(slot, np.int32)
    _chk(load().wc_bwd_apply_f32_cpu(_p(gy), _p(_f32(x)), _p(_f32(mu)), _p(_f32(At)), _p(_f32(S)), _p(_f32(gmean)), _p(slot),
                                     N, HW, Cc, At.shape[0], _p(dx), None, 0, None), "wc_bwd_apply_f32_cpu")
    return dx


def forward_backward(x, gamma, beta, slot, gy, eps=1e-3, momentum=0.99, moving_mean=None, moving_cov=None, relu=False):
    """The whole site through the `_cpu` stages (training mode): y, dx, dgamma, dbeta (+ updated moving statistics in place)."""
    N, Cc = x.shape[0], x.shape[-1]
    M = x.size // Cc
    s, xtx = stats(x.reshape(M, Cc))
    mu, L, W, _ = factor(s, xtx, M, Cc, eps, momentum, 1, True, moving_mean, moving_cov)
    A, At = color(W, gamma)
    y = apply(x, mu, A, beta, slot, relu)
    g = np.where(y > 0, gy, 0).astype(np.float32) if relu else gy
    R, gs = bwd_reduce(x, mu, g, slot, A.shape[0])
    dg, db, S, gm = bwd_factor(R, gs, W, L, gamma, A, M, eps, 1, True)
    dx = bwd_apply(g, x, mu, At, S, gm, slot)
    return y, dx, dg, db
