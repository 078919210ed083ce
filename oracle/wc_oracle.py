"""Float64 CPU oracle for the whitening-and-coloring (WC) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.

PARITY UNPINNED.  The arithmetic of the reference lives in an un-vendored,
empty git submodule (``/root/reference/.gitmodules:1-3`` ->
``AliaksandrSiarohin/gan``, ``gan/conditional_layers.py``; pinned SHA unknown)
on top of ``tensorflow==1.5.0`` / ``keras==2.0.8`` (``README.md:18-26``); neither
can be imported here and the reference holds no tests or golden vectors
(SURVEY.md section 8c).  This file therefore restates the *published* algorithm
(ICLR'19 "Whitening and Coloring batch transform for GANs", arXiv 1806.00420,
eqs. for Cholesky whitening and (conditional) coloring) anchored on the
reference's own call sites:

* ``generator.py:24,26``   ``DecorelationNormalization(name=, renorm=)``
* ``generator.py:49-51``   ``uconv``   : 1x1 ``Conv2D(filters=C)`` coloring
* ``generator.py:52-60``   ``ucconv``  : ``ConditionalConv11`` + 1x1 conv, added
* ``generator.py:69-78``   ``ufconv``  : ``FactorizedConv11(filters_emb=E)`` + 1x1 conv
* ``generator.py:28-40``   ``ucs/ccs/uccs`` diagonal colorings
* ``generator.py:83-87``   ``stack``: whitening (``_npart``) then coloring (``_repart``)
* ``scorer.py:60,72``      eval mode -> moving statistics

Every constant that comes from recollection of the upstream submodule rather
than from ``/root/reference`` (epsilon, momentum, the M-1 divisor, the
``(1-eps)*Sigma + eps*I`` shrinkage) is a parameter here, never a literal.

Conventions (row-vector form):
    X      (M, C)   the NHWC activation viewed row-major, M = N*H*W
    mu     (C,)     column mean;           f = X - mu
    Sigma  (C, C)   f^T f / (M - ddof)
    T      (C, C)   (1-eps) Sigma + eps I
    L              lower Cholesky factor of T;  W = L^{-1}
    xhat = f W^T                                  (whitening)
    y_n  = xhat_n Gamma_{k(n)} + beta_{k(n)}      (coloring, k(n) = class slot of sample n)
         = f_n A_{k(n)} + beta_{k(n)},  A_k = W^T Gamma_k   (the fused affine)
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

DEFAULT_EPS = 1e-3        # [UPSTREAM-RECALL]
DEFAULT_MOMENTUM = 0.99   # [UPSTREAM-RECALL]


# --------------------------------------------------------------------------
# statistics + small-matrix stage  (reference row a2: transpose -> GEMM ->
# Cholesky -> triangular solve)
# --------------------------------------------------------------------------
def batch_moments(X):
    """Raw additive moments of the rows of X: (sum (C,), X^T X (C,C), M)."""
    X = np.asarray(X, dtype=np.float64)
    return X.sum(axis=0), X.T @ X, X.shape[0]


def moments_to_stats(s, xtx, M, ddof=1):
    """mean and covariance from raw moments (the form a sync-WC all-reduce adds up)."""
    mu = s / M
    sigma = (xtx - np.outer(s, s) / M) / (M - ddof)
    return mu, 0.5 * (sigma + sigma.T)


def whitening_matrix(sigma, eps=DEFAULT_EPS):
    """L = chol((1-eps) Sigma + eps I) (lower), W = L^{-1} via a triangular solve against I."""
    C = sigma.shape[0]
    T = (1.0 - eps) * sigma + eps * np.eye(C)
    L = sla.cholesky(T, lower=True)
    W = sla.solve_triangular(L, np.eye(C), lower=True)
    return L, W


def zca_matrix(sigma, eps=DEFAULT_EPS):
    """decomposition='zca' (generator.py:24 comment): W = U diag(S^-1/2) U^T of Sigma + eps I."""
    C = sigma.shape[0]
    S, U = np.linalg.eigh(sigma + eps * np.eye(C))
    return (U * (1.0 / np.sqrt(S))) @ U.T


def update_moving(moving_mean, moving_cov, mu, sigma, momentum=DEFAULT_MOMENTUM):
    """moving <- m*moving + (1-m)*batch, with the un-shrunk Sigma (row a2)."""
    mm = momentum * np.asarray(moving_mean, np.float64).reshape(-1) + (1 - momentum) * mu
    mc = momentum * np.asarray(moving_cov, np.float64) + (1 - momentum) * sigma
    return mm, mc


# --------------------------------------------------------------------------
# coloring parameter assembly (rows a6-a9): every after_norm variant reduces to
# a table Gamma_eff (Kc, C, C), beta_eff (Kc, C) indexed per sample.
# --------------------------------------------------------------------------
def coloring_table(after_norm, C, params, number_of_classes=None):
    """Effective per-slot coloring (Gamma_eff[Kc,C,C], beta_eff[Kc,C]) for a create_norm alphabet value.

    ``params`` keys (all float64 arrays), following the Keras weight shapes of the
    call sites in generator.py:28-80:
      u_kernel (C,C), u_bias (C,)                    -- 1x1 Conv2D  (kernel[0,0] is (C_in, C_out))
      c_kernel (K,C,C), c_bias (K,C)                 -- ConditionalConv11
      f_kernel (E,C,C), f_alpha (K,E)                -- FactorizedConv11 (use_bias=False)
      u_gamma (C,), u_beta (C,)                      -- CenterScale
      c_gamma (K,C), c_beta (K,C)                    -- ConditionalCenterScale
    """
    K = number_of_classes
    eye = np.eye(C)
    z = np.zeros
    if after_norm == 'n':
        return eye[None], z((1, C))
    if after_norm == 'uconv':
        return params['u_kernel'][None], params['u_bias'][None]
    if after_norm == 'ucs':
        return np.diag(params['u_gamma'])[None], params['u_beta'][None]
    if after_norm == 'ccs':
        return np.stack([np.diag(g) for g in params['c_gamma']]), params['c_beta'].copy()
    if after_norm == 'uccs':
        G = np.stack([np.diag(g) for g in params['c_gamma']]) + np.diag(params['u_gamma'])[None]
        return G, params['c_beta'] + params['u_beta'][None]
    if after_norm == 'cconv':
        return params['c_kernel'].copy(), params['c_bias'].copy()
    if after_norm == 'ucconv':
        return params['c_kernel'] + params['u_kernel'][None], params['c_bias'] + params['u_bias'][None]
    if after_norm == 'fconv':
        G = np.einsum('ke,eio->kio', params['f_alpha'], params['f_kernel'])
        return G, z((K, C))
    if after_norm == 'ufconv':
        G = np.einsum('ke,eio->kio', params['f_alpha'], params['f_kernel']) + params['u_kernel'][None]
        return G, np.broadcast_to(params['u_bias'][None], (K, C)).copy()
    if after_norm == 'ccsuconv':
        G = np.stack([np.diag(g) for g in params['c_gamma']]) + params['u_kernel'][None]
        return G, params['c_beta'] + params['u_bias'][None]
    raise ValueError(after_norm)


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def wc_forward(x, gamma=None, beta=None, idx=None, *, training=True,
               moving_mean=None, moving_cov=None, eps=DEFAULT_EPS,
               momentum=DEFAULT_MOMENTUM, ddof=1, decomposition='cholesky'):
    """Whitening (+ coloring) of an NHWC tensor.  Returns (y, cache).

    gamma : None (whitening only) | (C,C) | (Kc,C,C);   beta : None | (C,) | (Kc,C)
    idx   : None | int (N,) slot of each sample into gamma/beta's leading axis.
    """
    x = np.asarray(x, dtype=np.float64)
    N, C = x.shape[0], x.shape[-1]
    M = x.size // C
    rows_per_sample = M // N
    X = x.reshape(M, C)
    if training:
        s, xtx, _ = batch_moments(X)
        mu, sigma = moments_to_stats(s, xtx, M, ddof)
    else:
        mu = np.asarray(moving_mean, np.float64).reshape(-1)
        sigma = np.asarray(moving_cov, np.float64)
    if decomposition == 'cholesky':
        L, W = whitening_matrix(sigma, eps)
    elif decomposition == 'zca':
        L, W = None, zca_matrix(sigma, eps)
    else:
        raise ValueError(decomposition)
    f = X - mu
    if gamma is None:
        G = np.eye(C)[None]
    else:
        G = np.asarray(gamma, np.float64).reshape(-1, C, C)
    Kc = G.shape[0]
    B = np.zeros((Kc, C)) if beta is None else np.asarray(beta, np.float64).reshape(-1, C)
    if B.shape[0] == 1 and Kc > 1:
        B = np.broadcast_to(B, (Kc, C))
    A = np.einsum('ji,kjo->kio', W, G)          # A_k = W^T Gamma_k
    if idx is None:
        slot = np.zeros(N, dtype=np.int64)
    else:
        slot = np.asarray(idx).reshape(-1).astype(np.int64)
    row_slot = np.repeat(slot, rows_per_sample)
    y = np.empty_like(X)
    for k in np.unique(row_slot):
        sel = row_slot == k
        y[sel] = f[sel] @ A[k] + B[k]
    cache = dict(mu=mu, sigma=sigma, L=L, W=W, A=A, f=f, G=G, B=B, row_slot=row_slot,
                 M=M, eps=eps, ddof=ddof, training=training, xhat=f @ W.T)
    if training and moving_mean is not None:
        cache['moving_mean'], cache['moving_cov'] = update_moving(moving_mean, moving_cov, mu, sigma, momentum)
    return y.reshape(x.shape), cache


# --------------------------------------------------------------------------
# backward (row a10, extended to Kc coloring slots)
# --------------------------------------------------------------------------
def wc_backward(gy, cache):
    """Closed-form gradients of wc_forward: returns (dx, dgamma (Kc,C,C), dbeta (Kc,C)).

    One big reduction  R_k = sum_{n in k} f_n^T g_n,  bbar_k = sum_{n in k} g_n;
    small stage        Gbar_k = W R_k;  Wbar = sum_k Gamma_k R_k^T;
                       Lbar = -tril(W^T Wbar W^T);  P = Phi(L^T Lbar)  (tril, diagonal halved);
                       Sbar = sym(W^T P W);  S = 2 (1-eps)/(M-ddof) Sbar;
    one big apply      fbar = g A_k^T + f S;   dx = fbar - mean_rows(fbar).
    In eval mode the statistics are constants: dx = g A_k^T.
    """
    W, L, A, G, f = cache['W'], cache['L'], cache['A'], cache['G'], cache['f']
    M, eps, ddof = cache['M'], cache['eps'], cache['ddof']
    row_slot = cache['row_slot']
    C = W.shape[0]
    Kc = G.shape[0]
    g = np.asarray(gy, np.float64).reshape(M, C)
    R = np.zeros((Kc, C, C))
    bbar = np.zeros((Kc, C))
    fbar = np.empty_like(g)
    for k in np.unique(row_slot):
        sel = row_slot == k
        R[k] = f[sel].T @ g[sel]
        bbar[k] = g[sel].sum(axis=0)
        fbar[sel] = g[sel] @ A[k].T
    dgamma = np.einsum('ij,kjo->kio', W, R)     # W R_k
    if not cache['training']:
        return fbar.reshape(np.shape(gy)), dgamma, bbar
    if L is None:
        raise NotImplementedError("closed-form backward is for decomposition='cholesky'")
    Wbar = np.einsum('kij,klj->il', G, R)       # sum_k Gamma_k R_k^T
    Lbar = -np.tril(W.T @ Wbar @ W.T)
    P = np.tril(L.T @ Lbar)
    P[np.diag_indices(C)] *= 0.5
    Sbar = W.T @ P @ W
    Sbar = 0.5 * (Sbar + Sbar.T)
    S = (2.0 * (1.0 - eps) / (M - ddof)) * Sbar
    fbar = fbar + f @ S
    dx = fbar - fbar.mean(axis=0, keepdims=True)
    return dx.reshape(np.shape(gy)), dgamma, bbar


# --------------------------------------------------------------------------
# renorm=True ('dr', generator.py:26) [UPSTREAM-RECALL]: the batch-renormalisation analogue
#   W_eff = L_mov^-1 . stop_gradient(L_batch) . L_batch^-1
# Its VALUE is the moving-statistics whitening (L_b L_b^-1 = I), its GRADIENT flows through the batch factor only.
# With C0 = L_mov^-1 L_batch held constant:  xhat = f W_b^T C0^T,  y = xhat Gamma + beta = f W_b^T (C0^T Gamma) + beta,
# i.e. the plain transform with the coloring Gamma' = C0^T Gamma -- which is how wc_gan_amd.layers folds it.
# The moving statistics used for L_mov are the ones BEFORE this batch's update.
# --------------------------------------------------------------------------
def wc_forward_renorm(x, gamma=None, beta=None, idx=None, *, moving_mean, moving_cov, eps=DEFAULT_EPS,
                      momentum=DEFAULT_MOMENTUM, ddof=1):
    x = np.asarray(x, dtype=np.float64)
    C = x.shape[-1]
    _, c0 = wc_forward(x, None, None, None, training=True, eps=eps, ddof=ddof)
    _, Wm = whitening_matrix(np.asarray(moving_cov, np.float64), eps)
    C0 = Wm @ c0['L']
    G = np.eye(C)[None] if gamma is None else np.asarray(gamma, np.float64).reshape(-1, C, C)
    Geff = np.einsum('ji,kjo->kio', C0, G)                      # C0^T Gamma_k
    y, cache = wc_forward(x, Geff, beta, idx, training=True, moving_mean=moving_mean, moving_cov=moving_cov,
                          eps=eps, momentum=momentum, ddof=ddof)
    cache['C0'] = C0
    return y, cache


def wc_backward_renorm(gy, cache):
    """(dx, dgamma, dbeta) of wc_forward_renorm: C0 is a constant of the step."""
    dx, dGeff, dB = wc_backward(gy, cache)
    return dx, np.einsum('ij,kjo->kio', cache['C0'], dGeff), dB


# --------------------------------------------------------------------------
# the reference's UNFUSED op order (row a2 + a6), used to show fused == unfused
# --------------------------------------------------------------------------
def wc_forward_unfused(x, kernel, bias, eps=DEFAULT_EPS, ddof=1):
    """transpose -> mean -> centre -> f f^T/(M-1) -> shrink -> cholesky -> solve vs I -> W f ->
    transpose back -> 1x1 conv + bias; exactly the op sequence of SURVEY.md row a2/a6."""
    x = np.asarray(x, np.float64)
    C = x.shape[-1]
    Xt = np.transpose(x, (3, 0, 1, 2)).reshape(C, -1)          # (C, M)
    M = Xt.shape[1]
    mu = Xt.mean(axis=1, keepdims=True)
    f = Xt - mu
    sigma = f @ f.T / (M - ddof)
    T = (1 - eps) * sigma + eps * np.eye(C)
    L = sla.cholesky(T, lower=True)
    W = sla.solve_triangular(L, np.eye(C), lower=True)
    xh = (W @ f).reshape((C,) + x.shape[:3])
    xh = np.transpose(xh, (1, 2, 3, 0))                        # NHWC
    return xh @ np.asarray(kernel, np.float64) + np.asarray(bias, np.float64)


# --------------------------------------------------------------------------
# deterministic synthetic inputs
# --------------------------------------------------------------------------
def synth_activation(rng, shape, conditioning='ill'):
    """x = z Mix + 0.2 with a rank-8 spike (cond ~1e6 at C=256) or plain z ('well')."""
    C = shape[-1]
    M = int(np.prod(shape[:-1]))
    z = rng.standard_normal((M, C))
    if conditioning == 'well':
        return z.reshape(shape)
    r = min(8, C)
    mix = rng.standard_normal((C, C)) / np.sqrt(C) + \
        0.3 * (rng.standard_normal((C, r)) @ rng.standard_normal((r, C))) / np.sqrt(r)
    return (z @ mix + 0.2).reshape(shape)


def synth_activation_family(rng, shape, family):
    """Ill-conditioned inputs whose ELEMENTS are not gaussian (round 4, VERDICT r3 item 6: the families on which a constant fitted on
    gaussian mixes could over- or under-correct):  x = z * s + 2 (F V^T) + 0.2  with z (M, C) and F (M, 8) iid unit-variance draws of
    `family` ('uniform', 'relu' = post-ReLU half-sparse, 'heavy' = Laplace, anything else: gaussian), s per-channel scales over two
    decades, V (C, 8) gaussian: eight eigenvalues ~ 4 C on top of idiosyncratic variances down to 1e-4 -- cond((1 - eps) Sigma + eps I)
    ~ 1.2e6 at C = 256 without a dense mix (which would make every element gaussian again).  The whitening's cancellation ratio is
    ~500 here (~30 for synth_activation's 'ill'): the reference's own op order in fp32 is 5e-3 .. 1e-2 from float64 on these."""
    C = shape[-1]
    M = int(np.prod(shape[:-1]))

    def draw(n):
        if family == "uniform":
            return rng.uniform(-np.sqrt(3.0), np.sqrt(3.0), (M, n))
        if family == "relu":                    # post-ReLU: half of the elements exactly zero; unit second moment about the mean
            return np.maximum(rng.standard_normal((M, n)), 0.0) / np.sqrt(0.5 - 1.0 / (2 * np.pi))
        if family == "heavy":                   # Laplace (kurtosis 6; the largest of 8e9 draws ~ 16 sigma)
            return rng.laplace(0.0, 1.0 / np.sqrt(2.0), (M, n))
        return rng.standard_normal((M, n))
    s = 10.0 ** rng.uniform(-2.0, 0.0, C)
    V = rng.standard_normal((C, 8))
    return (draw(C) * s + 2.0 * (draw(8) @ V.T) + 0.2).reshape(shape)


def synth_coloring(rng, C, Kc=1):
    gamma = rng.standard_normal((Kc, C, C)) / np.sqrt(C)
    beta = 0.1 * rng.standard_normal((Kc, C))
    return gamma, beta


# ---------------------------------------------------------------------------------------------------------------------
# N3: spectral normalisation of a weight matrix.
# Call sites: discriminator.py:26-33, generator.py:104-113 (SNConv2D / SNDense / SNEmbeding of the un-vendored
# gan.spectral_normalized_layers; knobs spectral_iterations / fully_diff_spectral, run.py:268-269).
# [UPSTREAM-RECALL] arithmetic: the power-iteration estimate of Miyato et al. (Algorithm 1), as restated here.
# ---------------------------------------------------------------------------------------------------------------------
def spectral_normalize(W, u, v, iterations=1, eps=1e-12):
    """W (R, K), u (R,), v (K,) float64.  `iterations` steps of v <- W^T u / max(|.|, eps), u <- W v / max(|.|, eps);
    sigma = u^T W v; returns (W / sigma, sigma, u, v)."""
    W = np.asarray(W, np.float64); u = np.asarray(u, np.float64).copy(); v = np.asarray(v, np.float64).copy()
    for _ in range(int(iterations)):
        t = W.T @ u
        v = t / max(np.linalg.norm(t), eps)
        s = W @ v
        u = s / max(np.linalg.norm(s), eps)
    sigma = float(u @ (W @ v))
    return W / sigma, sigma, u, v


def spectral_normalize_backward(g, w_sn, u, v, sigma, fully_diff):
    """Gradient w.r.t. W of w_sn = W / sigma, sigma = u^T W v with u, v held constant:
    dW = (g - fully_diff * <g, w_sn> u v^T) / sigma  (fully_diff False: sigma is a constant of the step)."""
    g = np.asarray(g, np.float64)
    c = float((g * w_sn).sum()) if fully_diff else 0.0
    return (g - c * np.outer(u, v)) / sigma
