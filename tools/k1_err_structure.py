"""CPU analysis of the dx error budget at 128x32x32x256 for the seed-sweep inputs: how much of it is K1's covariance, and what
structure the covariance error has.  Needs gpurun_out/k1_moments_seed<seed>.npz (tools/k1_moments_dump.py on the GPU)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
shape = (128, 32, 32, 256); C = 256
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for seed in (int(a) for a in sys.argv[1:]):
    rng = np.random.default_rng(seed)
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1); G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, 1, shape[0]).astype(np.int32)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B, slot)
    dx_ref, _, _ = o.wc_backward(gy, cache)
    M = cache['M']
    X = x.reshape(-1, C).astype(np.float64)
    mu_ref, sig_ref = o.moments_to_stats(X.sum(0), X.T @ X, M)
    T = (1 - 1e-3) * sig_ref + 1e-3 * np.eye(C)
    ev = np.linalg.eigvalsh(T)
    d = np.load("gpurun_out/k1_moments_seed%d.npz" % seed)
    mu_g, sig_g = o.moments_to_stats(d['s'], d['xtx'], M)
    sd = np.sqrt(np.diag(sig_ref))
    E = (sig_g - sig_ref) / np.outer(sd, sd)
    rho = sig_ref / np.outer(sd, sd)
    iu = np.triu_indices(C, 1)
    a, b = np.polyfit(rho[iu], E[iu], 1)[::-1]
    print("seed %d: cond(T) %.2e  lambda_min %.3e | cov err/sqrt(sii sjj): diag mean %.2e std %.2e | offdiag mean %.2e std %.2e | fit err = %.2e + %.2e rho, residual std %.2e"
          % (seed, ev[-1] / ev[0], ev[0], np.diag(E).mean(), np.diag(E).std(), E[iu].mean(), E[iu].std(), a, b, (E[iu] - a - b * rho[iu]).std()))
    def dx_with(sig, mu=mu_ref):
        L, W = o.whitening_matrix(sig, 1e-3)
        c = dict(cache); c['W'] = W; c['L'] = L; c['A'] = np.einsum('ji,kjo->kio', W, cache['G']) if cache['G'].ndim == 3 else W.T @ cache['G']
        c['f'] = X - mu
        return o.wc_backward(gy, c)[0]
    print("   dx error with the GPU covariance, everything else float64: %.2e" % rel(dx_with(sig_g), dx_ref))
    off = ~np.eye(C, dtype=bool)
    s1 = sig_g.copy(); s1[off] -= (a * np.outer(sd, sd))[off]
    print("   ... offdiag debiased by its mean (a sqrt(sii sjj)): %.2e" % rel(dx_with(s1), dx_ref))
    s2 = sig_g.copy(); s2[off] -= ((a + b * rho) * np.outer(sd, sd))[off]
    print("   ... offdiag debiased by a + b rho: %.2e" % rel(dx_with(s2), dx_ref))
    s3 = sig_g / (1 + b)          # everything scaled
    print("   ... whole matrix scaled by 1/(1+b): %.2e" % rel(dx_with(s3), dx_ref))
    s4 = sig_ref.copy(); s4[np.diag_indices(C)] = np.diag(sig_g)
    print("   ... exact off-diagonal, GPU diagonal: %.2e" % rel(dx_with(s4), dx_ref))
    s5 = sig_g.copy(); s5[np.diag_indices(C)] = np.diag(sig_ref)
    print("   ... GPU off-diagonal, exact diagonal: %.2e" % rel(dx_with(s5), dx_ref))
    rs = np.random.default_rng(1); N_ = rs.standard_normal((C, C)); N_ = (N_ + N_.T) / 2
    s6 = sig_ref + N_ * E[iu].std() * np.outer(sd, sd)
    print("   ... exact + symmetric gaussian noise of the same std: %.2e" % rel(dx_with(s6), dx_ref))
