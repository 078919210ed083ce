"""L and W of wc_factor_f64 against scipy float64 at C = 256 on an ill-conditioned covariance (development: Newton steps)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
from wc_gan_amd import ops
for C in (256, 128):
    rng = np.random.default_rng(2)
    M = 8 * C + 3
    X = o.synth_activation(rng, (M, C), "ill")
    s, xtx, _ = o.batch_moments(X)
    mu_ref, sigma = o.moments_to_stats(s, xtx, M)
    L_ref, W_ref = o.whitening_matrix(sigma, 1e-3)
    d = lambda a: torch.tensor(a, dtype=torch.float64, device='cuda')
    mu, L, W = ops.factor(d(s), d(xtx), M, C, 1e-3, 0.99, 1, True, None, None, 'cuda')
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(C, "L rel %.2e  W rel %.2e  cond %.1e" % (rel(L.cpu().numpy(), L_ref), rel(W.cpu().numpy(), W_ref), np.linalg.cond((1-1e-3)*sigma + 1e-3*np.eye(C))))
