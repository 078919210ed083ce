"""The identity behind wc_bwd_factor_f64's one-product Cholesky step (csrc/wc_abi.hip), in float64 numpy on random
symmetric positive definite matrices: with T = L L^T, W = L^-1 and any Wbar,

    Phi(L^T Lbar) with Lbar = -tril(W^T Wbar W^T)   ==   -Phi(Wbar W^T),

Phi = lower triangle with the diagonal halved (the reference's tf.cholesky gradient: /root/reference's whitening goes through
tf.cholesky + tf.matrix_triangular_solve, whose registered gradients are the textbook chain on the left).  No GPU."""
import numpy as np
import pytest


def _phi(X):
    P = np.tril(X)
    P[np.diag_indices_from(P)] *= 0.5
    return P


@pytest.mark.parametrize("C,cond", [(16, 1e2), (64, 1e4), (256, 1e6)])
def test_collapsed_cholesky_step_equals_the_textbook_chain(C, cond):
    rng = np.random.default_rng(C)
    Q, _ = np.linalg.qr(rng.standard_normal((C, C)))
    T = (Q * np.geomspace(1.0, 1.0 / cond, C)) @ Q.T
    T = 0.5 * (T + T.T)
    L = np.linalg.cholesky(T)
    W = np.linalg.inv(L)
    Wbar = rng.standard_normal((C, C))
    Lbar = -np.tril(W.T @ Wbar @ W.T)
    P_textbook = _phi(L.T @ Lbar)
    P_one = -_phi(Wbar @ W.T)
    scale = np.abs(P_textbook).max()
    assert np.abs(P_one - P_textbook).max() <= 1e-9 * scale          # (measured: 1e-13 .. 4e-11; what is left is W L = I to rounding times cond)
    # ... and the statistic's gradient built from it is the same symmetric matrix
    S1 = W.T @ P_textbook @ W
    S2 = W.T @ P_one @ W
    assert np.abs((S1 + S1.T) - (S2 + S2.T)).max() <= 1e-9 * np.abs(S1 + S1.T).max()
