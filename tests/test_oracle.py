"""CPU tests of the float64 oracle itself: it is the checker for everything else, so pin it.

The reference has no tests or golden vectors for this path: the oracle is checked
against torch's float64 autograd through linalg.cholesky / solve_triangular (an independent
implementation of the same published math), against its own unfused restatement of the reference's
op order, and against the committed golden fixtures.
"""
import os

import numpy as np
import pytest
import torch

from oracle import wc_oracle as o

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "wc_golden.npz")


def _torch_reference(x, G, B, slot, gy, eps=1e-3):
    N, C = x.shape[0], x.shape[-1]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Gt = torch.tensor(G, dtype=torch.float64, requires_grad=True)
    Bt = torch.tensor(B, dtype=torch.float64, requires_grad=True)
    X = xt.reshape(-1, C); M = X.shape[0]
    mu = X.mean(0); f = X - mu
    S = f.T @ f / (M - 1)
    T = (1 - eps) * S + eps * torch.eye(C, dtype=torch.float64)
    L = torch.linalg.cholesky(T)
    W = torch.linalg.solve_triangular(L, torch.eye(C, dtype=torch.float64), upper=False)
    xh = (f @ W.T).reshape(N, -1, C)
    it = torch.tensor(slot).long()
    y = torch.einsum('npc,nco->npo', xh, Gt[it]) + Bt[it][:, None, :]
    y.backward(torch.tensor(gy, dtype=torch.float64).reshape(N, -1, C))
    return y.detach().numpy().reshape(x.shape), xt.grad.numpy(), Gt.grad.numpy(), Bt.grad.numpy()


@pytest.mark.parametrize("shape,Kc", [((6, 3, 5, 16), 4), ((4, 4, 4, 32), 1), ((5, 2, 2, 24), 3)])
def test_oracle_matches_torch_autograd(shape, Kc):
    rng = np.random.default_rng(0)
    x = o.synth_activation(rng, shape)
    G, B = o.synth_coloring(rng, shape[-1], Kc)
    slot = rng.integers(0, Kc, shape[0])
    gy = rng.standard_normal(shape)
    y, cache = o.wc_forward(x, G, B, slot)
    dx, dG, dB = o.wc_backward(gy, cache)
    y_t, dx_t, dG_t, dB_t = _torch_reference(x, G, B, slot, gy)
    assert np.abs(y - y_t).max() < 1e-10
    assert np.abs(dx - dx_t).max() < 1e-9
    assert np.abs(dG - dG_t).max() < 1e-9
    assert np.abs(dB - dB_t).max() < 1e-11


def test_fused_equals_reference_op_order():
    rng = np.random.default_rng(1)
    x = o.synth_activation(rng, (5, 4, 4, 24))
    G, B = o.synth_coloring(rng, 24, 1)
    y, _ = o.wc_forward(x, G[0], B[0])
    assert np.abs(y - o.wc_forward_unfused(x, G[0], B[0])).max() < 1e-11


def test_whitened_covariance_is_identity_as_eps_vanishes():
    rng = np.random.default_rng(2)
    x = o.synth_activation(rng, (16, 8, 8, 16), "well")
    y, cache = o.wc_forward(x, eps=1e-12)
    assert np.abs(np.cov(cache['xhat'].T) - np.eye(16)).max() < 1e-8


def test_row_permutation_equivariance():
    rng = np.random.default_rng(3)
    x = o.synth_activation(rng, (8, 2, 2, 8))
    G, B = o.synth_coloring(rng, 8, 1)
    y, _ = o.wc_forward(x, G, B)
    perm = rng.permutation(8)
    y2, _ = o.wc_forward(x[perm], G, B)
    assert np.abs(y[perm] - y2).max() < 1e-11


def test_eval_mode_uses_moving_statistics_and_has_no_stats_gradient():
    rng = np.random.default_rng(4)
    C = 8
    x = o.synth_activation(rng, (4, 3, 3, C))
    mm, mc = o.moments_to_stats(*o.batch_moments(o.synth_activation(rng, (400, C))))
    G, B = o.synth_coloring(rng, C, 1)
    y, cache = o.wc_forward(x, G, B, training=False, moving_mean=mm, moving_cov=mc)
    L, W = o.whitening_matrix(mc)
    assert np.abs(y.reshape(-1, C) - ((x.reshape(-1, C) - mm) @ (W.T @ G[0]) + B[0])).max() < 1e-12
    gy = rng.standard_normal(x.shape)
    dx, _, _ = o.wc_backward(gy, cache)
    assert np.abs(dx.reshape(-1, C) - gy.reshape(-1, C) @ cache['A'][0].T).max() < 1e-12
    assert 'moving_mean' not in cache


def test_moving_statistics_update():
    rng = np.random.default_rng(5)
    x = o.synth_activation(rng, (4, 3, 3, 8))
    _, cache = o.wc_forward(x, moving_mean=np.zeros(8), moving_cov=np.eye(8), momentum=0.9)
    assert np.allclose(cache['moving_mean'], 0.1 * cache['mu'])
    assert np.allclose(cache['moving_cov'], 0.9 * np.eye(8) + 0.1 * cache['sigma'])


def test_moments_are_additive_across_shards():
    """What sync-WC all-reduces: per-shard (sum, xtx) add up to the full-batch statistics."""
    rng = np.random.default_rng(6)
    X = o.synth_activation(rng, (96, 8))
    s, xtx, M = o.batch_moments(X)
    parts = [o.batch_moments(X[i::3]) for i in range(3)]
    s2 = sum(p[0] for p in parts); xtx2 = sum(p[1] for p in parts); M2 = sum(p[2] for p in parts)
    mu, sig = o.moments_to_stats(s, xtx, M)
    mu2, sig2 = o.moments_to_stats(s2, xtx2, M2)
    assert M == M2 and np.allclose(mu, mu2) and np.allclose(sig, sig2)
    assert np.allclose(sig, np.cov(X.T))


@pytest.mark.parametrize("after_norm", ['ucs', 'ccs', 'uccs', 'uconv', 'fconv', 'ufconv', 'cconv', 'ucconv', 'ccsuconv', 'n'])
def test_coloring_table_matches_branch_composition(after_norm):
    """Each create_norm after_norm value (generator.py:17) equals 'apply every branch to xhat and add'."""
    rng = np.random.default_rng(7)
    C, K, E, N = 6, 3, 2, 5
    p = dict(u_kernel=rng.standard_normal((C, C)), u_bias=rng.standard_normal(C),
             c_kernel=rng.standard_normal((K, C, C)), c_bias=rng.standard_normal((K, C)),
             f_kernel=rng.standard_normal((E, C, C)), f_alpha=rng.standard_normal((K, E)),
             u_gamma=rng.standard_normal(C), u_beta=rng.standard_normal(C),
             c_gamma=rng.standard_normal((K, C)), c_beta=rng.standard_normal((K, C)))
    xh = rng.standard_normal((N, C)); cls = rng.integers(0, K, N)
    G, B = o.coloring_table(after_norm, C, p, K)
    k = cls if G.shape[0] > 1 else np.zeros(N, int)
    got = np.einsum('nc,nco->no', xh, G[k]) + B[k]
    uconv = xh @ p['u_kernel'] + p['u_bias']
    cconv = np.einsum('nc,nco->no', xh, p['c_kernel'][cls]) + p['c_bias'][cls]
    fconv = np.einsum('nc,nco->no', xh, np.einsum('ne,eio->nio', p['f_alpha'][cls], p['f_kernel']))
    ucs = xh * p['u_gamma'] + p['u_beta']
    ccs = xh * p['c_gamma'][cls] + p['c_beta'][cls]
    want = {'ucs': ucs, 'ccs': ccs, 'uccs': ucs + ccs, 'uconv': uconv, 'fconv': fconv, 'ufconv': fconv + uconv,
            'cconv': cconv, 'ucconv': cconv + uconv, 'ccsuconv': ccs + uconv, 'n': xh}[after_norm]
    assert np.abs(got - want).max() < 1e-12


def test_zca_matrix_whitens_symmetrically():
    rng = np.random.default_rng(8)
    X = o.synth_activation(rng, (500, 8), "well")
    _, sig = o.moments_to_stats(*o.batch_moments(X))
    W = o.zca_matrix(sig, 1e-9)
    assert np.abs(W - W.T).max() < 1e-12 and np.abs(W @ sig @ W.T - np.eye(8)).max() < 1e-6


def test_golden_fixtures_reproduce():
    """The committed vectors are what the oracle produces today (guards silent edits of either)."""
    g = np.load(GOLDEN)
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) >= 5
    for n in names:
        a = {k.split('/')[1]: g[k] for k in g.files if k.startswith(n + '/')}
        training = bool(a['training'])
        y, cache = o.wc_forward(a['x'], a['gamma'], a['beta'], a['slot'], training=training,
                                moving_mean=a['mm0'], moving_cov=a['mc0'])
        dx, dG, dB = o.wc_backward(a['gy'], cache)
        for got, want in ((y, a['y']), (dx, a['dx']), (dG, a['dgamma']), (dB, a['dbeta'])):
            assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-30), n
        if training:
            assert np.allclose(cache['moving_cov'], a['mc1'], rtol=1e-6, atol=1e-7)


def test_renorm_oracle_matches_torch_autograd_with_stop_gradient():
    """Row a4: W_eff = L_mov^-1 . stop_gradient(L_b) . L_b^-1 -- value and gradients against torch float64 autograd with
    L_b.detach() in the middle factor."""
    rng = np.random.default_rng(3)
    shape, C, Kc, eps = (6, 4, 4, 16), 16, 3, 1e-3
    x = o.synth_activation(rng, shape)
    G, B = o.synth_coloring(rng, C, Kc)
    slot = rng.integers(0, Kc, shape[0])
    gy = rng.standard_normal(shape)
    ref = o.synth_activation(rng, (500, C))
    mm, mc = o.moments_to_stats(*o.batch_moments(ref))
    y, cache = o.wc_forward_renorm(x, G, B, slot, moving_mean=mm, moving_cov=mc, eps=eps)
    dx, dG, dB = o.wc_backward_renorm(gy, cache)
    eye = torch.eye(C, dtype=torch.float64)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Gt = torch.tensor(G, dtype=torch.float64, requires_grad=True); Bt = torch.tensor(B, dtype=torch.float64, requires_grad=True)
    X = xt.reshape(-1, C); M = X.shape[0]
    f = X - X.mean(0)
    Lb = torch.linalg.cholesky((1 - eps) * (f.T @ f / (M - 1)) + eps * eye)
    Wb = torch.linalg.solve_triangular(Lb, eye, upper=False)
    Lm = torch.linalg.cholesky((1 - eps) * torch.tensor(mc) + eps * eye)
    Wm = torch.linalg.solve_triangular(Lm, eye, upper=False)
    Weff = Wm @ Lb.detach() @ Wb
    it = torch.tensor(slot).long()
    yt = torch.einsum('npc,nco->npo', (f @ Weff.T).reshape(shape[0], -1, C), Gt[it]) + Bt[it][:, None, :]
    yt.backward(torch.tensor(gy).reshape(shape[0], -1, C))
    assert np.abs(y.reshape(yt.shape) - yt.detach().numpy()).max() < 1e-10
    # the value is the moving-statistics whitening of the batch-centred activation
    assert np.abs((f.detach().numpy() @ Wm.numpy().T) - cache['f'] @ cache['W'].T @ cache['C0'].T).max() < 1e-9
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-9
    assert np.abs(dG - Gt.grad.numpy()).max() < 1e-9 and np.abs(dB - Bt.grad.numpy()).max() < 1e-10
    # moving statistics take the batch's update (un-shrunk covariance), as in the plain layer
    mm1, mc1 = o.update_moving(mm, mc, cache['mu'], cache['sigma'])
    assert np.allclose(cache['moving_mean'], mm1) and np.allclose(cache['moving_cov'], mc1)
