"""GPU parity tests: every stage of the HIP path (through the C ABI) against the float64 oracle.

These pin the BUILD's semantics (oracle/wc_oracle.py restates the published algorithm); they do
not certify equality with the un-vendored upstream layer -- parity with the reference itself is
unpinned.  Tolerance for the float32 path: 1e-4 relative (north_star),
measured as max-abs error over the max-abs of the reference tensor ("rel"), stated per test.
"""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as o

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


@pytest.fixture(scope="module")
def ops():
    from wc_gan_amd import ops as _ops, _lib
    _lib.load()
    return _ops


SHAPES = [(4, 4, 4, 32), (3, 5, 7, 64), (8, 8, 8, 64), (2, 4, 4, 128), (16, 16, 16, 256), (5, 3, 3, 96), (64, 4, 4, 160)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cond", ["ill", "well"])
def test_stats_moments(ops, shape, cond):
    rng = np.random.default_rng(1)
    x = o.synth_activation(rng, shape, cond).astype(np.float32)
    C = shape[-1]
    X = x.reshape(-1, C).astype(np.float64)
    s, xtx = ops.stats(dev(x).view(-1, C))
    s_ref, xtx_ref, M = o.batch_moments(X)
    _, cov_ref = o.moments_to_stats(s_ref, xtx_ref, M)
    _, cov = o.moments_to_stats(s.cpu().numpy(), xtx.cpu().numpy(), M)
    assert rel(s.cpu().numpy(), s_ref) < 1e-5, "channel sums"
    assert rel(cov, cov_ref) < 2e-6, f"covariance rel err {rel(cov, cov_ref)}"
    assert np.abs(xtx.cpu().numpy() - xtx.cpu().numpy().T).max() == 0.0, "xtx must be exactly symmetric"


@pytest.mark.parametrize("C", [32, 64, 96, 128, 160, 256, 512])
def test_factor_cholesky_inverse(ops, C):
    rng = np.random.default_rng(2)
    M = 4 * C + 3
    X = o.synth_activation(rng, (M, C), "ill")
    s, xtx, _ = o.batch_moments(X)
    mu_ref, sigma = o.moments_to_stats(s, xtx, M)
    L_ref, W_ref = o.whitening_matrix(sigma, 1e-3)
    mm = torch.zeros(C, device="cuda"); mc = torch.eye(C, device="cuda")
    mu, L, W = ops.factor(dev(s, torch.float64), dev(xtx, torch.float64), M, C, 1e-3, 0.99, 1, True, mm, mc, "cuda")
    assert rel(mu.cpu().numpy(), mu_ref) < 1e-6
    assert rel(L.cpu().numpy(), L_ref) < 1e-9, f"L rel {rel(L.cpu().numpy(), L_ref)}"
    assert rel(W.cpu().numpy(), W_ref) < 1e-8, f"W rel {rel(W.cpu().numpy(), W_ref)}"
    mm_ref, mc_ref = o.update_moving(np.zeros(C), np.eye(C), mu_ref, sigma, 0.99)
    assert rel(mm.cpu().numpy(), mm_ref) < 1e-6 and rel(mc.cpu().numpy(), mc_ref) < 1e-6


@pytest.mark.parametrize("C,Kc", [(32, 1), (64, 3), (128, 10), (256, 1)])
def test_color_gemm(ops, C, Kc):
    rng = np.random.default_rng(3)
    W = np.tril(rng.standard_normal((C, C)))
    G = rng.standard_normal((Kc, C, C)).astype(np.float32)
    A, At = ops.color(dev(W, torch.float64), dev(G))
    A_ref = np.einsum('ji,kjo->kio', W, G.astype(np.float64))
    assert rel(A.cpu().numpy(), A_ref) < 1e-6
    assert np.array_equal(At.cpu().numpy(), np.transpose(A.cpu().numpy(), (0, 2, 1)))
    A0, At0 = ops.color(dev(W, torch.float64), None)
    assert rel(A0.cpu().numpy()[0], W.T) < 1e-7 and rel(At0.cpu().numpy()[0], W) < 1e-7


@pytest.mark.parametrize("shape,Kc", [((4, 4, 4, 32), 1), ((3, 5, 7, 64), 1), ((6, 4, 4, 128), 3), ((2, 16, 16, 256), 1),
                                      ((5, 9, 9, 96), 5), ((3, 20, 20, 160), 2)])
def test_apply_kernel(ops, shape, Kc):
    rng = np.random.default_rng(4)
    N, C = shape[0], shape[-1]
    x = rng.standard_normal(shape).astype(np.float32)
    mu = rng.standard_normal(C).astype(np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((Kc, C)).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    y = ops.apply(dev(x), dev(mu), dev(A), dev(b), dev(slot, torch.int32) if Kc > 1 else None)
    f = x.astype(np.float64).reshape(N, -1, C) - mu.astype(np.float64)
    ref = np.einsum('npc,nco->npo', f, A.astype(np.float64)[slot]) + b.astype(np.float64)[slot][:, None, :]
    assert rel(y.cpu().numpy().reshape(ref.shape), ref) < 2e-6


VARIANTS = [((4, 4, 4, 32), 1), ((3, 5, 7, 64), 1), ((8, 8, 8, 64), 4), ((6, 4, 4, 128), 10), ((16, 16, 16, 256), 1),
            ((12, 8, 8, 256), 12), ((5, 3, 3, 96), 2)]


@pytest.mark.parametrize("shape,Kc", VARIANTS)
@pytest.mark.parametrize("cond", ["ill", "well"])
def test_forward_backward_train(shape, Kc, cond):
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(5)
    N, C = shape[0], shape[-1]
    x = o.synth_activation(rng, shape, cond).astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B, slot, moving_mean=np.zeros(C), moving_cov=np.eye(C))
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)

    xt = dev(x).requires_grad_(True); Gt = dev(G).requires_grad_(True); Bt = dev(B).requires_grad_(True)
    mm = torch.zeros(C, 1, device="cuda"); mc = torch.eye(C, device="cuda")
    y = whiten_color(xt, Gt, Bt, dev(slot, torch.int32) if Kc > 1 else None, mm, mc, True)
    y.backward(dev(gy))
    errs = dict(y=rel(y.detach().cpu().numpy(), y_ref), dx=rel(xt.grad.cpu().numpy(), dx_ref),
                dG=rel(Gt.grad.cpu().numpy(), dG_ref), dB=rel(Bt.grad.cpu().numpy(), dB_ref),
                mm=rel(mm.cpu().numpy().reshape(-1), cache['moving_mean']), mc=rel(mc.cpu().numpy(), cache['moving_cov']))
    print(shape, Kc, cond, errs)
    assert all(v < TOL for v in errs.values()), errs


@pytest.mark.parametrize("shape,Kc", [((4, 8, 8, 64), 1), ((6, 4, 4, 128), 3)])
def test_forward_backward_eval(shape, Kc):
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(6)
    N, C = shape[0], shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    ref_batch = o.synth_activation(rng, (64 * C, C), "ill")
    mmn, mcn = o.moments_to_stats(*o.batch_moments(ref_batch))
    mmn = mmn.astype(np.float32); mcn = mcn.astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B, slot, training=False, moving_mean=mmn, moving_cov=mcn)
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)
    xt = dev(x).requires_grad_(True); Gt = dev(G).requires_grad_(True); Bt = dev(B).requires_grad_(True)
    mm = dev(mmn).view(C, 1); mc = dev(mcn)
    y = whiten_color(xt, Gt, Bt, dev(slot, torch.int32) if Kc > 1 else None, mm, mc, False)
    y.backward(dev(gy))
    errs = dict(y=rel(y.detach().cpu().numpy(), y_ref), dx=rel(xt.grad.cpu().numpy(), dx_ref),
                dG=rel(Gt.grad.cpu().numpy(), dG_ref), dB=rel(Bt.grad.cpu().numpy(), dB_ref))
    assert all(v < TOL for v in errs.values()), errs
    assert np.array_equal(mm.cpu().numpy().reshape(-1), mmn) and np.array_equal(mc.cpu().numpy(), mcn), "eval must not touch moving stats"


def test_whitening_only_identity_covariance():
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(7)
    shape = (8, 16, 16, 64)
    x = o.synth_activation(rng, shape, "well").astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = whiten_color(xt, None, None, None, None, None, True, eps=1e-6)
    Y = y.detach().cpu().numpy().reshape(-1, 64).astype(np.float64)
    cov = np.cov(Y.T)
    assert np.abs(cov - np.eye(64)).max() < 1e-3
    gy = rng.standard_normal(shape).astype(np.float32)
    y.backward(dev(gy))
    y_ref, cache = o.wc_forward(x, eps=1e-6)
    dx_ref, _, _ = o.wc_backward(gy, cache)
    assert rel(y.detach().cpu().numpy(), y_ref) < TOL and rel(xt.grad.cpu().numpy(), dx_ref) < TOL


def test_abi_rejects_bad_arguments(ops):
    from wc_gan_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 48, device="cuda")
    with pytest.raises(_lib.WcHipError):
        ops.stats(x)                              # C = 48 is not a multiple of 32
    assert lib.wc_apply_f32(None, None, None, None, None, 1, 1, 32, 1, None, None, None, 0, None) == -1
    assert lib.wc_stats_f32(x.data_ptr(), 64, 64, 1, x.data_ptr(), x.data_ptr(), x.data_ptr(), 16, None) == -4
