"""GPU parity of the pre-split activation format (wc_split.hip, ABI 4): the split itself, and K3 reading it
(wc_apply_split_f16x2) against float64 -- kernel level on the exactly representable input, and end to end against the
oracle's forward at the 1e-4 contract.  The oracle is as unpinned w.r.t. upstream as everywhere else."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


@pytest.fixture(scope="module")
def ops():
    from wc_gan_amd import ops as _ops
    return _ops


def _planes64(xs):
    """float64 value of a SplitTensor: center + (hi + lo) / scale, computed on the host."""
    p = xs.planes.cpu().numpy().astype(np.float64)
    return (p[0] + p[1]) / xs.scale.cpu().numpy().astype(np.float64) + xs.center.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("shape", [(4, 8, 8, 32), (16, 16, 16, 128), (8, 32, 32, 256), (3, 5, 7, 64)])
def test_split_round_trip(ops, shape):
    rng = np.random.default_rng(3)
    C = shape[-1]
    mag = np.exp(rng.uniform(-6, 6, C))
    x = (rng.standard_normal(shape) * mag + 2.5 * mag).astype(np.float32)
    xs = ops.split(dev(x))
    assert int(xs.flag[0]) == 0
    v = _planes64(xs).reshape(shape)
    # 22 significant bits relative to the channel's scaled range: |err| <= 2^-21 of the channel's magnitude
    err = np.abs(v - x.astype(np.float64)) / mag
    assert err.max() < 2.0 ** -19, err.max()
    # per element: relative 2^-21 for anything that is not tiny against its channel
    big = np.abs(x - 2.5 * mag) > 1e-2 * mag
    assert (np.abs(v - x)[big] / np.abs(x - 2.5 * mag)[big]).max() < 2.0 ** -20
    back = ops.unsplit(xs).cpu().numpy()
    assert rel(back, v) < 1e-6
    # relu variant
    xr = ops.split(dev(x), relu=True)
    assert rel(_planes64(xr).reshape(shape), np.maximum(x.astype(np.float64), 0)) < 1e-5


def test_split_overflow_clamps_and_flags(ops):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((4, 16, 16, 64)).astype(np.float32)
    x[1, 3, 3, 5] = 1e7                      # not on a sampled row
    xs = ops.split(dev(x))
    assert int(xs.flag[0]) == 1
    assert torch.isfinite(xs.planes.float()).all()
    y = np.array(x); y[1, 3, 3, 5] = 0
    ys = ops.split(dev(y))
    assert int(ys.flag[0]) == 0


def _ref_apply(xv, mu, A, b, slot):
    N, C = xv.shape[0], xv.shape[-1]
    f = xv.reshape(N, -1, C) - mu.astype(np.float64)
    return np.einsum('npc,nco->npo', f, A.astype(np.float64)[slot]) + b.astype(np.float64)[slot][:, None, :]


CASES = [((16, 32, 32, 256), 1), ((17, 32, 32, 256), 1), ((16, 32, 32, 128), 3), ((33, 24, 24, 128), 1),
         ((128, 32, 32, 256), 1), ((128, 32, 32, 128), 10), ((128, 4, 4, 256), 1), ((128, 8, 8, 256), 7),
         ((128, 4, 4, 128), 10), ((12, 8, 4, 256), 5), ((1, 8, 4, 256), 1)]


@pytest.mark.parametrize("shape,Kc", CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_apply_split_matches_float64(ops, shape, Kc, relu):
    """Kernel level: the reference is float64 arithmetic on the value the planes hold exactly."""
    rng = np.random.default_rng(21)
    N, C = shape[0], shape[-1]
    mag = np.exp(rng.uniform(-6, 6, C))
    x = (rng.standard_normal(shape) * mag + 3 * mag).astype(np.float32)
    mu = (3 * mag + 0.05 * mag * rng.standard_normal(C)).astype(np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C) / mag[None, :, None]).astype(np.float32)
    b = rng.standard_normal((Kc, C)).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    xs = ops.split(dev(x))
    assert ops.apply_split_supported(shape)
    y = ops.apply_split(xs, dev(mu), dev(A), dev(b), st, relu=relu)
    ref = _ref_apply(_planes64(xs).reshape(shape), mu, A, b, slot)
    if relu:
        ref = np.maximum(ref, 0)
    e = rel(y.cpu().numpy().reshape(ref.shape), ref)
    print(shape, Kc, relu, e)
    assert e < 3e-6
    # null centre / null bias / null mu at the ABI
    xs0 = ops.split(dev(x), center=None, scale=xs.scale, flag=xs.flag)
    y0 = ops.apply_split(xs0, None, dev(A), None, st, relu=False)
    p = xs0.planes.cpu().numpy().astype(np.float64)
    v0 = ((p[0] + p[1]) / xs0.scale.cpu().numpy().astype(np.float64)).reshape(shape)
    ref0 = _ref_apply(v0, np.zeros(C), A, np.zeros((Kc, C)), slot)
    assert rel(y0.cpu().numpy().reshape(ref0.shape), ref0) < 3e-6


@pytest.mark.parametrize("shape,Kc", [((128, 32, 32, 256), 1), ((128, 16, 16, 256), 1), ((128, 32, 32, 128), 10)])
@pytest.mark.parametrize("kind", ["ill", "well"])
def test_forward_through_split_meets_the_contract(ops, shape, Kc, kind):
    """End to end: K1 -> K2 -> color (tables for the split tensor's scales) -> K3 on the split input, against the oracle's
    float64 forward of the fp32 input; 1e-4 relative (BASELINE.json north_star)."""
    from oracle import wc_oracle as o
    rng = np.random.default_rng(5)
    N, C = shape[0], shape[-1]
    M = int(np.prod(shape[:-1]))
    x = o.synth_activation(rng, shape, kind).astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    xt = dev(x)
    s, xtx = ops.stats(xt.view(M, C))
    mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, xt.device, want_scale=True)
    xs = ops.split(xt)
    A, At, plan = ops.color(W, dev(G), xs.scale)
    st = dev(slot, torch.int32) if Kc > 1 else None
    y = ops.apply_split(xs, mu, A, dev(B), st, plan=plan)
    y_ref, _ = o.wc_forward(x, G, B, slot)
    e = rel(y.cpu().numpy(), y_ref)
    print(shape, Kc, kind, e)
    assert e < 1e-4


# ---- K1 on the planes (wc_split_xty.hip) ------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,groups", [((20, 32, 32, 256), 1), ((128, 32, 32, 256), 1), ((32, 32, 32, 128), 1), ((128, 32, 32, 128), 1),
                                          ((128, 16, 16, 256), 1), ((320, 32, 32, 256), 5), ((320, 16, 16, 256), 5), ((128, 48, 48, 256), 1)])
def test_stats_split_matches_float64(ops, shape, groups):
    """Kernel level: the moments of the value the planes hold exactly, in float64 on the host (covariance to 1e-7, as
    wc_stats_f32's fast path), exact symmetry, and -- end to end -- the covariance of the fp32 input to 1e-6."""
    from oracle import wc_oracle as o
    rng = np.random.default_rng(31)
    C = shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    xs = ops.split(dev(x))
    M = xs.M
    assert ops.stats_split_supported(M, C, groups)
    s, xtx = ops.stats_split(xs, groups)
    V = _planes64(xs).reshape(groups, M // groups, C)
    sn, xn = s.cpu().numpy().reshape(groups, C), xtx.cpu().numpy().reshape(groups, C, C)
    for g in range(groups):
        s_ref, xtx_ref, Mg = o.batch_moments(V[g])
        _, cov_ref = o.moments_to_stats(s_ref, xtx_ref, Mg)
        _, cov = o.moments_to_stats(sn[g], xn[g], Mg)
        assert rel(sn[g], s_ref) < 1e-5
        e = rel(cov, cov_ref)
        print(shape, groups, g, "cov rel err", e)
        assert e < 1e-7, e
        assert np.abs(xn[g] - xn[g].T).max() == 0.0
        # against the covariance of the fp32 tensor itself: the split's 2^-22 perturbation of x
        X = x.reshape(groups, M // groups, C)[g].astype(np.float64)
        s0, x0, _ = o.batch_moments(X)
        _, cov0 = o.moments_to_stats(s0, x0, Mg)
        assert rel(cov, cov0) < 1e-6


def test_forward_site_on_planes_meets_the_contract(ops):
    """K1 (planes) -> K2 -> color -> K3 (planes) at the headline site against the oracle's forward of the fp32 input."""
    from oracle import wc_oracle as o
    rng = np.random.default_rng(6)
    shape, Kc = (128, 32, 32, 256), 1
    C, M = 256, 128 * 32 * 32
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    xs = ops.split(dev(x))
    s, xtx = ops.stats_split(xs)
    mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, xs.planes.device, want_scale=True)
    A, At, plan = ops.color(W, dev(G), xs.scale)
    be = ops.split_bias(A, dev(B), xs, mu)
    y = ops.apply_split(xs, None, A, be, None, plan=plan, folded=True)
    y_ref, _ = o.wc_forward(x, G, B, np.zeros(128, np.int32))
    e = rel(y.cpu().numpy(), y_ref)
    print("forward on planes", e)
    assert e < 1e-4
