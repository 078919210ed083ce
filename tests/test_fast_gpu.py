"""GPU parity of the split-fp16 fast affine path (wc_fast.hip) against float64, and of its overflow gate."""
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


def _ref_apply(x, mu, A, b, slot):
    N, C = x.shape[0], x.shape[-1]
    f = x.astype(np.float64).reshape(N, -1, C) - mu.astype(np.float64)
    return np.einsum('npc,nco->npo', f, A.astype(np.float64)[slot]) + b.astype(np.float64)[slot][:, None, :]


CASES = [((16, 32, 32, 256), 1), ((17, 32, 32, 256), 1), ((16, 32, 32, 128), 3), ((64, 16, 16, 64), 4), ((4, 64, 64, 32), 2),
         ((33, 24, 24, 128), 1), ((128, 32, 32, 256), 1), ((128, 32, 32, 128), 10), ((96, 32, 32, 64), 5)]


@pytest.mark.parametrize("shape,Kc", CASES)
def test_fast_apply_matches_float64(ops, shape, Kc):
    rng = np.random.default_rng(11)
    N, C = shape[0], shape[-1]
    chan_scale = np.exp(rng.uniform(-6, 6, C))                      # channels of wildly different magnitude
    x = (rng.standard_normal(shape) * chan_scale + 3 * chan_scale).astype(np.float32)
    mu = (3 * chan_scale).astype(np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C) / chan_scale[None, :, None]).astype(np.float32)
    b = rng.standard_normal((Kc, C)).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    y_fast = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, fast=True)
    y_exact = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, fast=False)
    ref = _ref_apply(x, mu, A, b, slot)
    e_fast, e_exact = rel(y_fast.cpu().numpy().reshape(ref.shape), ref), rel(y_exact.cpu().numpy().reshape(ref.shape), ref)
    print(shape, Kc, "fast", e_fast, "exact f32-MFMA", e_exact)
    assert e_fast < 3e-6 and e_exact < 3e-6


def test_fast_apply_out_of_range_tiles_take_the_exact_path(ops):
    rng = np.random.default_rng(12)
    shape = (16, 32, 32, 128); N, C = 16, 128
    x = rng.standard_normal(shape).astype(np.float32)
    x[5, 7, 9, 3] = 3.0e7            # 3e7 sigma: far outside fp16 after scaling, and not on the sampled rows
    x[11, 1, 2, 77] = -5.0e8
    mu = np.zeros(C, np.float32)
    A = (rng.standard_normal((1, C, C)) / np.sqrt(C)).astype(np.float32)
    b = np.zeros((1, C), np.float32)
    y = ops.apply(dev(x), dev(mu), dev(A), dev(b), None, fast=True)
    ref = _ref_apply(x, mu, A, b, np.zeros(N, int))
    assert torch.isfinite(y).all()
    assert rel(y.cpu().numpy().reshape(ref.shape), ref) < 3e-6
    # the redo path with NO bias and NO centre (null pointers at the ABI), and with both plus per-sample tables
    y0 = ops.apply(dev(x), None, dev(A), None, None, fast=True)
    assert rel(y0.cpu().numpy().reshape(ref.shape), ref) < 3e-6
    A3 = (rng.standard_normal((3, C, C)) / np.sqrt(C)).astype(np.float32)
    b3 = rng.standard_normal((3, C)).astype(np.float32); mu3 = rng.standard_normal(C).astype(np.float32)
    slot = rng.integers(0, 3, N).astype(np.int32)
    y3 = ops.apply(dev(x), dev(mu3), dev(A3), dev(b3), dev(slot, torch.int32), fast=True)
    ref3 = _ref_apply(x, mu3, A3, b3, slot)
    assert rel(y3.cpu().numpy().reshape(ref3.shape), ref3) < 3e-6


@pytest.mark.parametrize("shape,Kc,train", [((16, 32, 32, 256), 1, True), ((16, 32, 32, 128), 3, True), ((32, 32, 32, 64), 1, False),
                                            ((128, 32, 32, 256), 1, True), ((128, 32, 32, 128), 10, True)])
def test_fast_bwd_apply_matches_float64(ops, shape, Kc, train):
    rng = np.random.default_rng(13)
    N, C = shape[0], shape[-1]
    x = rng.standard_normal(shape).astype(np.float32) + 0.5
    gy = (rng.standard_normal(shape) * 1e-3).astype(np.float32)
    mu = np.full(C, 0.5, np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32)
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    S = rng.standard_normal((C, C)).astype(np.float32) * 1e-4; S = (S + S.T) / 2
    gm = (rng.standard_normal(C) * 1e-4).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    args = (dev(gy), dev(x), dev(mu), dev(At), dev(S) if train else None, dev(gm) if train else None, st)
    dx_fast = ops.bwd_apply(*args, fast=True)
    dx_exact = ops.bwd_apply(*args, fast=False)
    g3 = gy.astype(np.float64).reshape(N, -1, C)
    ref = np.einsum('npc,nco->npo', g3, At.astype(np.float64)[slot])
    if train:
        ref = ref + (x.astype(np.float64).reshape(N, -1, C) - mu) @ S.astype(np.float64) - gm.astype(np.float64)
    assert rel(dx_fast.cpu().numpy().reshape(ref.shape), ref) < 3e-6
    assert rel(dx_exact.cpu().numpy().reshape(ref.shape), ref) < 3e-6
    if train:
        # ABI 3: K4 hands its sampled input scales to K6 (three launches instead of six) -- the same samples, the same result
        scales = ops.bwd_reduce(dev(x), dev(mu), dev(gy), st, Kc, want_scales=True)[-1]
        assert scales.shape == (2 * C,) and bool((scales > 0).all())
        dx_shared = ops.bwd_apply(*args, fast=True, scales=scales)
        if C == 256:           # with the scales at hand C = 256 takes the ONE-pass kernel (K = 512): another summation order
            assert rel(dx_shared.cpu().numpy().reshape(ref.shape), ref) < 3e-6
        else:
            assert torch.equal(dx_shared, dx_fast)


@pytest.mark.parametrize("shape", [(16, 32, 32, 256), (32, 32, 32, 128), (96, 32, 32, 64)])
def test_fast_bwd_apply_out_of_range_in_the_accumulating_pass(ops, shape):
    """K6's second pass (dx += (x - mu) S) on the ring kernel adds with atomics, so a tile that holds an element beyond the
    fp16 range must be kept out of the MFMA pass altogether and added exactly: outliers in x (second pass), in gy (first
    pass, second-store-wins redo), in the first and the last tile of a workgroup and in neighbouring tiles."""
    rng = np.random.default_rng(31)
    N, H, _, C = shape
    x = rng.standard_normal(shape).astype(np.float32) + 0.5
    gy = (rng.standard_normal(shape) * 1e-3).astype(np.float32)
    x[0, 0, 0, 1] = 4.0e7; x[0, 1, 3, 5] = -2.0e8; x[N // 2, 7, 9, 3] = 3.0e7; x[N - 1, H - 1, H - 1, C - 1] = 6.0e7
    gy[3, 2, 1, 0] = 5.0e4
    mu = np.full(C, 0.5, np.float32)
    A = (rng.standard_normal((1, C, C)) / np.sqrt(C)).astype(np.float32)
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    S = rng.standard_normal((C, C)).astype(np.float32) * 1e-4; S = (S + S.T) / 2
    gm = (rng.standard_normal(C) * 1e-4).astype(np.float32)
    dx = ops.bwd_apply(dev(gy), dev(x), dev(mu), dev(At), dev(S), dev(gm), None, fast=True)
    scales = ops.bwd_reduce(dev(x), dev(mu), dev(gy), None, 1, want_scales=True)[-1]
    dx_one = ops.bwd_apply(dev(gy), dev(x), dev(mu), dev(At), dev(S), dev(gm), None, fast=True, scales=scales)   # C = 256: one pass
    ref = gy.astype(np.float64).reshape(-1, C) @ At[0].astype(np.float64) \
        + (x.astype(np.float64).reshape(-1, C) - mu) @ S.astype(np.float64) - gm.astype(np.float64)
    got = dx.cpu().numpy().reshape(ref.shape)
    assert np.isfinite(got).all()
    # row-wise: the outlier rows are 1e8 times larger than the others, which must be right too
    err = np.abs(got - ref).max(1) / np.maximum(np.abs(ref).max(1), 1e-30)
    assert err.max() < 1e-5, (err.max(), int(err.argmax()))
    got1 = dx_one.cpu().numpy().reshape(ref.shape)
    assert np.isfinite(got1).all()
    err1 = np.abs(got1 - ref).max(1) / np.maximum(np.abs(ref).max(1), 1e-30)
    assert err1.max() < 1e-5, (err1.max(), int(err1.argmax()))


@pytest.mark.parametrize("shape,Kc", [((32, 32, 32, 256), 1), ((32, 32, 32, 256), 4), ((64, 16, 16, 128), 1), ((8, 8, 8, 256), 1),
                                      ((128, 32, 32, 256), 1)])
def test_bwd_reduce_with_the_relu_mask_in_its_staging(ops, shape, Kc):
    """wc_bwd_reduce_relu_f32: gy := gy where y > 0 applied inside K4 (the quadrant kernel at C = 256 on the fast path, one
    elementwise pass inside the call elsewhere); R, gsum and the masked gradient against float64 of the masked input, and
    bit-identical to masking first and calling the plain function."""
    rng = np.random.default_rng(41)
    N, C = shape[0], shape[-1]
    x = (rng.standard_normal(shape) * np.exp(rng.uniform(-1, 1, C)) + 0.3).astype(np.float32)
    gy = (rng.standard_normal(shape) * 1e-2).astype(np.float32)
    y = rng.standard_normal(shape).astype(np.float32)
    y[0, 0, 0, :8] = 0.0; y[0, 0, 1, :8] = -0.0; y[1, 2, 3, 4] = np.nan          # the edges of `y > 0`
    mu = x.reshape(-1, C).mean(0).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    R, gsum, gm, scales = ops.bwd_reduce(dev(x), dev(mu), dev(gy), st, Kc, want_scales=True, relu_y=dev(y))
    # NaN in y lets the gradient through on every path (the in-kernel mask at C = 256, the elementwise pass elsewhere) -- what
    # aten::threshold_backward does (the three implementations used to disagree there)
    g_ref = np.where(~(y <= 0), gy, np.float32(0))
    assert np.array_equal(gm.cpu().numpy(), g_ref)
    assert gm[1, 2, 3, 4].item() == gy[1, 2, 3, 4]
    tb = torch.ops.aten.threshold_backward(dev(gy), dev(y), 0.0)
    assert torch.equal(tb, gm)
    R2, gsum2 = ops.bwd_reduce(dev(x), dev(mu), dev(g_ref), st, Kc)
    f = x.astype(np.float64).reshape(N, -1, C) - mu.astype(np.float64)
    g = g_ref.astype(np.float64).reshape(N, -1, C)
    for k in range(Kc):
        sel = slot == k if Kc > 1 else np.ones(N, bool)
        R_ref = np.einsum('npi,npj->ij', f[sel], g[sel])
        nat = np.sqrt(np.outer((f[sel] ** 2).sum((0, 1)), (g[sel] ** 2).sum((0, 1)))) + 1e-300
        assert np.abs((R[k].cpu().numpy() - R_ref) / nat).max() < 1e-6
        assert rel(gsum[k].cpu().numpy(), g[sel].sum((0, 1))) < 1e-5
    # the masked-in-staging result against mask-first: same products, same order
    assert rel(R.cpu().numpy(), R2.cpu().numpy()) < 1e-6 and bool((scales > 0).all())


# ---- fast reductions (wc_fast_xty.hip) -------------------------------------------------------------------------
XTY_CASES = [((16, 32, 32, 256), 1), ((128, 32, 32, 256), 1), ((32, 32, 32, 128), 1), ((16, 64, 64, 64), 1), ((8, 64, 64, 32), 1)]


@pytest.mark.parametrize("shape,_k", XTY_CASES)
def test_fast_stats_matches_float64(ops, shape, _k):
    from oracle import wc_oracle as o
    rng = np.random.default_rng(21)
    C = shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    X = x.reshape(-1, C).astype(np.float64)
    s, xtx = ops.stats(dev(x).view(-1, C))
    s_ref, xtx_ref, M = o.batch_moments(X)
    _, cov_ref = o.moments_to_stats(s_ref, xtx_ref, M)
    _, cov = o.moments_to_stats(s.cpu().numpy(), xtx.cpu().numpy(), M)
    assert rel(s.cpu().numpy(), s_ref) < 1e-5
    e = rel(cov, cov_ref)
    print(shape, "cov rel err", e)
    # float64-flushed split-fp16 products.  What remains is the dropped lo*lo term: a ~3e-8 relative, nearly uniform
    # inflation of the diagonal (a ridge 4 orders below eps) -- an fp32-accumulated covariance sits at 1e-6.
    assert e < 1e-7, e
    assert np.abs(xtx.cpu().numpy() - xtx.cpu().numpy().T).max() == 0.0


@pytest.mark.parametrize("shape,Kc", [((16, 32, 32, 256), 1), ((32, 32, 32, 128), 5), ((128, 32, 32, 128), 10), ((16, 64, 64, 64), 3),
                                      ((32, 32, 32, 256), 4), ((128, 16, 16, 256), 10)])      # C = 256 with class slots: the quadrant scheme per sample
def test_fast_bwd_reduce_matches_float64(ops, shape, Kc):
    rng = np.random.default_rng(22)
    N, C = shape[0], shape[-1]
    x = (rng.standard_normal(shape) * np.exp(rng.uniform(-3, 3, C)) + 0.3).astype(np.float32)
    gy = (rng.standard_normal(shape) * 1e-3 * np.exp(rng.uniform(-3, 3, C))).astype(np.float32)
    mu = x.reshape(-1, C).mean(0).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    R, gsum = ops.bwd_reduce(dev(x), dev(mu), dev(gy), dev(slot, torch.int32) if Kc > 1 else None, Kc)
    f = x.astype(np.float64).reshape(N, -1, C) - mu.astype(np.float64)
    g = gy.astype(np.float64).reshape(N, -1, C)
    for k in range(Kc):
        sel = slot == k if Kc > 1 else np.ones(N, bool)
        R_ref = np.einsum('npi,npj->ij', f[sel], g[sel])
        scale = np.sqrt(np.outer((f[sel] ** 2).sum((0, 1)), (g[sel] ** 2).sum((0, 1)))) + 1e-300   # per-entry natural scale
        assert np.abs((R[k].cpu().numpy() - R_ref) / scale).max() < 1e-7
        assert rel(gsum[k].cpu().numpy(), g[sel].sum((0, 1))) < 1e-5


@pytest.mark.parametrize("shape", [(32, 32, 32, 256), (64, 32, 32, 128), (16, 64, 64, 64)])
def test_fast_reductions_out_of_range_take_the_exact_redo(ops, shape):
    """One activation beyond the fp16 range of the scaled operands: the fast K1 / K4 kernels raise their gate and the gated
    exact kernels (a few workgroups walking every (slab, tile) pair) replace all partials."""
    rng = np.random.default_rng(23)
    N, C = shape[0], shape[-1]
    x = (rng.standard_normal(shape) + 0.3).astype(np.float32)
    x[1, 2, 3, 5] = 4.0e7
    x[0, 0, 0, 9] = -6.0e7                       # on a SAMPLED row (row 0): the scales and the shift come from 256 sampled rows
    gy = (rng.standard_normal(shape) * 1e-3).astype(np.float32)
    gy[2, 1, 0, 7] = 3.0e6
    gy[0, 0, 0, 11] = 2.0e5                      # sampled row too
    X = x.reshape(-1, C).astype(np.float64)
    s, xtx = ops.stats(dev(x).view(-1, C))
    nat = np.sqrt(np.outer((X ** 2).sum(0), (X ** 2).sum(0)))
    assert np.abs((xtx.cpu().numpy() - X.T @ X) / nat).max() < 1e-6
    assert np.abs((s.cpu().numpy() - X.sum(0)) / np.sqrt((X ** 2).sum(0) * X.shape[0])).max() < 1e-6
    mu = x.reshape(-1, C).mean(0).astype(np.float32)
    R, gsum = ops.bwd_reduce(dev(x), dev(mu), dev(gy), None, 1)
    f = X - mu.astype(np.float64)
    g = gy.astype(np.float64).reshape(-1, C)
    nat = np.sqrt(np.outer((f ** 2).sum(0), (g ** 2).sum(0)))
    assert np.abs((R[0].cpu().numpy() - f.T @ g) / nat).max() < 1e-6
    assert np.abs((gsum[0].cpu().numpy() - g.sum(0)) / np.sqrt((g ** 2).sum(0) * g.shape[0])).max() < 1e-6


@pytest.mark.parametrize("shape,Kc", [((16, 32, 32, 256), 1), ((17, 32, 32, 256), 1), ((128, 32, 32, 128), 10), ((4, 64, 64, 32), 2),
                                      ((8, 8, 8, 64), 3), ((33, 24, 24, 128), 1)])
def test_apply_with_fused_relu(ops, shape, Kc):
    """wc_apply_act_f32: max(y, 0) from every K3 path -- planned/unplanned fast, exact, slots, redo."""
    rng = np.random.default_rng(21)
    N, C = shape[0], shape[-1]
    x = rng.standard_normal(shape).astype(np.float32)
    x[1, 2, 3, 5] = 4.0e7                        # one tile takes the exact redo
    mu = rng.standard_normal(C).astype(np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((Kc, C)).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    ref = np.maximum(_ref_apply(x, mu, A, b, slot), 0.0)
    for fast in (True, False):
        y = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, fast=fast, relu=True)
        assert float(y.min()) >= 0.0
        assert rel(y.cpu().numpy().reshape(ref.shape), ref) < 3e-6


@pytest.mark.parametrize("shape", [(16, 32, 32, 256), (16, 32, 32, 128), (8, 8, 8, 64)])
def test_fused_relu_keeps_nan(ops, shape):
    """relu(NaN) = NaN, as torch.relu: the epilogue's ReLU is !(v <= 0) ? v : 0, not max(v, 0) (which would return 0)"""
    rng = np.random.default_rng(22)
    C = shape[-1]
    x = rng.standard_normal(shape).astype(np.float32)
    mu = rng.standard_normal(C).astype(np.float32)
    A = (rng.standard_normal((1, C, C)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((1, C)).astype(np.float32)
    b[0, 7] = np.nan                                  # the bias enters in the epilogue only: the fast path stays the fast path
    for fast in (True, False):
        y = ops.apply(dev(x), dev(mu), dev(A), dev(b), None, fast=fast, relu=True)
        assert bool(torch.isnan(y[..., 7]).all())
        rest = torch.cat([y[..., :7], y[..., 8:]], dim=-1)
        assert not bool(torch.isnan(rest).any()) and float(rest.min()) >= 0.0


# ---- the ReLU's gradient mask as one bit per element (wc_apply_mask_f32 / wc_bwd_reduce_mask_f32) -------------------------------
def _mask_ref(y):
    """(M/32, C) words: bit b of word (t, c) = y[32 t + b, c] passed (> 0, or NaN)"""
    M, C = y.shape
    keep = ~(y <= 0)
    w = (keep.reshape(M // 32, 32, C).astype(np.uint64) << np.arange(32, dtype=np.uint64)[None, :, None]).sum(1)
    return w.astype(np.uint32)


@pytest.mark.parametrize("shape,Kc", [((128, 32, 32, 256), 1), ((16, 32, 32, 256), 1), ((64, 16, 16, 128), 5), ((128, 4, 4, 256), 1),
                                      ((12, 8, 4, 96), 1), ((17, 8, 8, 64), 3)])
def test_apply_leaves_the_relu_bit_mask(ops, shape, Kc):
    """Every path of K3 (planned ring kernel, its per-row redo for straddling slots, the f32-MFMA kernel for widths the fast
    path does not take) gives the same y as wc_apply_act_f32 and the mask that y implies, bit for bit -- NaN and +-0 included."""
    rng = np.random.default_rng(51)
    N, C = shape[0], shape[-1]
    M = int(np.prod(shape[:-1]))
    x = rng.standard_normal(shape).astype(np.float32)
    x[0, 0, 0, 3] = np.nan                       # a NaN row: NaN in every output of that row -> its bits are set
    mu = (0.1 * rng.standard_normal(C)).astype(np.float32)
    A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32)
    A[:, :, 5] = 0.0                             # an all-zero output column with zero bias: exact +0 outputs -> bits clear
    b = rng.standard_normal((Kc, C)).astype(np.float32); b[:, 5] = 0.0
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    s_, xtx_ = ops.stats(dev(np.nan_to_num(x)).view(M, C))
    _, _, W, cs = ops.factor(s_, xtx_, M, C, 1e-3, 0.99, 1, True, None, None, "cuda", want_scale=True)
    for plan in (ops.color(W, dev(A), cs)[2] if C in (32, 64, 128, 256) else None, None):
        y_ref = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, plan=plan, relu=True)
        y, mask = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, plan=plan, relu=True, want_mask=True)
        yn = y.cpu().numpy().reshape(M, C)
        assert np.array_equal(yn, y_ref.cpu().numpy().reshape(M, C), equal_nan=True)
        assert np.array_equal(mask.cpu().numpy().view(np.uint32), _mask_ref(yn))
        if not np.isnan(yn[1:, 5]).any():
            assert (yn[1:, 5] == 0).all()
    # the elementwise consumer of the bits
    gy = rng.standard_normal(shape).astype(np.float32)
    out = ops.relu_mask_bits(dev(gy), mask)
    assert np.array_equal(out.cpu().numpy().reshape(M, C), np.where(~(yn <= 0), gy.reshape(M, C), np.float32(0)))


@pytest.mark.parametrize("shape,Kc", [((32, 32, 32, 256), 1), ((32, 32, 32, 256), 4), ((64, 16, 16, 128), 1), ((128, 32, 32, 256), 1)])
def test_bwd_reduce_with_the_bit_mask(ops, shape, Kc):
    """wc_bwd_reduce_mask_f32 == wc_bwd_reduce_relu_f32 given the y the mask came from: R, gsum, scales and the masked gradient
    bit for bit (same products in the same order; the quadrant kernel at C = 256 reads one mask word per column and 32 rows)."""
    rng = np.random.default_rng(52)
    N, C = shape[0], shape[-1]
    M = int(np.prod(shape[:-1]))
    x = (rng.standard_normal(shape) * np.exp(rng.uniform(-1, 1, C)) + 0.3).astype(np.float32)
    gy = (rng.standard_normal(shape) * 1e-2).astype(np.float32)
    y = rng.standard_normal(shape).astype(np.float32)
    y[0, 0, 0, :8] = 0.0; y[0, 0, 1, :8] = -0.0; y[1, 2, 3, 4] = np.nan
    mu = x.reshape(-1, C).mean(0).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    mask = dev(_mask_ref(y.reshape(M, C)).view(np.int32), torch.int32)
    R1, g1, gm1, sc1 = ops.bwd_reduce(dev(x), dev(mu), dev(gy), st, Kc, want_scales=True, relu_y=dev(y))
    R2, g2, gm2, sc2 = ops.bwd_reduce(dev(x), dev(mu), dev(gy), st, Kc, want_scales=True, relu_mask=mask)
    assert torch.equal(gm1, gm2) and torch.equal(sc1, sc2)
    assert torch.equal(R1, R2) and torch.equal(g1, g2)


@pytest.mark.parametrize("shape,groups", [((128, 32, 32, 256), 1), ((320, 16, 16, 256), 5), ((64, 8, 8, 128), 1), ((40, 8, 8, 64), 5),
                                          ((16, 8, 8, 256), 1), ((12, 6, 6, 96), 3)])
def test_whiten_is_stats_then_factor_bit_for_bit(shape, groups):
    """wc_whiten_f32 (K1 + K2 as one call: the K1 tail's slab reduction and the K2 head's bookkeeping in ONE launch, the moments never
    stored) returns exactly what wc_stats_f32 followed by wc_factor_f64 returns -- mu, L, W, chan_scale and both moving statistics,
    over several statistic groups too (their moving-statistics updates are applied one after the other)."""
    from oracle import wc_oracle as o
    from wc_gan_amd import ops
    rng = np.random.default_rng(41)
    C = shape[-1]
    x = dev(o.synth_activation(rng, shape, "ill").astype(np.float32))
    M = x.numel() // C
    mm1 = torch.randn(C, device="cuda") * 0.1; mc1 = torch.eye(C, device="cuda") * 1.5
    mm2, mc2 = mm1.clone(), mc1.clone()
    s, xtx = ops.stats(x.view(M, C), groups)
    mu1, L1, W1, cs1 = ops.factor(s, xtx, M // groups, C, 1e-3, 0.99, 1, True, mm1, mc1, x.device, want_scale=True, groups=groups)
    mu2, L2, W2, cs2 = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, mm2, mc2, groups)
    torch.cuda.synchronize()
    for a, b in ((mu1, mu2), (torch.tril(L1), torch.tril(L2)), (W1, W2), (cs1, cs2), (mm1, mm2), (mc1, mc2)):
        assert torch.equal(a, b)


def test_k2_error_words_are_clear_after_a_clean_call():
    """the one-launch K2 leaves a sticky error word per statistic group when its bounded wait runs out (and poisons W);
    wc_factor_error_offset / wc_whiten_error_offset say where.  With WC_CHECK_K2 the wrappers read them back: clean calls pass."""
    from oracle import wc_oracle as o
    from wc_gan_amd import _lib, ops
    lib = _lib.load()
    assert lib.wc_factor_error_offset(256, 5) == 5 * 256 * 16 * 8 + 4
    assert lib.wc_factor_error_offset(64, 1) == 0                       # two launches at this width: no wait, nothing to check
    x = dev(o.synth_activation(np.random.default_rng(3), (40, 8, 8, 256), "ill").astype(np.float32))
    prev = ops.CHECK_K2
    ops.CHECK_K2 = True
    try:
        mu, L, W, cs = ops.whiten(x.view(-1, 256), 1e-3, 0.99, 1, None, None, 5)
        s, xtx = ops.stats(x.view(-1, 256), 5)
        mu2, L2, W2 = ops.factor(s, xtx, 512, 256, 1e-3, 0.99, 1, True, None, None, x.device, groups=5)
    finally:
        ops.CHECK_K2 = prev
    assert bool(torch.isfinite(W).all()) and torch.equal(W, W2)


@pytest.mark.parametrize("shape,Kc", [((128, 32, 32, 256), 1), ((128, 16, 16, 256), 10), ((128, 12, 12, 256), 7), ((72, 16, 16, 256), 1),
                                      ((136, 12, 12, 256), 1), ((136, 12, 12, 256), 5), ((320, 8, 8, 256), 1)])
def test_relu_backward_without_a_masked_copy_equals_the_one_with(shape, Kc):
    """VERDICT r2 item 3, second half: K4 applies the one-bit mask in its staging and writes NO masked gradient
    (wc_bwd_reduce_bits_f32), K6 applies the same bits while it converts gy (wc_bwd_apply_bits_f32).  R, gsum and the scales are
    bit-identical to the route that writes the copy, dx equal to 1e-6 of its maximum (K6 converts gy*bit instead of the stored
    product: the same numbers), for plain and per-class tables, uneven tile counts per workgroup pair (9 and 10) and tiles that
    straddle samples of different slots."""
    from oracle import wc_oracle as o
    from wc_gan_amd import ops
    rng = np.random.default_rng(5)
    N, C = shape[0], shape[-1]
    x = dev(o.synth_activation(rng, shape, "ill").astype(np.float32))
    gy = dev(rng.standard_normal(shape).astype(np.float32))
    G, B = o.synth_coloring(rng, C, Kc)
    Gd, Bd = dev(G.astype(np.float32)), dev(B.astype(np.float32))
    slot = dev(rng.integers(0, Kc, N).astype(np.int32), torch.int32) if Kc > 1 else None
    M = x.numel() // C
    if not ops.bwd_bits_supported(shape, slot is not None):
        assert shape not in ((128, 32, 32, 256), (128, 16, 16, 256))          # the generator's own sites must be on this route
        pytest.skip("no bits-only route for this shape (the per-sample reduction takes another kernel): the copy is written")
    mu, L, W, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
    A, At, plan = ops.color(W, Gd, cs)
    y, mask = ops.apply(x, mu, A, Bd, slot, plan=plan, relu=True, want_mask=True)
    R1, g1, gm, sc1 = ops.bwd_reduce(x, mu, gy, slot, Kc, want_scales=True, relu_mask=mask)
    R2, g2, sc2 = ops.bwd_reduce(x, mu, gy, slot, Kc, want_scales=True, relu_mask=mask, write_masked=False)
    assert torch.equal(R1, R2) and torch.equal(g1, g2) and torch.equal(sc1, sc2)
    _, _, S, gmean = ops.bwd_factor(R1, g1, W, L, Gd, A, M, 1e-3, 1, True)
    dx1 = ops.bwd_apply(gm, x, mu, At, S, gmean, slot, scales=sc1)
    dx2 = ops.bwd_apply(gy, x, mu, At, S, gmean, slot, scales=sc2, relu_mask=mask)
    torch.cuda.synchronize()
    assert float((dx1 - dx2).abs().max()) <= 1e-6 * float(dx1.abs().max())


def test_relu_backward_bits_route_redoes_out_of_range_tiles_exactly():
    """An element of gy far outside the sampled fp16 range: both kernels take their exact path for that tile -- K4's gated redo
    masks while it loads, K6's row-wise redo masks per element -- and the two routes still agree."""
    from oracle import wc_oracle as o
    from wc_gan_amd import ops
    rng = np.random.default_rng(6)
    shape = (96, 16, 16, 256); C = 256
    assert ops.bwd_bits_supported(shape, False)
    x = dev(o.synth_activation(rng, shape, "ill").astype(np.float32))
    gy = rng.standard_normal(shape).astype(np.float32)
    gy[5, 3, 3, 17] = 3e7; gy[9, 1, 2, 200] = -2e7; gy[77, 15, 15, 0] = 1e7
    gy = dev(gy)
    M = x.numel() // C
    mu, L, W, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
    A, At, plan = ops.color(W, None, cs)
    y, mask = ops.apply(x, mu, A, None, None, plan=plan, relu=True, want_mask=True)
    R1, g1, gm, sc1 = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask)
    R2, g2, sc2 = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask, write_masked=False)
    assert float((R1 - R2).abs().max()) <= 1e-9 * float(R1.abs().max()) and float((g1 - g2).abs().max()) <= 1e-9 * float(g1.abs().max())
    _, _, S, gmean = ops.bwd_factor(R1, g1, W, L, None, A, M, 1e-3, 1, True)
    dx1 = ops.bwd_apply(gm, x, mu, At, S, gmean, None, scales=sc1)
    dx2 = ops.bwd_apply(gy, x, mu, At, S, gmean, None, scales=sc2, relu_mask=mask)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx2).all())
    assert float((dx1 - dx2).abs().max()) <= 1e-6 * float(dx1.abs().max())
