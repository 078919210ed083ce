"""GPU parity at the FULL site sizes of the four GPU configurations in BASELINE.json:configs, through the C ABI, against
the float64 oracle -- plus one training step of each configuration.

  CIFAR-10 uncond   scripts/cifar10_resnet_sn_uncond.sh   C = 256, sites 4..32,  headline site (128, 32, 32, 256)
  CIFAR-10 cond     scripts/cifar10_resnet_sn_cond.sh     C = 128, ucconv, 10 classes; critic-phase passes grouped x5
  STL-10 uncond     scripts/stl10_resnet_sn_uncond.sh     C = 256, sites 6..48 (run.py:152,333)
  Tiny-ImageNet     scripts/tinyimagenet_resnet_sn_cond_sa.sh   C = 128, ufconv, 200 classes, filters_emb 15, sites 4..64
                    (run.py:155-158,172-173,335): more classes than samples -> per-sample coloring tables

Tolerance: 1e-4 relative (north_star), max-abs error over max-abs reference, float32 path against float64.  Parity with
the un-vendored upstream layer itself is unpinned; these pin the build's semantics.
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


# (shape, conditioning, Kc): whole-site forward + backward, as the generator update runs it (N = 128) and as one
# critic-phase pass runs it (N = 64)
FULL_SITES = [
    ((128, 32, 32, 256), "ill", 1),      # the north-star site, cond(Sigma) ~ 1e6, end to end
    ((128, 32, 32, 256), "well", 1),
    ((128, 48, 48, 256), "ill", 1),      # STL-10 final site, 288 MiB
    ((64, 12, 12, 256), "ill", 1),       # STL-10, one critic-phase pass
    ((128, 12, 12, 256), "ill", 1),      # STL-10, generator update
    ((128, 32, 32, 128), "ill", 10),     # CIFAR-10 cond, per-class tables
    ((128, 8, 8, 128), "ill", 10),       # CIFAR-10 cond 8x8: HW = 64 rows per sample
    ((128, 12, 12, 256), "ill", 7),      # per-class tables with HW = 144: tiles straddle samples of different slots
]


@pytest.mark.parametrize("shape,cond,Kc", FULL_SITES)
def test_full_size_site_forward_backward(shape, cond, Kc):
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(11)
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
    print(shape, cond, Kc, errs)
    assert all(v < TOL for v in errs.values()), errs


@pytest.mark.parametrize("shape,seed", [((128, 32, 32, 256), 100), ((128, 32, 32, 256), 103), ((128, 12, 12, 256), 101), ((128, 16, 16, 256), 102)])
def test_seed_sweep_cases_hold_the_contract_with_margin(shape, seed):
    """The (site, seed) pairs of tools/seed_sweep.py that decided round 3's parity fix: 128x32x32x256 seed 100
    read dx 1.39e-4 and 128x12x12x256 seed 101 1.03e-4 against the 1e-4 contract, all of it the matrix pipe's bias on the covariance's
    off-diagonal sums; with the bias compensated (wc_fast_xty.hip kXtyOffdiagBias) they read 3.2e-5 / 7.9e-5.  Bound here: 9e-5."""
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(seed)
    C = shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1)
    G = G.astype(np.float32); B = B.astype(np.float32)
    rng.integers(0, 1, shape[0])                      # (keeps the generator in step with tools/seed_sweep.py)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B, None)
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)
    xt = dev(x).requires_grad_(True); Gt = dev(G).requires_grad_(True); Bt = dev(B).requires_grad_(True)
    y = whiten_color(xt, Gt, Bt, None, None, None, True)
    y.backward(dev(gy))
    errs = dict(y=rel(y.detach().cpu().numpy(), y_ref), dx=rel(xt.grad.cpu().numpy(), dx_ref), dG=rel(Gt.grad.cpu().numpy(), dG_ref),
                dB=rel(Bt.grad.cpu().numpy(), dB_ref))
    assert all(v < 9e-5 for v in errs.values()), errs


@pytest.mark.parametrize("family,shape", [("uniform", (128, 32, 32, 256)), ("relu", (128, 16, 16, 256)), ("heavy", (128, 32, 32, 256)),
                                          ("uniform", (128, 32, 32, 128)), ("relu", (128, 32, 32, 64))])
def test_non_gaussian_families_hold_their_documented_bounds(family, shape):
    """the off-diagonal bias compensation of K1 (kXtyOffdiagBias) was fitted on gaussian mixes -- full-size,
    cond(Sigma~) ~ 1e6 cases on the families where it could over-correct: uniform, post-ReLU half-sparse and heavy-tailed (Laplace)
    elements, C = 256 / 128 / 64 (oracle.synth_activation_family).  These inputs are HARDER than the contract's kernel-bench input (the
    whitening cancels terms of size ~500 into results of size 1, ~30 there), and the arithmetic the contract names -- the reference's op
    order in fp32 -- is itself 5e-3 .. 1e-2 (y) and 1e-2 .. 3e-2 (dx) from float64 on them (tools/seed_sweep.py --families --ref32,
    profiles/r4_seed_sweep.txt).  Documented bounds of this build over 3 seeds x 4 sites per family: y <= 1.2e-4, dx <= 3.6e-4, dGamma
    <= 1.9e-4 with the compensation (dx <= 4.5e-4 without: it helps on every family, by 20-35 %).  Asserted here with a margin: y 2e-4,
    dx 5e-4, dGamma 3e-4 -- 50 x closer to float64 than the fp32 reference order."""
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(100)
    C = shape[-1]
    x = o.synth_activation_family(rng, shape, family).astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1)
    G = G.astype(np.float32); B = B.astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B, None)
    ev = np.linalg.eigvalsh((1 - 1e-3) * cache['sigma'] + 1e-3 * np.eye(C))
    assert ev[-1] / ev[0] > (1e6 if C == 256 else 3e5)
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)
    xt = dev(x).requires_grad_(True); Gt = dev(G).requires_grad_(True); Bt = dev(B).requires_grad_(True)
    y = whiten_color(xt, Gt, Bt, None, None, None, True)
    y.backward(dev(gy))
    errs = dict(y=rel(y.detach().cpu().numpy(), y_ref), dx=rel(xt.grad.cpu().numpy(), dx_ref), dG=rel(Gt.grad.cpu().numpy(), dG_ref),
                dB=rel(Bt.grad.cpu().numpy(), dB_ref))
    print(family, shape, errs)
    assert errs['y'] < 2e-4 and errs['dx'] < 5e-4 and errs['dG'] < 3e-4 and errs['dB'] < 1e-5, errs


def _remaining_sites():
    """Every (shape, Kc) of the four configurations' generator sites at the generator update's batch (N = 128), taken from
    train.wc_sites() so that the list cannot drift from the recipes, minus what FULL_SITES above already runs.  Kc: block
    sites of the conditional recipes carry per-class tables (10 classes) or -- 200 classes > 128 samples -- one table per
    sample; the final site is always unconditional (generator.py:154)."""
    from wc_gan_amd.train import CONFIGS, wc_sites
    done = {(sh, kc) for sh, _, kc in FULL_SITES}
    out = []
    for name, cfg in CONFIGS.items():
        K = cfg['generator']['number_of_classes'] if cfg['conditional'] else 1
        for site, N, H, W, C in wc_sites(cfg, 128):
            kc = 1 if (site.endswith('Final') or K == 1) else (K if K <= N else N)
            key = ((N, H, W, C), kc)
            if key not in done:
                done.add(key)
                out.append(pytest.param((N, H, W, C), kc, id=f"{name}:{site}:{H}x{W}x{C}:Kc{kc}"))
    return out


@pytest.mark.parametrize("shape,Kc", _remaining_sites())
def test_every_remaining_site_forward_backward(shape, Kc):
    """VERDICT r2 'parity depth': the site shapes that were exercised only through the property-style step test, forward +
    backward + moving statistics against the oracle at full size, cond(Sigma) ~ 1e6."""
    from wc_gan_amd.functional import whiten_color
    rng = np.random.default_rng(23)
    N, C = shape[0], shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = (np.arange(N) if Kc == N else rng.integers(0, Kc, N)).astype(np.int32)
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
    print(shape, Kc, errs)
    assert all(v < TOL for v in errs.values()), errs


@pytest.mark.parametrize("shape,Kc", [((64, 32, 32, 256), 1), ((64, 48, 48, 256), 1), ((64, 32, 32, 128), 10)])
def test_full_size_eval_mode(shape, Kc):
    """Evaluation mode at full size (scorer.py:60,72 feed batches of 64 with learning_phase = False): moving statistics in,
    no updates, the cached plan on the second call -- against the oracle's eval forward."""
    from wc_gan_amd.functional import EvalPlan, whiten_color_eval_cached
    rng = np.random.default_rng(29)
    N, C = shape[0], shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    # moving statistics of a "trained" layer: the batch statistics of another draw of the same distribution
    x0 = o.synth_activation(np.random.default_rng(30), (32,) + shape[1:], "ill").astype(np.float64).reshape(-1, C)
    mm = x0.mean(0).astype(np.float32); mc = np.cov(x0, rowvar=False).astype(np.float32)
    y_ref, _ = o.wc_forward(x, G, B, slot, moving_mean=mm.astype(np.float64), moving_cov=mc.astype(np.float64), training=False)
    mmt, mct = dev(mm.reshape(C, 1)), dev(mc)
    cache = EvalPlan()
    st = dev(slot, torch.int32) if Kc > 1 else None
    for _ in range(2):      # the second call is served by the cached factorisation
        y = whiten_color_eval_cached(dev(x), cache, dev(G), dev(B), st, mmt, mct, 1e-3)
        assert rel(y.cpu().numpy(), y_ref) < TOL
    assert torch.equal(mmt.cpu(), torch.tensor(mm.reshape(C, 1))) and torch.equal(mct.cpu(), torch.tensor(mc))


def test_full_size_relu_epilogue_and_mask():
    """The site as the generator runs it (generator.py:144-151: norm -> relu): relu folded into the apply, its mask
    into the backward."""
    from wc_gan_amd.functional import whiten_color
    shape = (128, 32, 32, 256)
    rng = np.random.default_rng(12)
    C = shape[-1]
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1)
    G = G.astype(np.float32); B = B.astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    y_ref, cache = o.wc_forward(x, G, B)
    xt = dev(x).requires_grad_(True); Gt = dev(G).requires_grad_(True); Bt = dev(B).requires_grad_(True)
    y = whiten_color(xt, Gt, Bt, None, None, None, True, relu=True)
    y.backward(dev(gy))
    yn = y.detach().cpu().numpy()
    assert rel(yn, np.maximum(y_ref, 0)) < TOL
    # the mask is a discontinuous function of y: elements within the forward error of zero may fall on either side
    # (a few thousand of 33 million here), everywhere else the two masks must agree ...
    sure = np.abs(y_ref) > TOL * np.abs(y_ref).max()
    assert np.array_equal((yn > 0)[sure], (y_ref > 0)[sure])
    assert (~sure).mean() < 1e-3
    # ... and the backward is checked for the mask the forward actually produced
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy * (yn > 0), cache)
    errs = dict(dx=rel(xt.grad.cpu().numpy(), dx_ref), dG=rel(Gt.grad.cpu().numpy(), dG_ref), dB=rel(Bt.grad.cpu().numpy(), dB_ref))
    print(errs)
    assert all(v < TOL for v in errs.values()), errs


# critic-phase passes: `groups` independent batches of 64 in one call (train.GanTrainer.generate), forward only
GROUPED = [
    ((320, 12, 12, 256), 5, 1),          # STL-10: HW = 144 is no multiple of the 32-row tile
    ((320, 24, 24, 256), 5, 1),
    ((320, 8, 8, 128), 5, 10),           # CIFAR-10 cond: slot = group*10 + class, 64 rows per sample
    ((320, 32, 32, 128), 5, 10),
    ((320, 32, 32, 256), 5, 1),          # CIFAR-10 uncond, the headline critic-phase site
]


@pytest.mark.parametrize("shape,groups,Kc", GROUPED)
def test_full_size_grouped_forward(shape, groups, Kc):
    from wc_gan_amd.functional import whiten_color_grouped
    rng = np.random.default_rng(13)
    N, C = shape[0], shape[-1]
    n = N // groups
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    x *= np.repeat(1.0 + 0.25 * np.arange(groups), n).astype(np.float32)[:, None, None, None]      # groups differ in scale
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    mm_ref, mc_ref = np.zeros(C), np.eye(C)
    y_ref = np.empty(shape)
    for g in range(groups):                      # `groups` separate passes, moving statistics updated one after the other
        sl = slice(g * n, (g + 1) * n)
        y_ref[sl], cache = o.wc_forward(x[sl], G, B, slot[sl], moving_mean=mm_ref, moving_cov=mc_ref)
        mm_ref, mc_ref = cache['moving_mean'], cache['moving_cov']
    mm = torch.zeros(C, 1, device="cuda"); mc = torch.eye(C, device="cuda")
    y = whiten_color_grouped(dev(x), groups, dev(G), dev(B), dev(slot, torch.int32) if Kc > 1 else None, mm, mc)
    errs = dict(y=rel(y.cpu().numpy(), y_ref), mm=rel(mm.cpu().numpy().reshape(-1), mm_ref), mc=rel(mc.cpu().numpy(), mc_ref))
    print(shape, groups, Kc, errs)
    assert all(v < TOL for v in errs.values()), errs


def _ufconv_params(stack):
    fc, uc = stack.branches
    return dict(f_kernel=fc.kernel.detach().cpu().numpy().astype(np.float64),
                f_alpha=fc.class_matrix.detach().cpu().numpy().astype(np.float64),
                u_kernel=uc.kernel.detach().cpu().numpy().astype(np.float64).reshape(uc.channels, uc.channels),
                u_bias=uc.bias.detach().cpu().numpy().astype(np.float64))


@pytest.mark.parametrize("shape", [(128, 64, 64, 128), (128, 8, 8, 128), (128, 4, 4, 128)])
def test_ufconv_200_classes_per_sample_tables(shape):
    """Tiny-ImageNet cWC_sa block site: FactorizedConv11(number_of_classes=200, filters_emb=15) + Conv2D 1x1 -> Add
    (generator.py:69-78) behind the fused stack; K = 200 > N = 128 -> one table per sample.  Values, dx and the
    gradients of all four weight tensors against the oracle."""
    from wc_gan_amd.generator import create_norm
    N, C = shape[0], shape[-1]
    K, E = 200, 15
    torch.manual_seed(3)
    stack = create_norm('d', 'ufconv', number_of_classes=K, filters_emb=E)(axis=-1, name='s', channels=C).cuda()
    for p in stack.parameters():
        torch.nn.init.normal_(p, std=0.2)
    rng = np.random.default_rng(14)
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    cls = rng.integers(0, K, (N, 1)).astype(np.int32)
    gy = rng.standard_normal(shape).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = stack(xt, dev(cls, torch.int32))
    y.backward(dev(gy))
    p = _ufconv_params(stack)
    Gk, Bk = o.coloring_table('ufconv', C, p, K)
    idx = cls.reshape(-1)
    y_ref, cache = o.wc_forward(x, Gk[idx], Bk[idx], np.arange(N))          # the reference gathers per sample too
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)                       # dG_ref: (N, C, C) per sample
    fc, uc = stack.branches
    d_fk = np.einsum('ne,nio->eio', p['f_alpha'][idx], dG_ref)
    d_alpha = np.zeros((K, E)); np.add.at(d_alpha, idx, np.einsum('nio,eio->ne', dG_ref, p['f_kernel']))
    errs = dict(y=rel(y.detach().cpu().numpy(), y_ref), dx=rel(xt.grad.cpu().numpy(), dx_ref),
                d_u_kernel=rel(uc.kernel.grad.cpu().numpy().reshape(C, C), dG_ref.sum(0)),
                d_u_bias=rel(uc.bias.grad.cpu().numpy(), dB_ref.sum(0)),
                d_f_kernel=rel(fc.kernel.grad.cpu().numpy(), d_fk),
                d_class_matrix=rel(fc.class_matrix.grad.cpu().numpy(), d_alpha))
    print(shape, errs)
    assert all(v < TOL for v in errs.values()), errs


def test_ufconv_per_sample_tables_grouped():
    """The same site in the critic phase: 5 x 64 samples in one call, 200 classes > 64 samples per group -> 320
    per-sample tables, each folded into ITS group's whitening matrix (wc_color_f32 per_group)."""
    from wc_gan_amd.generator import create_norm
    from wc_gan_amd.layers import statistic_groups
    shape, groups, K, E = (320, 16, 16, 128), 5, 200, 15
    N, C = shape[0], shape[-1]
    n = N // groups
    torch.manual_seed(4)
    stack = create_norm('d', 'ufconv', number_of_classes=K, filters_emb=E)(axis=-1, name='s', channels=C).cuda()
    for p in stack.parameters():
        torch.nn.init.normal_(p, std=0.2)
    rng = np.random.default_rng(15)
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    cls = rng.integers(0, K, (N, 1)).astype(np.int32)
    with torch.no_grad(), statistic_groups(groups):
        y = stack(dev(x), dev(cls, torch.int32))
    Gk, Bk = o.coloring_table('ufconv', C, _ufconv_params(stack), K)
    idx = cls.reshape(-1)
    for g in range(groups):
        sl = slice(g * n, (g + 1) * n)
        y_ref, _ = o.wc_forward(x[sl], Gk[idx[sl]], Bk[idx[sl]], np.arange(n))
        assert rel(y[sl].cpu().numpy(), y_ref) < TOL, g


@pytest.mark.parametrize("name", ["cifar10_uncond", "cifar10_cond", "stl10_uncond", "tinyimagenet_cond_sa"])
def test_training_step_of_every_config(name):
    """One eager G+D step of each BASELINE configuration at its full widths and image size (training_ratio 2 to keep
    the grouped critic-phase path in): finite losses, every parameter of both networks moves, moving statistics move."""
    from wc_gan_amd.train import CONFIGS, build_trainer, wc_sites
    cfg = CONFIGS[name]
    torch.manual_seed(5)
    tr = build_trainer(cfg, "cuda", training_ratio=2)
    H, W, Ci = cfg['image_shape']
    g = torch.Generator(device="cpu"); g.manual_seed(6)
    reals = [(torch.rand(64, H, W, Ci, generator=g) * 2 - 1).cuda() for _ in range(2)]
    K = cfg['generator']['number_of_classes']
    labels = [torch.randint(0, K, (64, 1), generator=g, dtype=torch.int32).cuda() for _ in range(2)] if cfg['conditional'] else None
    sites = wc_sites(cfg, 128)
    assert sites[-1][1:] == (128, H, W, cfg['generator']['block_sizes'][-1])
    before_g = [p.detach().clone() for p in tr.G.parameters()]
    before_d = [p.detach().clone() for p in tr.D.parameters()]
    mc0 = tr.G.final_norm.npart.moving_cov.clone()
    d_loss, g_loss = tr.step(reals, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
    # (the critic's output bias is exempt: while every hinge term is active its gradient is -1 + 1 = 0 exactly)
    assert all((a != b.detach()).any() for a, b in zip(before_g, tr.G.parameters()) if a.dim() > 1)
    assert all((a != b.detach()).any() for a, b in zip(before_d, tr.D.parameters()) if a.dim() > 1)
    assert not torch.equal(mc0, tr.G.final_norm.npart.moving_cov)
    with torch.no_grad():
        tr.G.eval()
        z, c = tr._noise(4)
        img = tr.G(z, c)
        tr.G.train()
    assert img.shape == (4, H, W, Ci) and torch.isfinite(img).all()
    if cfg['conditional']:
        with pytest.raises(ValueError):
            tr.step(reals)                       # a conditional recipe refuses to train the critic without real labels
