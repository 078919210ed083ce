"""SURVEY row a8 on the GPU: the soft-assignment coloring's dictionary mix (wc_factor_mix_f32 / wc_factor_mix_bwd_f32) against the float64
restatement oracle.coloring_table('ufconv' / 'fconv') (generator.py:69-78), forward and all three gradients, and the layer route that uses it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(K, E, C, N, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((E, C, C)) / np.sqrt(C)
    a = rng.standard_normal((K, E)) / np.sqrt(E)
    u = rng.standard_normal((C, C)) / np.sqrt(C)
    cls = rng.integers(0, K, size=N)
    g = rng.standard_normal((N, C, C))
    return d, a, u, cls, g


@pytest.mark.parametrize("K,E,C,N", [(200, 15, 128, 128), (10, 4, 256, 64), (1000, 32, 64, 48), (100, 10, 32, 7)])
def test_mix_forward_and_gradients_match_the_float64_restatement(K, E, C, N):
    from oracle import wc_oracle as O
    from wc_gan_amd import ops
    d, a, u, cls, g = _case(K, E, C, N, 7 + K)
    G64, _ = O.coloring_table('ufconv', C, {'f_kernel': d, 'f_alpha': a, 'u_kernel': u, 'u_bias': np.zeros(C)}, number_of_classes=K)
    dev = torch.device('cuda')
    td, ta, tu = (torch.tensor(v, dtype=torch.float32, device=dev) for v in (d, a, u))
    idx = torch.tensor(cls, dtype=torch.int32, device=dev)
    # one table per sample (K > N in the reference's Tiny-ImageNet / ImageNet recipes), and one per class
    out = ops.factor_mix(td, ta, idx, tu)
    assert np.abs(out.double().cpu().numpy() - G64[cls]).max() <= 2e-6 * np.abs(G64).max()
    out_all = ops.factor_mix(td, ta, None, tu)
    assert np.abs(out_all.double().cpu().numpy() - G64).max() <= 2e-6 * np.abs(G64).max()
    assert torch.equal(out_all[idx.long()], out)                  # same expression for a table whichever way it is addressed
    nobase = ops.factor_mix(td, ta, idx, None)
    G0, _ = O.coloring_table('fconv', C, {'f_kernel': d, 'f_alpha': a}, number_of_classes=K)
    assert np.abs(nobase.double().cpu().numpy() - G0[cls]).max() <= 2e-6 * np.abs(G0).max()
    # gradients: d dict[e] = sum_n alpha[cls n, e] g[n];  d alpha[k, e] = sum_{n: cls n = k} <g[n], dict[e]>;  d base = sum_n g[n]
    tg = torch.tensor(g, dtype=torch.float32, device=dev)
    dd, da, db = ops.factor_mix_bwd(td, ta, idx, tg, True, True, True)
    dd64 = np.einsum('ne,nio->eio', a[cls], g)
    da64 = np.zeros((K, E)); np.add.at(da64, cls, np.einsum('nio,eio->ne', g, d))
    db64 = g.sum(0)
    assert np.abs(dd.double().cpu().numpy() - dd64).max() <= 5e-6 * np.abs(dd64).max()
    assert np.abs(da.double().cpu().numpy() - da64).max() <= 5e-6 * np.abs(da64).max()
    assert np.abs(db.double().cpu().numpy() - db64).max() <= 5e-6 * np.abs(db64).max()
    absent = np.setdiff1d(np.arange(K), cls)
    assert absent.size == 0 or float(da[torch.tensor(absent, device=dev)].abs().max()) == 0.0
    dd2, da2, db2 = ops.factor_mix_bwd(td, ta, idx, tg, True, True, True)
    assert torch.equal(dd, dd2) and torch.equal(da, da2) and torch.equal(db, db2)          # fixed summation order


def test_autograd_function_matches_the_torch_expression_it_replaces():
    from wc_gan_amd import functional as WF
    K, E, C, N = 200, 15, 128, 64
    d, a, u, cls, g = _case(K, E, C, N, 3)
    dev = torch.device('cuda')
    mk = lambda v: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True)
    td, ta, tu = mk(d), mk(a), mk(u)
    idx = torch.tensor(cls, dtype=torch.int32, device=dev)
    tg = torch.tensor(g, dtype=torch.float32, device=dev)
    out = WF.factor_mix(td, ta, idx, tu)
    out.backward(tg)
    rd, ra, ru = (t.detach().clone().requires_grad_(True) for t in (td, ta, tu))
    ref = ((ra @ rd.view(E, C * C)).view(K, C, C) + ru.view(1, C, C))[idx.long()]       # rounds 1-3: matmul over all K classes, add, gather
    ref.backward(tg)
    assert float((out - ref).detach().abs().max()) <= 1e-5 * float(ref.detach().abs().max())
    for got, want in ((td.grad, rd.grad), (ta.grad, ra.grad), (tu.grad, ru.grad)):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_ufconv_site_builds_its_tables_with_the_mix_kernel(monkeypatch):
    """create_norm('d', 'ufconv') at a Tiny-ImageNet block site builds its tables through wc_factor_mix_f32 -- one per sample where K = 200 > N,
    one per class where the batch is larger (tests/test_configs_gpu.py::test_ufconv_* check values and all gradients of such sites against
    the oracle)."""
    from wc_gan_amd import ops
    from wc_gan_amd.generator import create_norm
    seen = []
    real = ops.factor_mix
    monkeypatch.setattr(ops, "factor_mix", lambda d, a, idx=None, base=None: (seen.append(None if idx is None else idx.numel()), real(d, a, idx, base))[1])
    torch.manual_seed(0)
    dev = torch.device('cuda')
    for N, K in ((32, 200), (64, 10)):
        C, H, E = 128, 16, 15
        stack = create_norm('d', 'ufconv', number_of_classes=K, filters_emb=E)(axis=-1, name='s', channels=C).cuda()
        x = torch.randn(N, H, H, C, device=dev, requires_grad=True)
        cls = torch.randint(0, K, (N, 1), dtype=torch.int32, device=dev)
        y = stack(x, cls)
        y.square().mean().backward()
        fc, uc = stack.branches
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in (fc.kernel, fc.class_matrix, uc.kernel, uc.bias))
    assert seen == [32, None], seen
