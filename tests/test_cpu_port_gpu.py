"""The HIP library against its CPU twin (oracle/wc_cpu.cpp, the `_cpu` C ABI) stage by stage on the same inputs -- the
conformance SURVEY.md section 8b describes: same symbols, same arguments, host pointers there, device pointers here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


@pytest.mark.parametrize("shape,Kc", [((16, 16, 16, 128), 1), ((24, 8, 8, 64), 4), ((32, 32, 32, 256), 1),
                                      ((128, 32, 32, 256), 1), ((128, 16, 16, 128), 10)])      # + the headline site and a conditional one at full size
def test_hip_stages_match_their_cpu_twins(shape, Kc):
    from oracle import cpu_port as cp
    from oracle import wc_oracle as o
    from wc_gan_amd import ops
    rng = np.random.default_rng(17)
    N, C = shape[0], shape[-1]
    M = int(np.prod(shape[:-1]))
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32) if Kc > 1 else None
    gy = rng.standard_normal(shape).astype(np.float32)
    st = dev(slot, torch.int32) if slot is not None else None
    # K1
    s_c, xtx_c = cp.stats(x.reshape(M, C))
    s_g, xtx_g = ops.stats(dev(x).view(M, C))
    _, cov_c = o.moments_to_stats(s_c, xtx_c, M)
    _, cov_g = o.moments_to_stats(s_g.cpu().numpy(), xtx_g.cpu().numpy(), M)
    assert rel(s_g.cpu().numpy(), s_c) < 1e-5 and rel(cov_g, cov_c) < 1e-7
    # K2 on the SAME moments
    mm_c = np.zeros(C, np.float32); mc_c = np.eye(C, dtype=np.float32)
    mu_c, L_c, W_c, cs_c = cp.factor(s_c, xtx_c, M, C, moving_mean=mm_c, moving_cov=mc_c)
    mm_g = torch.zeros(C, device="cuda"); mc_g = torch.eye(C, device="cuda")
    mu_g, L_g, W_g, cs_g = ops.factor(dev(s_c, torch.float64), dev(xtx_c, torch.float64), M, C, 1e-3, 0.99, 1, True, mm_g, mc_g,
                                      "cuda", want_scale=True)
    assert rel(mu_g.cpu().numpy(), mu_c) < 1e-6 and rel(L_g.cpu().numpy(), L_c) < 1e-9 and rel(W_g.cpu().numpy(), W_c) < 1e-8
    assert np.array_equal(cs_g.cpu().numpy(), cs_c)
    assert rel(mm_g.cpu().numpy(), mm_c) < 1e-6 and rel(mc_g.cpu().numpy(), mc_c) < 1e-6
    # color + K3 on the SAME W
    A_c, At_c = cp.color(W_c, G)
    A_g, At_g, plan = ops.color(dev(W_c, torch.float64), dev(G), cs_g)
    assert rel(A_g.cpu().numpy(), A_c) < 1e-6 and rel(At_g.cpu().numpy(), At_c) < 1e-6
    y_c = cp.apply(x, mu_c, A_c, B, slot, relu=True)
    y_g = ops.apply(dev(x), dev(mu_c), dev(A_c), dev(B), st, plan=None, relu=True)
    # (cond ~ 1e6: terms of size ~30 cancel to results of size ~1, so fp32-GEMM accuracy reads 1e-5 here; the contract is 1e-4)
    assert rel(y_g.cpu().numpy(), y_c) < 3e-5
    # K4 / K5 / K6 on the SAME inputs
    R_c, gs_c = cp.bwd_reduce(x, mu_c, gy, slot, Kc)
    R_g, gs_g = ops.bwd_reduce(dev(x), dev(mu_c), dev(gy), st, Kc)
    f2 = ((x.reshape(M, C).astype(np.float64) - mu_c) ** 2).sum(0); g2 = (gy.reshape(M, C).astype(np.float64) ** 2).sum(0)
    assert np.abs((R_g.cpu().numpy() - R_c) / np.sqrt(np.outer(f2, g2))[None]).max() < 1e-6
    assert rel(gs_g.cpu().numpy(), gs_c) < 1e-5
    dg_c, db_c, S_c, gm_c = cp.bwd_factor(R_c, gs_c, W_c, L_c, G, A_c, M)
    dg_g, db_g, S_g, gm_g = ops.bwd_factor(dev(R_c, torch.float64), dev(gs_c, torch.float64), dev(W_c, torch.float64),
                                           dev(L_c, torch.float64), dev(G), dev(A_c), M, 1e-3, 1, True)
    assert rel(dg_g.cpu().numpy(), dg_c) < 1e-6 and rel(db_g.cpu().numpy(), db_c) < 1e-6
    assert rel(S_g.cpu().numpy(), S_c) < 1e-5 and rel(gm_g.cpu().numpy(), gm_c) < 1e-5
    dx_c = cp.bwd_apply(gy, x, mu_c, At_c, S_c, gm_c, slot)
    dx_g = ops.bwd_apply(dev(gy), dev(x), dev(mu_c), dev(At_c), dev(S_c), dev(gm_c), st)
    assert rel(dx_g.cpu().numpy(), dx_c) < 3e-5


def test_grouped_hip_stages_match_their_cpu_twins():
    """The critic-phase form (five statistic groups in one call) of K1 / K2 / color / group bias / K3 against the `_cpu` twins."""
    from oracle import cpu_port as cp
    from oracle import wc_oracle as o
    from wc_gan_amd import ops
    rng = np.random.default_rng(19)
    shape, groups, Kc = (320, 16, 16, 256), 5, 1
    N, C = shape[0], shape[-1]
    M = int(np.prod(shape[:-1])); Mg = M // groups
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    s_c, xtx_c = cp.stats(x.reshape(M, C), groups)
    s_g, xtx_g = ops.stats(dev(x).view(M, C), groups)
    for g in range(groups):
        _, cov_c = o.moments_to_stats(s_c[g], xtx_c[g], Mg)
        _, cov_g = o.moments_to_stats(s_g[g].cpu().numpy(), xtx_g[g].cpu().numpy(), Mg)
        assert rel(cov_g, cov_c) < 1e-7
    mu_g, L_g, W_g, cs_g = ops.factor(dev(s_c, torch.float64), dev(xtx_c, torch.float64), Mg, C, 1e-3, 0.99, 1, True, None, None,
                                      "cuda", want_scale=True, groups=groups)
    A_g, At_g, plan = ops.color(W_g, dev(G), cs_g, groups)
    center, bias = ops.group_bias(mu_g.view(groups, C), A_g, dev(B), groups, Kc)
    slot = (torch.arange(N, device="cuda", dtype=torch.int32) // (N // groups)).contiguous()
    y_g = ops.apply(dev(x), center, A_g, bias, slot, plan=plan)
    for g in range(groups):          # each group against the single-group CPU pipeline on its own rows
        xg = x[g * (N // groups):(g + 1) * (N // groups)]
        mu_c, L_c, W_c, _ = cp.factor(s_c[g], xtx_c[g], Mg, C)
        A_c, _ = cp.color(W_c, G)
        y_c = cp.apply(xg, mu_c, A_c, B, None)
        assert rel(y_g[g * (N // groups):(g + 1) * (N // groups)].cpu().numpy(), y_c) < 5e-5
