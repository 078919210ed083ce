"""K2 on the GPU with the inverse inside the factor launch (wc_small.hip: tri_inverse_role): the
hand-off between the factorising and the inverting workgroups is a protocol (row-block counter, write-through stores) --
what is tested here is that it holds under repetition, with several matrices per launch, on two streams at once and
inside a replayed graph, against float64 torch on the same device (tolerance stated per check)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from wc_gan_amd import ops as _ops
    return _ops


def _moments(C, G, M, seed):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    mix = torch.randn(C, C, generator=g, dtype=torch.float64) / C ** 0.5 + \
        0.5 * (torch.randn(C, 4, generator=g, dtype=torch.float64) @ torch.randn(4, C, generator=g, dtype=torch.float64))
    x = torch.randn(G, M, C, generator=g, dtype=torch.float64) @ mix + 0.3
    s = x.sum(1); xtx = x.transpose(1, 2) @ x
    if G == 1:
        s, xtx = s[0], xtx[0]
    return s.cuda(), xtx.cuda()


def _reference(s, xtx, M, eps):
    s = s.reshape(-1, s.shape[-1]); xtx = xtx.reshape(-1, *xtx.shape[-2:])
    mu = s / M
    sigma = (xtx - M * mu[:, :, None] * mu[:, None, :]) / (M - 1)
    C = s.shape[-1]
    T = (1 - eps) * sigma + eps * torch.eye(C, dtype=torch.float64, device=s.device)
    L = torch.linalg.cholesky(T)
    W = torch.linalg.solve_triangular(L, torch.eye(C, dtype=torch.float64, device=s.device).expand_as(L), upper=False)
    return L, W


def _check(out, ref, G, C):
    L = out[1].view(G, C, C); W = out[2].view(G, C, C)
    Lr, Wr = ref
    # float64 Cholesky + forward substitution on a cond ~ 1e4..1e6 matrix: 1e-9 relative to the largest entry is > 1000 x the
    # rounding level of either implementation and far below anything the float32 tables built from W can see
    assert float((L - Lr).abs().max() / Lr.abs().max()) < 1e-9
    assert float((W - Wr).abs().max() / Wr.abs().max()) < 1e-9
    assert float(torch.triu(L, 1).abs().max()) == 0.0 and float(torch.triu(W, 1).abs().max()) == 0.0


@pytest.mark.parametrize("C,G", [(256, 1), (256, 5), (256, 8), (224, 3), (128, 1), (128, 7), (160, 2), (256, 9), (64, 2)])
def test_factor_and_inverse_repeated(ops, C, G):
    """one launch for 128 <= C <= 256 and up to 8 groups, two launches otherwise: both against float64 torch, 25 times over"""
    M = 2048
    s, xtx = _moments(C, G, M, 100 + C + G)
    ref = _reference(s, xtx, M, 1e-3)
    for _ in range(25):
        out = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, s.device, groups=G)
        _check(out, ref, G, C)


def test_factor_on_two_streams_at_once(ops):
    """two factor launches in flight together (the trainer runs the generator's forward beside the critic updates): each
    launch's inverting workgroups follow their own factorisation"""
    C, M = 256, 2048
    a = _moments(C, 1, M, 7); b = _moments(C, 5, M, 8)
    ra = _reference(*a, M, 1e-3); rb = _reference(*b, M, 1e-3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs_a, outs_b = [], []
    for _ in range(20):
        with torch.cuda.stream(s1):
            outs_a.append(ops.factor(a[0], a[1], M, C, 1e-3, 0.99, 1, True, None, None, a[0].device))
        with torch.cuda.stream(s2):
            outs_b.append(ops.factor(b[0], b[1], M, C, 1e-3, 0.99, 1, True, None, None, b[0].device, groups=5))
    torch.cuda.synchronize()
    for o in outs_a:
        _check(o, ra, 1, C)
    for o in outs_b:
        _check(o, rb, 5, C)


def test_factor_inside_a_replayed_graph(ops):
    """the row-block counters are re-armed by the launch in front (factor_prepare_kernel), so a captured K2 replays"""
    C, M, G = 256, 2048, 3
    s, xtx = _moments(C, G, M, 21)
    s_in, xtx_in = s.clone(), xtx.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.factor(s_in, xtx_in, M, C, 1e-3, 0.99, 1, True, None, None, s.device, groups=G)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.factor(s_in, xtx_in, M, C, 1e-3, 0.99, 1, True, None, None, s.device, groups=G)
    for seed in (31, 32, 33, 34):
        s2, xtx2 = _moments(C, G, M, seed)
        s_in.copy_(s2); xtx_in.copy_(xtx2)
        graph.replay()
        torch.cuda.synchronize()
        _check(out, _reference(s2, xtx2, M, 1e-3), G, C)
