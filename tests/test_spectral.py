"""N3: the fused spectral-norm op.  CPU: the oracle against torch's own parametrisation and against torch autograd
(independent implementations).  GPU: the HIP op, through the C ABI, against the oracle."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O


def _rand(shape, seed):
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def test_oracle_power_iteration_properties():
    W = _rand((20, 48), 11).numpy()
    u0 = np.ones(20) / np.sqrt(20.0); v0 = np.ones(48) / np.sqrt(48.0)
    top = np.linalg.svd(W, compute_uv=False)[0]
    w1, s1, u1, v1 = O.spectral_normalize(W, u0, v0, iterations=1)
    # one step by hand, in the order of SN-GAN's Algorithm 1 (v from the persistent u, then u)
    v_ref = W.T @ u0; v_ref /= np.linalg.norm(v_ref); u_ref = W @ v_ref; s_ref = np.linalg.norm(u_ref); u_ref /= s_ref
    assert np.abs(u1 - u_ref).max() < 1e-14 and np.abs(v1 - v_ref).max() < 1e-14 and abs(s1 - s_ref) < 1e-12
    assert np.abs(w1 - W / s_ref).max() < 1e-14
    w, s, u, v = O.spectral_normalize(W, u0, v0, iterations=200)
    assert abs(s - top) / top < 1e-9 and abs(np.linalg.svd(w, compute_uv=False)[0] - 1.0) < 1e-9
    # torch's parametrisation updates u first; started from its own (u, v) our order reproduces it one half-step later
    lin = torch.nn.Linear(48, 20, bias=False).double()
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(W))
    sn = torch.nn.utils.parametrizations.spectral_norm(lin, n_power_iterations=1)
    p = sn.parametrizations.weight[0]
    ut = p._u.clone().numpy(); vt = p._v.clone().numpy()
    sn.train(); w_t = sn.weight.detach().numpy()            # u <- N(W v), v <- N(W^T u), sigma = u^T W v
    u_a = W @ vt; u_a /= np.linalg.norm(u_a)
    w_o, s_o, _, _ = O.spectral_normalize(W, u_a, vt, iterations=0)
    v_a = W.T @ u_a; v_a /= np.linalg.norm(v_a)
    assert np.abs(w_t - W / float(u_a @ (W @ v_a))).max() < 1e-12 and s_o > 0 and ut.shape == (20,)
    w0, s0, u_same, v_same = O.spectral_normalize(W, ut, vt, iterations=0)     # inference: u, v untouched
    assert np.array_equal(u_same, ut) and np.array_equal(v_same, vt) and abs(s0 - ut @ (W @ vt)) < 1e-14


@pytest.mark.parametrize("fully_diff", [True, False])
def test_oracle_backward_matches_autograd(fully_diff):
    W = _rand((12, 30), 1).requires_grad_(True); g = _rand((12, 30), 2)
    u = torch.nn.functional.normalize(_rand((12,), 3), dim=0); v = torch.nn.functional.normalize(_rand((30,), 4), dim=0)
    sigma = u @ (W @ v)
    w_sn = W / (sigma if fully_diff else sigma.detach())
    (w_sn * g).sum().backward()
    d = O.spectral_normalize_backward(g.numpy(), w_sn.detach().numpy(), u.numpy(), v.numpy(), float(sigma), fully_diff)
    assert np.abs(W.grad.numpy() - d).max() < 1e-12


SHAPES = [(128, 3, 3, 128), (128, 3, 3, 3), (128, 1, 1, 128), (1, 128), (10, 128), (256, 3, 3, 256), (64, 100),
          (1024, 3, 3, 512)]        # the last one: the 128-workgroup form (a Tiny-ImageNet critic weight)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("iterations", [0, 1, 3])
def test_hip_forward_matches_oracle(shape, iterations):
    from wc_gan_amd import ops
    if len(shape) == 4:      # (Cout, kh, kw, Cin) memory order of a channels_last kernel
        co, kh, kw, ci = shape
        w = _rand((co, ci, kh, kw), 5).float().cuda().contiguous(memory_format=torch.channels_last)
        Wm = w.permute(0, 2, 3, 1).reshape(co, -1).double().cpu().numpy()
    else:
        w = _rand(shape, 5).float().cuda(); Wm = w.double().cpu().numpy()
    R, K = Wm.shape
    u = torch.nn.functional.normalize(_rand((R,), 6), dim=0).float().cuda()
    v = torch.nn.functional.normalize(_rand((K,), 7), dim=0).float().cuda()
    u0, v0 = u.double().cpu().numpy(), v.double().cpu().numpy()
    ws = ops.spectral_norm_workspace(R, K, 'cuda')
    for rep in range(2):      # the second launch finds the meeting words re-armed
        if rep: u.copy_(torch.from_numpy(u0).float()); v.copy_(torch.from_numpy(v0).float())
        w_sn, sigma = ops.spectral_norm(w, u, v, iterations, ws)
    w_o, s_o, u_o, v_o = O.spectral_normalize(Wm, u0, v0, iterations)
    got = (w_sn.permute(0, 2, 3, 1).reshape(R, K) if len(shape) == 4 else w_sn).double().cpu().numpy()
    assert w_sn.stride() == w.stride()
    assert abs(float(sigma) - s_o) / abs(s_o) < 2e-5           # fp32 sums of up to 4608 terms vs float64
    assert np.abs(got - w_o).max() / np.abs(w_o).max() < 2e-5
    assert np.abs(u.double().cpu().numpy() - u_o).max() < 2e-5 and np.abs(v.double().cpu().numpy() - v_o).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 3, 3, 128), (10, 128), (1, 128), (512, 3, 3, 512)])
@pytest.mark.parametrize("fully_diff", [True, False])
def test_hip_backward_matches_oracle(shape, fully_diff):
    from wc_gan_amd.spectral import SpectralNormFunction
    if len(shape) == 4:
        co, kh, kw, ci = shape
        w = _rand((co, ci, kh, kw), 8).float().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        g = _rand((co, ci, kh, kw), 9).float().cuda().contiguous(memory_format=torch.channels_last)
        mat = lambda t: t.permute(0, 2, 3, 1).reshape(co, -1).double().cpu().numpy()
    else:
        w = _rand(shape, 8).float().cuda().requires_grad_(True); g = _rand(shape, 9).float().cuda()
        mat = lambda t: t.double().cpu().numpy()
    R, K = mat(w.detach()).shape
    u = torch.nn.functional.normalize(_rand((R,), 6), dim=0).float().cuda()
    v = torch.nn.functional.normalize(_rand((K,), 7), dim=0).float().cuda()
    from wc_gan_amd import ops
    ws = ops.spectral_norm_workspace(R, K, 'cuda')
    w_sn, sigma = SpectralNormFunction.apply(w, u, v, ws, 1, 1e-12, fully_diff)
    (w_sn * g).sum().backward()
    d = O.spectral_normalize_backward(mat(g), mat(w_sn.detach()), u.double().cpu().numpy(), v.double().cpu().numpy(), float(sigma), fully_diff)
    assert w.grad.stride() == w.stride()
    assert np.abs(mat(w.grad) - d).max() / np.abs(d).max() < 2e-5


@pytest.mark.gpu
def test_sn_layers_train_eval_and_state():
    from wc_gan_amd.spectral import SNConv2d, SNEmbedding, SNLinear
    conv = SNConv2d(16, 32, 3, padding=1).cuda()
    x = torch.randn(4, 16, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    u0 = conv.sn_u.clone()
    conv.train(); y = conv(x); y.sum().backward()
    assert conv.weight.grad is not None and not torch.equal(conv.sn_u, u0)          # one power iteration ran
    conv.eval(); u1 = conv.sn_u.clone(); y1 = conv(x); y2 = conv(x)
    assert torch.equal(conv.sn_u, u1) and torch.equal(y1, y2)                        # inference leaves u, v alone
    wm = conv.weight.detach().permute(0, 2, 3, 1).reshape(32, -1)
    top = torch.linalg.svdvals(wm.double())[0]
    sig = (conv.sn_u.double() @ (wm.double() @ conv.sn_v.double()))
    assert abs(float(sig / top) - 1) < 0.05                                           # 15 warm-up iterations at construction
    lin = SNLinear(128, 1).cuda(); emb = SNEmbedding(10, 128).cuda()
    h = torch.randn(8, 128, device='cuda')
    out = lin(h) + (emb(torch.arange(8, device='cuda') % 10) * h).sum(1, keepdim=True)
    out.sum().backward()
    assert lin.weight.grad.shape == (1, 128) and emb.weight.grad.shape == (10, 128)
    with pytest.raises(RuntimeError):
        SNLinear(4, 4)(torch.randn(2, 4))                                             # no non-HIP path


import os
SN_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "sn_golden.npz")
SN_NAMES = ["conv3x3", "conv1x1", "dense_row", "embedding", "eval"]


@pytest.mark.parametrize("name", SN_NAMES)
def test_oracle_reproduces_sn_golden(name):
    z = np.load(SN_GOLDEN)
    f = lambda k: z[f"{name}/{k}"]
    w_sn, sigma, u1, v1 = O.spectral_normalize(f("W"), f("u0"), f("v0"), int(f("iterations")))
    assert np.abs(w_sn - f("w_sn")).max() < 1e-6 and abs(sigma - float(f("sigma"))) < 1e-6 * abs(sigma)
    assert np.abs(u1 - f("u1")).max() < 1e-6 and np.abs(v1 - f("v1")).max() < 1e-6
    for key, fd in (("dW_full", True), ("dW_const", False)):
        d = O.spectral_normalize_backward(f("g"), w_sn, u1, v1, sigma, fd)
        assert np.abs(d - f(key)).max() < 1e-5 * np.abs(f(key)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SN_NAMES)
def test_hip_matches_sn_golden(name):
    from wc_gan_amd import ops
    z = np.load(SN_GOLDEN)
    f = lambda k: torch.from_numpy(np.ascontiguousarray(z[f"{name}/{k}"])).cuda()
    W, u, v, g = f("W"), f("u0").clone(), f("v0").clone(), f("g")
    ws = ops.spectral_norm_workspace(W.shape[0], W.shape[1], 'cuda')
    w_sn, sigma = ops.spectral_norm(W, u, v, int(z[f"{name}/iterations"]), ws)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(w_sn, f("w_sn")) < 2e-5 and rel(sigma, f("sigma").reshape(1)) < 2e-5
    assert float((u - f("u1")).abs().max()) < 2e-5 and float((v - f("v1")).abs().max()) < 2e-5
    for key, fd in (("dW_full", True), ("dW_const", False)):
        assert rel(ops.spectral_norm_bwd(g, w_sn, u, v, sigma, fd, ws), f(key)) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("fully_diff", [False, True])
def test_batched_layers_match_layer_by_layer(fully_diff):
    """prepare_spectral (one launch for every layer) gives bit-identical weights, gradients and u/v to per-layer calls."""
    import copy
    import torch.nn as nn
    from wc_gan_amd.spectral import SNConv2d, SNLinear, prepare_spectral
    torch.manual_seed(9)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            kw = dict(fully_diff_spectral=fully_diff)
            self.a = SNConv2d(8, 16, 3, padding=1, **kw); self.b = SNConv2d(16, 16, 1, **kw); self.c = SNLinear(16, 1, **kw)

        def forward(self, x, batched):
            if batched:
                prepare_spectral(self)
            h = torch.relu(self.a(x)); h = torch.relu(self.b(h))
            return self.c(h.mean(dim=(2, 3)))

    n1 = Net().cuda(); n2 = copy.deepcopy(n1)
    x = torch.randn(4, 8, 10, 10, device='cuda').contiguous(memory_format=torch.channels_last)
    for step in range(2):                          # second round: meeting words re-armed, u/v moved on identically
        y1 = n1(x, True); y2 = n2(x, False)
        assert torch.equal(y1, y2)
        y1.sum().backward(); y2.sum().backward()
        for p1, p2 in zip(n1.parameters(), n2.parameters()):      # (MIOpen's weight-gradient kernels are not bit-reproducible)
            assert float((p1.grad - p2.grad).abs().max()) <= 1e-5 * float(p2.grad.abs().max())
        for b1, b2 in zip(n1.buffers(), n2.buffers()):
            assert torch.equal(b1, b2)
        n1.zero_grad(); n2.zero_grad()
    # a weight that changed after prepare_spectral is re-normalised by its layer, not served stale
    n1.eval(); n2.eval()                           # inference: u, v stay put, so the extra prepare changes no state
    prepare_spectral(n1)
    with torch.no_grad():
        n1.a.weight.mul_(2.0); n2.a.weight.mul_(2.0)
    assert torch.equal(n1(x, False), n2(x, False))


@pytest.mark.gpu
def test_forward_leaves_the_maximum_of_the_normalised_weight():
    """the floats behind wc_spectral_norm_amax_offset: their maximum is max|w_sn| (the convolution's weight split reads
    them instead of sweeping the weight again)"""
    from wc_gan_amd.spectral import SNConv2d
    torch.manual_seed(4)
    for cin, cout in ((128, 128), (3, 128), (256, 64)):
        m = SNConv2d(cin, cout, 3, padding=1)
        m._sn_init(1, False, True)
        m = m.cuda()
        m.train()
        w = m.normalized_weight()
        assert float(w._wc_amax.max()) == float(w.detach().abs().max())
        m.eval()
        w = m.normalized_weight()
        assert float(w._wc_amax.max()) == float(w.detach().abs().max())


@pytest.mark.gpu
def test_one_power_iteration_per_forward_on_every_layer_path():
    """a spectrally normalised layer advances u, v ONCE per training forward, also when the fast convolution kernel does
    not take the shape and the layer falls back to MIOpen (the critic's 3 -> 128 first layer)"""
    from wc_gan_amd.generator import Conv2D
    torch.manual_seed(3)
    for cin, cout, k in ((3, 128, 3), (3, 128, 1), (128, 128, 3)):
        layer = Conv2D(cin, cout, (k, k), spectral=True).cuda().train()
        ref = Conv2D(cin, cout, (k, k), spectral=True).cuda().train()
        ref.load_state_dict(layer.state_dict())
        x = torch.randn(8, 8, 8, cin, device='cuda')
        layer(x)                                       # whatever path the shape takes
        ref.conv.normalized_weight()                   # exactly one iteration
        torch.cuda.synchronize()
        assert torch.equal(layer.conv.sn_u, ref.conv.sn_u) and torch.equal(layer.conv.sn_v, ref.conv.sn_v)
        layer.forward_relu(x)
        ref.conv.normalized_weight()
        assert torch.equal(layer.conv.sn_u, ref.conv.sn_u)


@pytest.mark.gpu
def test_error_word_stays_clear_and_batches_split_by_residency():
    """the meeting of a weight's workgroups is a bounded wait that sets a sticky error word instead of hanging; a
    normal run (per layer, and the whole Tiny-ImageNet-sized critic in batched launches that are cut to what the device
    holds at once) leaves every word 0 and gives the per-layer results."""
    from wc_gan_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(7)
    shapes = [(1024, 9216)] * 6 + [(512, 4608), (128, 27), (256, 1152)] * 4       # > 768 workgroups of up to ~42 KB LDS in one call
    ws_, us, vs, wss = [], [], [], []
    for r, c in shapes:
        ws_.append(torch.randn(r, c, device='cuda') / c ** 0.5)
        us.append(torch.randn(r, device='cuda')); vs.append(torch.randn(c, device='cuda'))
        wss.append(ops.spectral_norm_workspace(r, c, 'cuda'))
    us2, vs2 = [u.clone() for u in us], [v.clone() for v in vs]
    wss2 = [ops.spectral_norm_workspace(r, c, 'cuda') for r, c in shapes]
    outs = ops.spectral_norm_batched(ws_, us, vs, wss, 1)
    for i, (r, c) in enumerate(shapes):
        w1, s1 = ops.spectral_norm(ws_[i], us2[i], vs2[i], 1, wss2[i])
        assert torch.equal(outs[i][0], w1) and torch.equal(outs[i][1], s1)
        off = lib.wc_spectral_norm_error_offset(r, c)
        assert off + 32 == wss[i].numel()
        for buf in (wss[i], wss2[i]):
            assert int(buf[off:off + 4].view(torch.int32)[0]) == 0
            assert int(buf[-16:].view(torch.int32).abs().sum()) == 0          # the meeting words are left zero
