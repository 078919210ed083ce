"""The custom-operator surface torch.ops.wc.* (SURVEY.md section 8b): registration and fake kernels on the CPU,
opcheck and numerics against the float64 oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as o


def test_ops_are_registered_with_schemas_and_fake_kernels():
    import wc_gan_amd.torch_ops  # noqa: F401
    names = ["stats", "factor", "color", "apply", "bwd_reduce", "bwd_factor", "bwd_apply", "whiten_color"]
    for n in names:
        assert hasattr(torch.ops.wc, n), n
    assert "!)? moving_cov" in str(torch.ops.wc.factor.default._schema)          # the in-place update of the moving statistics is declared
    # shape propagation without a GPU: meta tensors go to the fake kernels
    x = torch.empty(4096, 64, device="meta")
    s, xtx = torch.ops.wc.stats(x, 1)
    assert s.shape == (64,) and xtx.shape == (64, 64) and s.dtype == torch.float64
    s5, xtx5 = torch.ops.wc.stats(x, 4)
    assert s5.shape == (4, 64) and xtx5.shape == (4, 64, 64)
    mu, L, W, cs = torch.ops.wc.factor(s, xtx, 4096, 64, 1e-3, 0.99, 1, True, None, None, 1)
    assert mu.shape == (64,) and L.shape == W.shape == (64, 64) and cs.shape == (64,)
    A, At, plan = torch.ops.wc.color(W, torch.empty(3, 64, 64, device="meta"), cs, 1, False)
    assert A.shape == At.shape == (3, 64, 64) and plan.dtype == torch.uint8 and plan.numel() > 1
    x4 = torch.empty(8, 4, 4, 64, device="meta")
    y = torch.ops.wc.apply(x4, mu, A, None, torch.empty(8, dtype=torch.int32, device="meta"), plan, True)
    assert y.shape == x4.shape
    R, gs = torch.ops.wc.bwd_reduce(x4, mu, x4, None, 1)
    assert R.shape == (1, 64, 64) and gs.shape == (1, 64)
    out = torch.ops.wc.whiten_color(x4, None, None, None, None, None, True, 1e-3, 0.99, 1, False)
    assert out[0].shape == x4.shape and out[3].shape == (64, 64)


def test_ops_refuse_cpu_tensors():
    import wc_gan_amd.torch_ops  # noqa: F401
    from wc_gan_amd import _lib
    with pytest.raises(_lib.WcHipError):
        torch.ops.wc.stats(torch.zeros(64, 32), 1)


@pytest.mark.gpu
def test_opcheck_and_numerics():
    import wc_gan_amd.torch_ops as T
    rng = np.random.default_rng(21)
    shape, C, Kc = (8, 8, 8, 64), 64, 3
    x = torch.tensor(o.synth_activation(rng, shape, "ill"), dtype=torch.float32, device="cuda")
    G, B = o.synth_coloring(rng, C, Kc)
    Gt = torch.tensor(G, dtype=torch.float32, device="cuda"); Bt = torch.tensor(B, dtype=torch.float32, device="cuda")
    slot = torch.tensor(rng.integers(0, Kc, shape[0]), dtype=torch.int32, device="cuda")
    # stage ops: schema / fake-kernel / aliasing checks by torch's own checker
    torch.library.opcheck(torch.ops.wc.stats.default, (x.view(-1, C), 1), test_utils=("test_schema", "test_faketensor"))
    s, xtx = torch.ops.wc.stats(x.view(-1, C), 1)
    mm = torch.zeros(C, 1, device="cuda"); mc = torch.eye(C, device="cuda")
    mu, L, W, cs = torch.ops.wc.factor(s, xtx, x.numel() // C, C, 1e-3, 0.99, 1, True, mm, mc, 1)
    assert float(mc.diagonal().sub(1).abs().max()) > 0           # the declared mutation happened
    A, At, plan = torch.ops.wc.color(W, Gt, cs, 1, False)
    torch.library.opcheck(torch.ops.wc.apply.default, (x, mu, A, Bt, slot, plan, False), test_utils=("test_schema", "test_faketensor"))
    y = torch.ops.wc.apply(x, mu, A, Bt, slot, plan, False)
    y_ref, cache = o.wc_forward(x.cpu().numpy(), G.astype(np.float32), B.astype(np.float32), slot.cpu().numpy())
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
    assert rel(y.cpu().numpy(), y_ref) < 1e-4
    # the fused operator with its registered autograd
    xt = x.clone().requires_grad_(True); Gp = Gt.clone().requires_grad_(True); Bp = Bt.clone().requires_grad_(True)
    gy = torch.tensor(rng.standard_normal(shape), dtype=torch.float32, device="cuda")
    mm2 = torch.zeros(C, 1, device="cuda"); mc2 = torch.eye(C, device="cuda")
    yy = T.whiten_color_site(xt, Gp, Bp, slot, mm2, mc2)
    yy.backward(gy)
    assert rel(mc2.cpu().numpy(), o.update_moving(np.zeros(C), np.eye(C), cache['mu'], cache['sigma'])[1]) < 1e-5
    out = (yy,)
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy.cpu().numpy(), cache)
    assert rel(out[0].detach().cpu().numpy(), y_ref) < 1e-4
    assert rel(xt.grad.cpu().numpy(), dx_ref) < 1e-4 and rel(Gp.grad.cpu().numpy(), dG_ref) < 1e-4 and rel(Bp.grad.cpu().numpy(), dB_ref) < 1e-4
    torch.library.opcheck(torch.ops.wc.whiten_color.default, (x, Gt, Bt, slot, None, None, True, 1e-3, 0.99, 1, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
