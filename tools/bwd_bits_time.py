"""The ReLU'd backward at the headline site with and without a masked copy of the gradient (wc_bwd_reduce_mask_f32 + wc_bwd_apply_scaled_f32
against wc_bwd_reduce_bits_f32 + wc_bwd_apply_bits_f32): stage times, and a stress loop -- K6 with the bits run many times against
the first result (bit-for-bit: a lost hand-counted wait shows up as a rare wrong row)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 32, 256)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 300
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(3)
x = (torch.randn(N, H, H, C, generator=g) + 0.3).cuda(); gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / 16).cuda(); b = torch.zeros(1, C).cuda()
mu, L, W, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
A, At, plan = ops.color(W, gamma, cs)
y, mask = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True)
R, gs, gm, sc = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask)
_, _, S, gmean = ops.bwd_factor(R, gs, W, L, gamma, A, M, 1e-3, 1, True)
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
print("K4 mask, writes the masked copy   %.1f us" % t(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask)))
print("K4 bits, no copy                  %.1f us" % t(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask, write_masked=False)))
print("K4 plain (no ReLU)                %.1f us" % t(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)))
print("K6 on the masked copy             %.1f us" % t(lambda: ops.bwd_apply(gm, x, mu, At, S, gmean, None, scales=sc)))
print("K6 with the bits                  %.1f us" % t(lambda: ops.bwd_apply(gy, x, mu, At, S, gmean, None, scales=sc, relu_mask=mask)))
ref = ops.bwd_apply(gy, x, mu, At, S, gmean, None, scales=sc, relu_mask=mask)
ref0 = ops.bwd_apply(gm, x, mu, At, S, gmean, None, scales=sc)
print("bits vs copy: max |diff| / max %.2e" % float((ref - ref0).abs().max() / ref0.abs().max()))
bad = 0
for i in range(iters):
    d = ops.bwd_apply(gy, x, mu, At, S, gmean, None, scales=sc, relu_mask=mask)
    if not torch.equal(d, ref):
        bad += 1
        rows = ((d != ref).view(M, C).any(1)).nonzero().flatten()
        print("iteration", i, "differs in", int(rows.numel()), "rows, first", rows[:8].tolist())
print("stress: %d of %d runs differ" % (bad, iters))
