"""Development check of wc_apply_planes_f32 (K3 -> convolution hand-off): the planes against K3's fp32 output split by the
convolution's own split (wc_conv_split_f32), the mask against the fp32 form's, and the timing of both routes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops, conv
torch.manual_seed(0)
def run(N, H, C, Kc, gscale=1.0, boost=None, time_it=False):
    M = N * H * H
    g = torch.Generator(device='cpu'); g.manual_seed(N + H + C + Kc)
    x = (torch.randn(N, H, H, C, generator=g) * 2 + 0.5).cuda()
    if boost: x[3, 1, 1, 7] = boost
    gamma = (torch.randn(Kc, C, C, generator=g) / C ** 0.5 * gscale).cuda(); beta = (torch.randn(Kc, C, generator=g) * 0.1).cuda()
    slot = torch.randint(0, Kc, (N,), generator=g).int().cuda() if Kc > 1 else None
    s, xtx = ops.stats(x.view(M, C))
    mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
    A, At, plan = ops.color(W, gamma, cs)
    y, mask = ops.apply(x, mu, A, beta, slot, plan=plan, relu=True, want_mask=True)
    rec = ops.out_scale(gamma, beta, C, x.device)
    planes, rec, pmask = ops.apply_planes(x, mu, A, beta, slot, plan, rec, relu=True, want_mask=True)
    torch.cuda.synchronize()
    s_used, s_pred = float(rec[0]), float(rec[1])
    back = (planes[0].double() + planes[1].double()) / s_used
    err = float((back - y.double()).abs().max()); ymax = float(y.abs().max())
    hi, lo, xs = conv.split_planes(y)
    print("N %d H %d C %d Kc %d: scale used %g predicted %g (conv's own %g)  max|y| %.3f  max|planes - y| %.3g (%.2g of max)  mask equal %s  amax partial max*1/s %.3f"
          % (N, H, C, Kc, s_used, s_pred, float(xs[0]), ymax, err, err / ymax, bool((mask == pmask).all()), float(rec[2:2 + 256].max()) / s_used if s_used == s_pred else -1))
    assert err <= ymax * 2.0 ** -20, err
    assert bool((mask == pmask).all())
    if time_it:
        def t(f, n=50):
            for _ in range(5): f()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
        a = t(lambda: ops.apply(x, mu, A, beta, slot, plan=plan, relu=True, want_mask=True, out=y))
        b = t(lambda: conv.split_planes(y))
        c = t(lambda: ops.apply_planes(x, mu, A, beta, slot, plan, rec, relu=True, want_mask=True))
        print("   K3 fp32+mask %.1f us, conv absmax+split %.1f us | K3 planes+mask (two launches) %.1f us" % (a, b, c))
run(128, 32, 256, 1, time_it=True)
run(128, 16, 256, 1, time_it=True)
run(128, 8, 256, 1, time_it=True)
run(128, 32, 256, 10)
run(64, 16, 128, 1)
run(128, 16, 256, 1, gscale=40.0)                # large outputs: scale < 1
run(128, 16, 256, 1, boost=3.0e4)                # an outlier beyond the prediction: the gated second pass
