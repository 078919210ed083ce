"""K2 (wc_factor_f64: prepare + Cholesky + inverse) time per call, HIP events, several widths and group counts."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for C, G in ((256, 1), (256, 5), (128, 1), (128, 5), (64, 1), (32, 1), (512, 1)):
    M = 16384
    g = torch.Generator(device='cpu'); g.manual_seed(1)
    x = torch.randn(G * M, C, generator=g).cuda()
    s, xtx = ops.stats(x, groups=G)
    mm = torch.zeros(C).cuda(); mc = torch.eye(C).cuda()
    print("C=%d groups=%d: K2 %.1f us" % (C, G, t(lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, mm, mc, x.device, want_scale=True, groups=G))))
