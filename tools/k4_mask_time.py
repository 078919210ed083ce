"""K4 behind a ReLU'd site: the bit-mask form (wc_bwd_reduce_mask_f32) against the fp32-y form (wc_bwd_reduce_relu_f32) and the
unmasked K4; K3 with and without the mask output.  HIP events, headline site."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x); gy = torch.randn(N, H, H, C, generator=g).cuda()
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
yr, bits = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for _ in range(10): t(lambda: ops.stream_copy(x, y))
f = lambda fn: " ".join("%.1f" % v for v in sorted(t(fn) for _ in range(5)))
print("K3 relu            us:", f(lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan, relu=True)))
print("K3 relu + bit mask us:", f(lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan, relu=True, want_mask=True)))
print("K4 plain           us:", f(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)))
print("K4 mask from y     us:", f(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_y=yr)))
print("K4 bit mask        us:", f(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=bits)))
