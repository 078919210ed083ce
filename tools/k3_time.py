"""K3 (planned fast apply) and stream copy at the headline site: HIP-event times, 9 x 20 launches each (sorted)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for _ in range(10): t(lambda: ops.stream_copy(x, y))          # clocks up
k3 = sorted(t(lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan)) for _ in range(9))
cp = sorted(t(lambda: ops.stream_copy(x, y)) for _ in range(9))
print("K3 us:", " ".join("%.1f" % v for v in k3), "| copy us:", " ".join("%.1f" % v for v in cp), "| ratio of medians %.3f" % (cp[4] / k3[4]))
