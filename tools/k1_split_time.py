"""K1 on the planes (wc_stats_split_f16x2) against the fp32-input K1 at the headline site (or N H C from argv): HIP-event times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 32, 256)
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); y = torch.empty_like(x)
xs = ops.split(x)
s1, x1 = ops.stats(x.view(M, C), G)
s2, x2 = ops.stats_split(xs, G)
print("xtx split vs fp32 path: %.3g   sum: %.3g" % (float((x2 - x1).abs().max() / x1.abs().max()), float((s2 - s1).abs().max() / s1.abs().max())))
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for _ in range(10): t(lambda: ops.stream_copy(x, y))
a = sorted(t(lambda: ops.stats_split(xs, G)) for _ in range(7)); b = sorted(t(lambda: ops.stats(x.view(M, C), G)) for _ in range(7))
f = lambda v: " ".join("%.1f" % q for q in v)
print("K1 stage on planes us:", f(a)); print("K1 stage fp32 us:", f(b))
