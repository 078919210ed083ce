"""K3 on the pre-split input (wc_apply_split_f16x2) against the fp32-input K3 and the stream copy at the headline site:
HIP-event times, 9 x 20 launches each (sorted)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, int(os.environ.get("H", 32)), int(os.environ.get("C", 256))
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
xs = ops.split(x)
A2, At2, plan2 = ops.color(W, gamma, xs.scale)
be = ops.split_bias(A2, b, xs, mu)
y2 = ops.apply_split(xs, None, A2, be, None, plan=plan2, folded=True)
y1 = ops.apply(x, mu, A, b, None, plan=plan)
print("max |split - fp32 path| / max|y| = %.3g" % float((y2 - y1).abs().max() / y1.abs().max()))
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for _ in range(10): t(lambda: ops.stream_copy(x, y))          # clocks up
import ctypes
from wc_gan_amd import _lib
lib = _lib.load()
# the split apply without the bias pre-launch would need a dedicated entry; time the ABI call (bias kernel + apply) and the apply alone via rocprof
k3s = sorted(t(lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, folded=True)) for _ in range(9))
k3 = sorted(t(lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan)) for _ in range(9))
cp = sorted(t(lambda: ops.stream_copy(x, y)) for _ in range(9))
sp = sorted(t(lambda: ops.split(x, xs.center, xs.scale, xs.flag)) for _ in range(5))
f = lambda v: " ".join("%.1f" % q for q in v)
print("K3 split us:", f(k3s)); print("K3 fp32 us:", f(k3)); print("copy us:", f(cp)); print("split producer us:", f(sp))
print("ratio copy/K3split %.3f  copy/K3 %.3f" % (cp[4] / k3s[4], cp[4] / k3[4]))
