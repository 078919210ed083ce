"""The power floor of K3: the layers' kernel at the headline site (apply_split_kernel<256, false, true, false>:
planes in, ReLU + bit mask) on the real planes, on ALL-ZERO planes (same launch, same traffic, no switching activity in the matrix
pipe's operands) and the stream copy of the same bytes, alternating, raw C-ABI calls between HIP events.  usage: python tools/k3_zero_planes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
mu, L, W, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
xs = ops.split(x)
A, At, plan, be = ops.color_split(W, gamma, xs, mu, b)
zero = ops.SplitTensor(torch.zeros_like(xs.planes), xs.center, xs.scale, xs.flag, xs.shape)
y = torch.empty_like(x); y2 = torch.empty_like(x)
def raw(st):
    ops.TRACE = []
    try:
        ops.apply_split(st, None, A, be, None, plan=plan, out=y, relu=True, folded=True, want_mask=True)
        return ops.TRACE[0][2], ops.TRACE[0][3]
    finally:
        ops.TRACE = None
k_real, keep1 = raw(xs)
k_zero, keep2 = raw(zero)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {"real": [], "zero": [], "copy": []}
for _ in range(5):
    res["real"].append(timed(k_real)); res["zero"].append(timed(k_zero)); res["copy"].append(timed(lambda: ops.stream_copy(x, y2)))
alg = 2 * M * C * 4 + (C * C + C) * 4 + M * C // 8
for k, v in res.items():
    v = sorted(v)
    print(f"{k:5s} min {v[0]:6.2f} median {v[2]:6.2f} max {v[-1]:6.2f} us" + (f"   = {alg / v[2] / 1e3 / 8000:.3f} of the 8 TB/s peak" if k != "copy" else f"   ({2 * M * C * 4 / v[2] / 1e3:.0f} GB/s)"))
print("real / copy = %.3f of the stream copy's rate;  zero-data floor / copy = %.3f" % ((alg / sorted(res['real'])[2]) / (2 * M * C * 4 / sorted(res['copy'])[2]), (alg / sorted(res['zero'])[2]) / (2 * M * C * 4 / sorted(res['copy'])[2])))
