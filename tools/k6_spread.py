"""K6 (onepass_ring_kernel, 128x32x32x256, with the ReLU bits) measured 124-196 us launch to launch on a fixed
input.  Where does the spread come from?  The kernel reads gy and x (268 MB) and writes dx (134 MB); the memory-side cache (MALL)
holds 256 MB, so what a launch finds there depends on what ran before it.  Four loops of 24 single launches between HIP events:
  a) back to back (the timing loops of bench.py / stage_only.py)
  b) a 1-GiB flush (fill of another buffer) in front of every launch: everything cold
  c) x and gy read once (a column-sum pass) in front of every launch: the inputs as warm as the cache can hold them
  d) as the layer runs it: K4 (reads x, gy) and K5 in front of every launch
prints min / median / max per loop and per XCD-independent statistics; usage: python tools/k6_spread.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()
gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
_, mask = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True)
R, gsum, scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask, write_masked=False)
_, _, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
k6 = lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales, relu_mask=mask)
def loop(front, n=24):
    ev = []
    for _ in range(n + 3):
        front()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); k6(); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[3:])
    return t
def k4k5():
    R_, g_, sc_ = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask, write_masked=False)
    ops.bwd_factor(R_, g_, W, L, gamma, A, M, 1e-3, 1, True)
for name, front in (("a) back to back", lambda: None), ("b) 1-GiB flush in front", lambda: flush.fill_(1.0)),
                    ("c) x, gy read once in front", lambda: (x.sum(), gy.sum())), ("d) K4 + K5 in front (the layer's flow)", k4k5)):
    t = loop(front)
    print(f"{name:42s} min {t[0]:6.1f}  median {t[len(t)//2]:6.1f}  max {t[-1]:6.1f}  max/min {t[-1]/t[0]:.2f}   (us, events around the K6 call: 2 launches = tables + kernel)")
