"""why does K3 take ~7 us longer inside the layer's flow than in a loop of its own?  Single launches of K3
(HIP events around each) behind different predecessors at the headline site:
  (a) K3 itself (the timing loop's situation: x and y of the previous launch are what the caches hold)
  (b) K1 over the same x (the layer's flow: K1, then K2's 75 us on 4 workgroups, color, K3)
  (c) K1 -> K2 -> color (the real flow)
  (d) a stream copy of two OTHER 128-MiB tensors (nothing of x left in the 256-MiB memory-side cache)
  (e) 80 us of a 4-workgroup kernel (K2 alone: does the chip slow down behind a nearly idle stretch?)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x); o1 = torch.randn_like(x); o2 = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
k3 = lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan)
def k2(): ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
def flow():
    s_, x_ = ops.stats(x.view(M, C)); m_, _, W_, c_ = ops.factor(s_, x_, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
    ops.color(W_, gamma, c_)
pred = {"(a) K3": k3, "(b) K1 on x": lambda: ops.stats(x.view(M, C)), "(c) K1, K2, color": flow,
        "(d) copy of other tensors": lambda: ops.stream_copy(o1, o2), "(e) K2 alone": k2}
for _ in range(30): k3()
for name, p in pred.items():
    ts = []
    for _ in range(25):
        p()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); k3(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("K3 behind %-28s min %.1f  median %.1f  max %.1f us" % (name, ts[0], ts[len(ts) // 2], ts[-1]))
