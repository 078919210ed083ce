"""K5 (wc_bwd_factor_f64) replayed from a hipGraph under rocprofv3 --kernel-trace: what its chain of small dependent launches costs
inside a graph (kernel durations and the gaps between them) -- the number a persistent one-launch form has to beat."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
C = 256; M = 131072
g = torch.Generator(device="cpu"); g.manual_seed(0)
x = torch.randn(M, C, generator=g).cuda(); gy = torch.randn(M, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
mu, L, W, cs = ops.whiten(x, 1e-3, 0.99, 1, None, None)
A, At, plan = ops.color(W, gamma, cs)
R, gs = ops.bwd_reduce(x.view(128, 32, 32, C), mu, gy.view(128, 32, 32, C), None, 1)
for _ in range(3): ops.bwd_factor(R, gs, W, L, gamma, A, M, 1e-3, 1, True)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(10): ops.bwd_factor(R, gs, W, L, gamma, A, M, 1e-3, 1, True)
gr.replay(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
print("K5 inside a graph: %.1f us per call (10 calls replayed)" % (e0.elapsed_time(e1) * 100))
e0.record()
for _ in range(10): ops.bwd_factor(R, gs, W, L, gamma, A, M, 1e-3, 1, True)
e1.record(); torch.cuda.synchronize()
print("K5 eager: %.1f us per call" % (e0.elapsed_time(e1) * 100))
