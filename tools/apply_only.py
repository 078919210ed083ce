"""Runs the planned fast apply (K3) at the headline site N times: the target of the rocprofv3 --pmc passes."""
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
relu = len(sys.argv) > 2 and sys.argv[2] == 'relu'
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ops.apply(x, mu, A, b, None, out=y, plan=plan, relu=relu)
torch.cuda.synchronize()
