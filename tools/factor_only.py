"""Runs only the small fp64 stages (K2 factor, K5 bwd_factor) so a rocprofv3 --kernel-trace --stats run isolates them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M = 131072
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(G * M, C, generator=g).cuda()
s, xtx = ops.stats(x, groups=G)
mm = torch.zeros(C).cuda(); mc = torch.eye(C).cuda()
for _ in range(20):
    out = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, mm, mc, x.device, want_scale=True, groups=G)
torch.cuda.synchronize()
