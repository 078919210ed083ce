import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(N, H, H, C, generator=g).cuda(); A = (torch.randn(1, C, C, generator=g) / 16).cuda()
mu = torch.zeros(C).cuda(); b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ops.apply(x, mu, A, b, None, out=y, fast=True)
torch.cuda.synchronize()
