"""Runs only K1 (wc_stats_f32) and K4 (wc_bwd_reduce_f32) at the headline site 128x32x32x256 on the SURVEY section 8d
kernel-bench input, n times each, so that a rocprofv3 pass isolates xty_f16x3_kernel<256,false> / <256,true>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()
gy = torch.randn(N, H, H, C, generator=g).cuda()
mu = x.view(M, C).mean(0)
for _ in range(n):
    ops.stats(x.view(M, C))
    ops.bwd_reduce(x, mu, gy, None, 1)
torch.cuda.synchronize()
