"""Runs K3 on the pre-split planes (wc_apply_split_f16x2, bias folded: one launch) and, alternating with it, the fp32-input K3
and K1 on the planes / on fp32 at the headline site on the SURVEY section 8d kernel-bench input, n times each: the target of
the rocprofv3 --kernel-trace / --pmc passes of round 3 (apply_split_kernel<256,false>, affine_ring_kernel<256,false>,
xtx_split_kernel<256>, xty_f16x3_kernel<256,false>)."""
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
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
xs = ops.split(x)
A2, At2, plan2 = ops.color(W, gamma, xs.scale)
be = ops.split_bias(A2, b, xs, mu)
for _ in range(n):
    ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, folded=True)
    ops.apply(x, mu, A, b, None, out=y, plan=plan)
    ops.stats_split(xs)
    ops.stats(x.view(M, C))
torch.cuda.synchronize()
