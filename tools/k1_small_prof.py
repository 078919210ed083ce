"""K1 / K4 stages at the small (exact float64 path) sites, 50 calls each: run under rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
for shape in ((128, 4, 4, 256), (128, 8, 8, 256)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda'); gy = torch.randn(*shape, device='cuda'); mu = torch.zeros(C, device='cuda')
    for _ in range(50):
        ops.stats(x.view(-1, C)); ops.bwd_reduce(x, mu, gy, None, 1)
    torch.cuda.synchronize()
