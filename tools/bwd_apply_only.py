"""Runs only K6 (wc_bwd_apply_scaled_f32, the one-pass kernel at C = 256) at the headline site n times: the target of the
rocprofv3 --pmc passes (tools/gpu_job_pmc.sh tools/bwd_apply_only.py <tag>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N, H, C = 128, 32, 256
g = torch.Generator(device="cpu"); g.manual_seed(1234)
x = (torch.randn(N, H, H, C, generator=g) + 0.5).cuda()
gy = (torch.randn(N, H, H, C, generator=g) * 1e-3).cuda()
mu = torch.full((C,), 0.5, device="cuda")
At = (torch.randn(1, C, C, generator=g) / 16).cuda()
S = torch.randn(C, C, generator=g) * 1e-4; S = ((S + S.t()) / 2).cuda()
gm = torch.zeros(C, device="cuda")
scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)[-1]
torch.cuda.synchronize()
for _ in range(n):
    ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales)
torch.cuda.synchronize()
