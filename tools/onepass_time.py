import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
x = torch.randn(N, H, H, C, device='cuda') + 0.5; gy = torch.randn(N, H, H, C, device='cuda') * 1e-3; mu = torch.full((C,), 0.5, device='cuda')
At = (torch.randn(1, C, C, device='cuda') / 16); S = torch.randn(C, C, device='cuda') * 1e-4; S = (S + S.t()) / 2; gm = torch.zeros(C, device='cuda')
scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)[-1]
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("K6 with scales (one pass unless WC_K6_TWO_PASS): %.1f us" % t(lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales)))
