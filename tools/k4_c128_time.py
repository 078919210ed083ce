"""K4 stage at the C = 128 sites (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for shape in ((128, 32, 32, 128), (128, 64, 64, 128)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda'); gy = torch.randn(*shape, device='cuda'); mu = torch.zeros(C, device='cuda')
    print(shape, "K4 stage %.1f us" % min(t(lambda: ops.bwd_reduce(x, mu, gy, None, 1)) for _ in range(5)))
