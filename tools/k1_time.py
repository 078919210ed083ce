"""K1 / K4 launch times at full-size sites (HIP events on the launching stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for shape in ((64, 4, 4, 256), (64, 8, 8, 256), (64, 16, 16, 256), (128, 4, 4, 128), (128, 32, 32, 256), (64, 32, 32, 256), (128, 16, 16, 256), (128, 32, 32, 128), (128, 48, 48, 256), (128, 64, 64, 128)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda'); gy = torch.randn(*shape, device='cuda'); mu = torch.zeros(C, device='cuda')
    print(shape, "K1 stage %.1f us   K4 stage %.1f us" % (t(lambda: ops.stats(x.view(-1, C))), t(lambda: ops.bwd_reduce(x, mu, gy, None, 1))))
for shape in ((320, 32, 32, 256), (320, 16, 16, 256), (320, 8, 8, 256)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda')
    print(shape, "groups=5 K1 stage %.1f us" % t(lambda: ops.stats(x.view(-1, C), groups=5)))
from wc_gan_amd import functional as F
for shape in ((320, 32, 32, 256), (320, 16, 16, 256)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda')
    mm = torch.zeros(C, device='cuda'); mc = torch.eye(C, device='cuda')
    with torch.no_grad():
        print(shape, "groups=5 whole grouped forward site %.1f us" % t(lambda: F.whiten_color_grouped(x, 5, None, None, None, mm, mc)))
