import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = 128, 32, 256
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(N, H, H, C, generator=g).cuda(); A = (torch.randn(1, C, C, generator=g) / 16).cuda()
mu = torch.zeros(C).cuda(); b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for v in [0]:
    os.environ['WC_FAST_VARIANT'] = str(v)
    print('variant', v, '%.1f us' % t(lambda: ops.apply(x, mu, A, b, None, out=y, fast=True)), flush=True)
