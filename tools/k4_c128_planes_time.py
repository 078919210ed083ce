"""Development: K4 on planes at the C = 128 sites (wc_bwd_reduce_xsplit_f32, HIP events over back-to-back loops) with the library in WC_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if os.environ.get("WC_LIB"): _lib.LIB_PATH = os.environ["WC_LIB"]
from wc_gan_amd import ops
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for shape in ((128, 32, 32, 128), (64, 64, 64, 128)):
    C = shape[-1]
    x = torch.randn(*shape, device='cuda'); gy = torch.randn(*shape, device='cuda'); mu = torch.zeros(C, device='cuda')
    xs = ops.split(x)
    print(shape, "K4 on planes, stage (all its launches) %.1f us" % min(t(lambda: ops.bwd_reduce_xsplit(xs, mu, gy, None, 1)) for _ in range(5)), flush=True)
