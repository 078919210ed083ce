"""Development: times the planned K3 apply at the headline site with every library variant under csrc/build/abl/."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
N, H, C = (int(v) for v in sys.argv[2:5])
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
yref = ops.apply(x, mu, A, b, None, fast=False)
for _ in range(5): ops.apply(x, mu, A, b, None, out=y, plan=plan)
err = ((y - yref).abs().max() / yref.abs().max()).item()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.apply(x, mu, A, b, None, out=y, plan=plan)
e1.record(); torch.cuda.synchronize(); print(f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us   max err vs exact {err:.2e}")
''' % ROOT
shape = sys.argv[1:4] if len(sys.argv) > 3 else ["128", "32", "256"]
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "abl", "lib_*.so"))):
    r = subprocess.run([sys.executable, "-c", child, lib] + shape, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
