"""Development: times wc_factor_f64 (K2) at C (default 256) with every library variant under csrc/build/abl/."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
C = int(sys.argv[2]); M = 131072
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(M, C, generator=g).cuda()
s, xtx = ops.stats(x)
mm = torch.zeros(C).cuda(); mc = torch.eye(C).cuda()
f = lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, mm, mc, x.device, want_scale=True)
for _ in range(5): f()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize(); print(f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
''' % ROOT
C = sys.argv[1] if len(sys.argv) > 1 else "256"
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "abl", "lib_*.so"))):
    r = subprocess.run([sys.executable, "-c", child, lib, C], capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
