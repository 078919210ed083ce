"""the two levers the diagnosis of the non-gaussian families names -- shorter fp32 MFMA chains (a float64 flush every 6
accumulations instead of 12: -DXTY_SUBFLUSH=2) and the dropped lo*lo product (-DXTY_LOLO=1) -- in K1's fp32-input kernel (xty_f16x3_kernel,
covariance form), each as a build under csrc/build/var/lib_xt*.so (tools/build_var.py wc_fast_xty xtbase= xtlolo=-DXTY_LOLO=1 ...): K1's kernel time
at 128x32x32x256 (one call at a time behind a spin) and tools/seed_sweep.py --families over 3 seeds at the two C = 256 sites."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
timer = r'''
import sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
x = torch.randn(128, 32, 32, 256, device='cuda')
run = lambda: ops.stats(x.view(-1, 256))
for _ in range(3): run()
ts = []
for rep in range(15):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(600000); e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("K1 stage (sample + kernel + gate + tail) us: min %%.1f median %%.1f" %% (ts[0], ts[len(ts) // 2]))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_xt*.so"))):
    print("=====", os.path.basename(lib), flush=True)
    r = subprocess.run([sys.executable, "-c", timer, lib], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
    env = dict(os.environ, WC_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "seed_sweep.py"), "3", "--families", "128x32x32x256", "128x16x16x256"],
                       capture_output=True, text=True, timeout=1500, env=env)
    for l in r.stdout.splitlines():
        if l.startswith("WORST") or l.startswith("worst"):
            print(l, flush=True)
    if r.returncode not in (0, 1):
        print(r.stderr.strip()[-500:], flush=True)
