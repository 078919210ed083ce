"""Development: the one-pass K6 (bits mode: fp32 x, ReLU bit mask) timed with every library under csrc/build/var/ (tools/build_var.py wc_fast
k6base= k6abl<bits>=-DWC_K6_ABL=<bits> ...): one launch at a time behind a register-only spin.  Ablated builds compute wrong
results: only their times mean anything."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(1234)
x = (torch.randn(M, C, generator=g) + 0.2).view(N, H, H, C).cuda()
gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
_, mask = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True, out=y)
R, gsum, scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)
_, _, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)
run = lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales, relu_mask=mask)
for _ in range(3): run()
ts = []
for rep in range(15):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(600000); e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("min %%.1f median %%.1f" %% (ts[0], ts[len(ts) // 2]), flush=True)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_k6*.so"))):
    line = os.path.basename(lib).ljust(16)
    r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True, timeout=300)
    line += " | call (tables + kernel) us: %s" % (r.stdout.strip() or ("FAILED " + r.stderr.strip()[-200:]))
    print(line, flush=True)
