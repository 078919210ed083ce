"""Development: times wc_apply_split_f16x2 at the headline site (or N H C from argv) with every library under csrc/build/var/;
stamp builds (-DWC_SPLIT_STAMPS=1) also print where a wave's time went."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, ctypes, torch
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
xs = ops.split(x)
A, At, plan = ops.color(W, gamma, xs.scale)
yref = ops.apply(x, mu, A, b, None, fast=False)
be = ops.split_bias(A, b, xs, mu)
lib = _lib.load()
dbg = torch.zeros(256 * 8 * 16, dtype=torch.int64, device='cuda')
stamps = "STAMPS" in sys.argv[1]
if stamps:
    lib.wc_dev_split_dbg.argtypes = [ctypes.c_void_p]; lib.wc_dev_split_dbg(dbg.data_ptr())
mk = torch.empty(M // 32, C, dtype=torch.int32, device='cuda')
run = lambda: ops.apply_split(xs, None, A, be, None, plan=plan, out=y, folded=True, relu=True, want_mask=True, _mask_out=mk)    # as the layers run it
for _ in range(5): run()
err = ((y - yref.clamp_min(0)).abs().max() / yref.abs().max()).item()
ts = []
for rep in range(25):       # one launch at a time behind a register-only spin (bench.py's time_isolated)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(400000); e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("apply (ReLU + mask) us: min %%.1f median %%.1f max %%.1f   max err vs exact %%.2e" %% (ts[0], ts[len(ts) // 2], ts[-1], err))
if stamps:
    d = dbg.view(256, 8, 16).cpu().double()
    t0 = d[..., 4].min()
    print("  timeline (us from the first workgroup's start): last start %%.2f | loop starts %%.2f .. %%.2f | ends %%.2f .. %%.2f" %% (
        (d[..., 4].max() - t0) / 100, (d[..., 5].min() - t0) / 100, (d[..., 5].max() - t0) / 100, (d[..., 6].min() - t0) / 100, (d[..., 6].max() - t0) / 100))
    ends = (d[..., 6].amax(dim=1) - t0) / 100
    print("  workgroup end times, sorted deciles:", " ".join("%%.1f" %% v for v in ends.sort().values[::26].tolist()))
    w, l, st, tot = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    f = lambda t: "mean %%.0f min %%.0f max %%.0f" %% (t.mean(), t.min(), t.max())
    print("  per wave, whole tile loop (s_memtime ticks): wait+dma-issue", f(w), "| k-loop", f(l), "| stores", f(st), "| total", f(tot))
    print("  waves 0-3 total", f(tot[:, :4]), "| waves 4-7 total", f(tot[:, 4:]))
    ghz = tot / ((d[..., 6] - d[..., 5]) * 10.0)          # s_memtime ticks of the tile loop over its s_memrealtime span (100 MHz, chip-wide)
    print("  shader clock over the tile loop (ticks / wall time): mean %%.3f GHz  min %%.3f  max %%.3f" %% (ghz.mean(), ghz.min(), ghz.max()))
    # the prologue and tile 0 (round 6): cycles after the wave's first vector-memory instruction
    pr = d[..., 8:16]; base = pr[..., 0:1]
    names = ["my pieces of tile 0 landed", "tile 0 landed for all eight waves (first MFMA may issue)", "table k-step 0 landed", "table, last k-step landed",
             "my pieces of tile 1 landed", "tile 1 landed for all (tile 1's loop begins)", "tile 2's loop begins"]
    for i, nm in enumerate(names):
        v = (pr[..., i + 1] - base[..., 0])[pr[..., i + 1] > 0]
        if v.numel(): print("    %%-58s mean %%6.0f  min %%6.0f  max %%6.0f cycles" %% (nm, v.mean(), v.min(), v.max()))
''' % ROOT
shape = sys.argv[1:4] if len(sys.argv) > 3 else ["128", "32", "256"]
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_*.so"))):
    r = subprocess.run([sys.executable, "-c", child, lib] + shape, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), r.stdout.strip() or r.stderr.strip()[-600:], flush=True)
