"""Development: the fused producer (wc_resadd_stats_split_f32: sample + resadd_xtx_kernel + gate) timed with every library under
csrc/build/var/lib_rx*.so (tools/build_var.py wc_resadd rxbase= rxabl1=-DWC_RX_ABL=1 ...): one call at a time behind a register-only spin,
beside round 4's chain (wc_resadd_split_f32 + wc_stats_split_f16x2's kernel).  Ablated builds compute wrong results: only their times mean anything."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
N, H, C = 128, 32, 256
g = torch.Generator(device="cpu"); g.manual_seed(1234)
hh = torch.randn(N, H, H, C, generator=g).cuda(); ss = torch.randn(N, H // 2, H // 2, C, generator=g).cuda()
def timeit(run):
    for _ in range(3): run()
    ts = []
    for rep in range(15):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(600000); e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return "min %%.1f median %%.1f" %% (ts[0], ts[len(ts) // 2])
print("fused: " + timeit(lambda: ops.resadd_stats_split(hh, ss, True, 1)) + " | round 4 (add, then K1 kernel + tail): " +
      timeit(lambda: ops.stats_split(ops.resadd_split(hh, ss, True))), flush=True)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_rx*.so"))):
    line = os.path.basename(lib).ljust(18)
    r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True, timeout=300)
    line += " | per call us: %s" % (r.stdout.strip() or ("FAILED " + r.stderr.strip()[-300:]))
    print(line, flush=True)
