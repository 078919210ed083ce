"""K2 time per call for a variant library, no accuracy check (development ablations give wrong results on purpose).
usage: python tools/k2_time_only.py lib1.so lib2.so ...   (each in a fresh process)"""
import os, sys, subprocess
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].endswith(".so")):
    pass
if len(sys.argv) >= 2 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from wc_gan_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from wc_gan_amd import ops
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    M = 4096
    g = torch.Generator(device='cpu'); g.manual_seed(C + 1)
    x = torch.randn(M, C, generator=g).cuda()
    s, xtx = ops.stats(x)
    fn = lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print("%-60s C=%d K2 %.1f us" % (os.path.basename(sys.argv[2]), C, best), flush=True)
else:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", lib, "256"])
