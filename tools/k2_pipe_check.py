"""K2 with the inverse inside the factor launch (tri_inverse_role): W L = I and L L^T = T residuals in float64 for several
widths / group counts, repeated (the hand-off is a race if it is wrong), and the time per call.  Run once per mode:
default (one launch), WC_K2_SPLIT=1 (the four-waves-per-column inverse as its own launch), WC_K2_TWO_LAUNCH=1 (round-2a kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if len(sys.argv) > 2: _lib.LIB_PATH = sys.argv[2]      # a variant library (tools/build_var.py wc_small ...)
from wc_gan_amd import ops
def t(fn, it=30, loops=5):
    """(best, median) of `loops` back-to-back loops of `it` calls each, per call.  Back to back because a single eager call between two
    events is host-bound (4 launches: +8-10 us); several loops because an eager call allocates its outputs and now and then the allocator's
    housekeeping lands inside a loop (a single loop's mean once carried that as +500 us per call)"""
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(loops):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for C, G in ((256, 1), (256, 5), (128, 1), (128, 5), (64, 1), (32, 1), (96, 2), (224, 3), (256, 8)):
    M = 4096
    g = torch.Generator(device='cpu'); g.manual_seed(C + G)
    mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.5 * (torch.randn(C, 4, generator=g) @ torch.randn(4, C, generator=g))
    x = (torch.randn(G * M, C, generator=g) @ mix + 0.3).cuda()
    s, xtx = ops.stats(x, groups=G)
    bad = 0; emax = 0.0
    for r in range(reps):
        out = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, groups=G)
        mu, L, W = out[0], out[1], out[2]
        L = L.view(G, C, C); W = W.view(G, C, C)
        I = torch.eye(C, dtype=torch.float64, device='cuda')
        e = float((W @ L - I).abs().max())
        up = float(torch.triu(W, 1).abs().max()) + float(torch.triu(L, 1).abs().max())
        emax = max(emax, e + up)
        if not (e < 1e-9 and up == 0.0): bad += 1
    worst = max(worst, emax)
    tt = t(lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, groups=G))
    print("C=%d groups=%d: max |W L - I| + upper = %.2e, bad %d / %d, K2 %.1f us (median of 5 loops %.1f)" % (C, G, emax, bad, reps, tt[0], tt[1]), flush=True)
print("worst", worst)
