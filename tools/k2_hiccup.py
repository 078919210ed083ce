"""Per-call times of K2 (HIP events around every call) and a bitwise check of every result against the first call's:
looks for rare long calls (a hand-off protocol that stalls would show here) -- run with and without WC_K2_TWO_LAUNCH=1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for C, G in ((256, 1), (256, 5), (128, 5), (224, 3), (64, 1)):
    M = 4096
    g = torch.Generator(device='cpu'); g.manual_seed(C + G)
    x = torch.randn(G * M, C, generator=g).cuda()
    s, xtx = ops.stats(x, groups=G)
    f = lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, groups=G)
    ref = f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    outs = []
    for a, b in ev:
        a.record(); outs.append(f()); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    bad = sum(1 for o in outs if not (torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])))
    print("C=%d groups=%d: n %d median %.1f us  p99 %.1f  max %.1f  calls > 1 ms: %d  results differing from the first: %d"
          % (C, G, n, ts[n // 2], ts[int(n * 0.99)], ts[-1], sum(t > 1000 for t in ts), bad), flush=True)
