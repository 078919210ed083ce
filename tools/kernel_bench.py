"""Times the individual WC stages at a given site shape (default: the headline 128x32x32x256)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
N, H, C = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 32, 256)))
Kc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(N, H, H, C, generator=g).cuda(); gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(Kc, C, C, generator=g) / C ** 0.5).cuda(); beta = torch.zeros(Kc, C).cuda()
slot = torch.randint(0, Kc, (N,), generator=g).to(torch.int32).cuda() if Kc > 1 else None
M = N * H * H
mm = torch.zeros(C).cuda(); mc = torch.eye(C).cuda()
s, xtx = ops.stats(x.view(M, C)); mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, mm, mc, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs); y = torch.empty_like(x)
R, gsum = ops.bwd_reduce(x, mu, gy, slot, Kc)
dg, db, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)
by = 2 * M * C * 4
res = {
 'stream_copy': t(lambda: ops.stream_copy(x, y)),
 'stats(K1)': t(lambda: ops.stats(x.view(M, C))),
 'factor(K2)': t(lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, mm, mc, x.device)),
 'color(+plan)': t(lambda: ops.color(W, gamma, cs)),
 'apply planned(K3)': t(lambda: ops.apply(x, mu, A, beta, slot, out=y, plan=plan)),
 'apply fast(K3)': t(lambda: ops.apply(x, mu, A, beta, slot, out=y, fast=True)),
 'apply exact(K3)': t(lambda: ops.apply(x, mu, A, beta, slot, out=y, fast=False)),
 'bwd_reduce(K4)': t(lambda: ops.bwd_reduce(x, mu, gy, slot, Kc)),
 'bwd_factor(K5)': t(lambda: ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)),
 'bwd_apply fast(K6)': t(lambda: ops.bwd_apply(gy, x, mu, At, S, gm, slot, fast=True)),
 'bwd_apply exact(K6)': t(lambda: ops.bwd_apply(gy, x, mu, At, S, gm, slot, fast=False)),
}
print(f"shape N={N} H={H} C={C} Kc={Kc}  M={M}  x bytes={M*C*4/2**20:.0f} MiB")
for k, v in res.items():
    print(f"  {k:22s} {v:9.1f} us   ({by/v/1e3:7.0f} GB/s if 2*M*C*4 B)")
