import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
rng = np.random.default_rng(0)
N, H, C = int(os.environ.get('NB', 64)), 32, 256
shape = (N, H, H, C)
x = rng.standard_normal(shape).astype(np.float32) + 0.5
gy = (rng.standard_normal(shape) * 1e-3).astype(np.float32)
mu = np.full(C, 0.5, np.float32)
A = (rng.standard_normal((1, C, C)) / 16).astype(np.float32); At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
S = rng.standard_normal((C, C)).astype(np.float32) * 1e-4; S = (S + S.T) / 2
gm = np.zeros(C, np.float32)
if os.environ.get('MODE') == 'gyonly': S[:] = 0
if os.environ.get('MODE') == 'xonly': At[:] = 0
d = lambda a, t=torch.float32: torch.tensor(a, dtype=t, device='cuda')
xd, gyd, mud, Atd, Sd, gmd = d(x), d(gy), d(mu), d(At), d(S), d(gm)
scales = ops.bwd_reduce(xd, mud, gyd, None, 1, want_scales=True)[-1]
ref = torch.tensor(gy.reshape(-1, C).astype(np.float64) @ At[0].astype(np.float64) + (x.reshape(-1, C).astype(np.float64) - mu) @ S.astype(np.float64), device='cuda')
scale = float(ref.abs().max())
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    dx = ops.bwd_apply(gyd, xd, mud, Atd, Sd, gmd, None, fast=True, scales=scales).view(-1, C).double()
    err = ((dx - ref).abs() / scale).view(-1, 16, 16, 16)
    bad = (err > 1e-4)
    if bad.any():
        idx = bad.nonzero()
        tiles = sorted(set(idx[:, 0].tolist())); rows = sorted(set(idx[:, 1].tolist())); cbs = sorted(set(idx[:, 2].tolist()))
        print(f"run {it}: tiles {tiles[:8]} (i={[t // 128 for t in tiles[:8]]}, pair={[t % 128 for t in tiles[:8]]}) rows {rows} colblocks {cbs}", flush=True)
print("done")
