"""Randomised shapes through the C ABI against float64 numpy: K1, K4 (with and without class slots), K3, K6.
A development safety net for the shape-dependent dispatch (exact / fast paths, quadrant scheme, tile straddling)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device='cuda')
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    C = int(rng.choice([32, 64, 96, 128, 160, 256, 320]))
    H = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 32]))
    N = int(rng.choice([1, 2, 5, 16, 33, 64, 100, 128]))
    if N * H * H * C > 40e6 or N * H * H < 2: continue
    Kc = int(rng.choice([1, 1, 3, 10]))
    x = (rng.standard_normal((N, H, H, C)) * np.exp(rng.uniform(-2, 2, C)) + rng.uniform(-1, 1, C)).astype(np.float32)
    gy = (rng.standard_normal((N, H, H, C)) * 1e-2 * np.exp(rng.uniform(-2, 2, C))).astype(np.float32)
    M = N * H * H
    X = x.reshape(M, C).astype(np.float64); G = gy.reshape(M, C).astype(np.float64)
    mu = X.mean(0).astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32)
    st = dev(slot, torch.int32) if Kc > 1 else None
    tag = f"C={C} N={N} H={H} Kc={Kc}"
    try:
        s, xtx = ops.stats(dev(x).view(M, C))
        nat = np.sqrt(np.outer((X ** 2).sum(0), (X ** 2).sum(0))) + 1e-300
        e1 = np.abs((xtx.cpu().numpy() - X.T @ X) / nat).max()
        R, gs = ops.bwd_reduce(dev(x), dev(mu), dev(gy), st, Kc)
        f = X - mu.astype(np.float64)
        e4 = 0.0
        for k in range(Kc):
            sel = np.repeat(slot == k, H * H) if Kc > 1 else np.ones(M, bool)
            ref = f[sel].T @ G[sel]
            nat4 = np.sqrt(np.outer((f[sel] ** 2).sum(0), (G[sel] ** 2).sum(0))) + 1e-300
            e4 = max(e4, np.abs((R[k].cpu().numpy() - ref) / nat4).max() if sel.any() else np.abs(R[k].cpu().numpy()).max())
        A = (rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32); b = rng.standard_normal((Kc, C)).astype(np.float32)
        y = ops.apply(dev(x), dev(mu), dev(A), dev(b), st, fast=True)
        fr = f.reshape(N, -1, C)
        yref = np.einsum('npc,nco->npo', fr, A.astype(np.float64)[slot]) + b.astype(np.float64)[slot][:, None, :]
        e3 = np.abs(y.cpu().numpy().reshape(yref.shape) - yref).max() / max(np.abs(yref).max(), 1e-30)
        At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
        S = rng.standard_normal((C, C)).astype(np.float32) * 1e-3; S = (S + S.T) / 2
        gm = (rng.standard_normal(C) * 1e-4).astype(np.float32)
        dx = ops.bwd_apply(dev(gy), dev(x), dev(mu), dev(At), dev(S), dev(gm), st, fast=True)
        dref = np.einsum('npc,nco->npo', G.reshape(N, -1, C), At.astype(np.float64)[slot]) + fr @ S.astype(np.float64) - gm.astype(np.float64)
        e6 = np.abs(dx.cpu().numpy().reshape(dref.shape) - dref).max() / max(np.abs(dref).max(), 1e-30)
        ok = e1 < 2e-6 and e4 < 2e-6 and e3 < 1e-5 and e6 < 1e-5
        if not ok: bad += 1
        print(("ok  " if ok else "BAD ") + tag + f"  K1 {e1:.1e} K4 {e4:.1e} K3 {e3:.1e} K6 {e6:.1e}", flush=True)
    except Exception as ex:
        print("EXC " + tag + " " + repr(ex)[:200], flush=True); bad += 1
print("bad:", bad)
