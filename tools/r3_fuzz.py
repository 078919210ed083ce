"""Randomised shapes through the round-3 routes against the routes they replace (development safety net for the shape-dependent dispatch):
  wc_whiten_f32            == wc_stats_f32 + wc_factor_f64                 (bit for bit)
  wc_apply_planes_f32      == wc_apply_mask_f32 (planes vs fp32 y to 2^-20 of max, masks equal)
  wc_bwd_reduce_bits_f32 + wc_bwd_apply_bits_f32 == wc_bwd_reduce_mask_f32 + wc_bwd_apply_scaled_f32   (R, gsum, scales equal; dx to 1e-6)
usage: r3_fuzz.py [seed] [iterations]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device='cuda')
bad = ran = [0, 0, 0]
bad = [0, 0, 0]; ran = [0, 0, 0]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    C = int(rng.choice([64, 128, 256, 256, 256]))
    H = int(rng.choice([4, 6, 8, 12, 16, 24, 32]))
    N = int(rng.choice([16, 33, 64, 100, 128, 160, 320]))
    if N * H * H * C > 60e6 or N * H * H < 64: continue
    Kc = int(rng.choice([1, 1, 3, 10]))
    groups = int(rng.choice([1, 1, 1, 5])) if N % 5 == 0 else 1
    shape = (N, H, H, C); M = N * H * H
    x = dev((rng.standard_normal(shape) * np.exp(rng.uniform(-1, 1, C)) + rng.uniform(-1, 1, C)).astype(np.float32))
    gy = dev((rng.standard_normal(shape) * 1e-2).astype(np.float32))
    gamma = dev((rng.standard_normal((Kc, C, C)) / np.sqrt(C)).astype(np.float32)); beta = dev((0.1 * rng.standard_normal((Kc, C))).astype(np.float32))
    slot = dev(rng.integers(0, Kc, N).astype(np.int32), torch.int32) if Kc > 1 else None
    tag = f"C={C} N={N} H={H} Kc={Kc} groups={groups}"
    try:
        # whiten
        s, xtx = ops.stats(x.view(M, C), groups)
        mu1, L1, W1, cs1 = ops.factor(s, xtx, M // groups, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True, groups=groups)
        mu2, L2, W2, cs2 = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None, groups)
        ran[0] += 1
        if not (torch.equal(mu1, mu2) and torch.equal(W1, W2) and torch.equal(cs1, cs2) and torch.equal(torch.tril(L1), torch.tril(L2))):
            bad[0] += 1; print("WHITEN differs", tag)
        if groups > 1: continue
        A, At, plan = ops.color(W1, gamma, cs1)
        if M % 32 != 0 or plan is None: continue
        y, mask = ops.apply(x, mu1, A, beta, slot, plan=plan, relu=True, want_mask=True)
        if ops.apply_planes_supported(shape):
            rec = ops.out_scale(gamma, beta, C, x.device)
            planes, rec, pmask = ops.apply_planes(x, mu1, A, beta, slot, plan, rec, relu=True, want_mask=True)
            back = (planes[0].double() + planes[1].double()) / float(rec[0])
            ran[1] += 1
            if float((back - y.double()).abs().max()) > float(y.abs().max()) * 2.0 ** -20 or not torch.equal(mask, pmask):
                bad[1] += 1; print("PLANES differ", tag, float((back - y.double()).abs().max()) / float(y.abs().max()))
        if ops.bwd_bits_supported(shape, slot is not None):
            R1, g1, gm, sc1 = ops.bwd_reduce(x, mu1, gy, slot, Kc, want_scales=True, relu_mask=mask)
            R2, g2, sc2 = ops.bwd_reduce(x, mu1, gy, slot, Kc, want_scales=True, relu_mask=mask, write_masked=False)
            _, _, S, gmean = ops.bwd_factor(R1, g1, W1, L1, gamma, A, M, 1e-3, 1, True)
            dx1 = ops.bwd_apply(gm, x, mu1, At, S, gmean, slot, scales=sc1)
            dx2 = ops.bwd_apply(gy, x, mu1, At, S, gmean, slot, scales=sc2, relu_mask=mask)
            ran[2] += 1
            if not (torch.equal(R1, R2) and torch.equal(g1, g2) and torch.equal(sc1, sc2)) or float((dx1 - dx2).abs().max()) > 1e-6 * float(dx1.abs().max()):
                bad[2] += 1; print("BITS differ", tag, float((dx1 - dx2).abs().max()) / float(dx1.abs().max()))
    except Exception as e:
        bad[0] += 1; print("EXCEPTION", tag, repr(e)[:200])
torch.cuda.synchronize()
print("ran (whiten, planes, bits):", ran, " bad:", bad)
sys.exit(1 if sum(bad) else 0)
