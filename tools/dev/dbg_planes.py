import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wc_gan_amd import ops
torch.manual_seed(0)
N, H, C = 128, 32, 256
M = N * H * H
x = torch.randn(N, H, H, C, device="cuda")
G = torch.randn(1, C, C, device="cuda") / 16; B = torch.randn(1, C, device="cuda") * 0.1
s, xtx = ops.stats(x.view(M, C))
mu, L, W = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device)
st = ops.split(x)
A, _, plan = ops.color(W, G, st.scale)
be = ops.split_bias(A, B, st, mu)
y = ops.apply_split(st, None, A, be, None, plan=plan, relu=True, folded=True)
for mask in (False, True):
    rec = ops.out_scale(G, B, C, x.device)
    out = ops.apply_split(st, None, A, be, None, plan=plan, relu=True, folded=True, want_mask=mask, oscale=rec)
    planes, rec = out[0], out[1]
    torch.cuda.synchronize()
    sc = float(rec[0])
    back = ((planes[0].double() + planes[1].double()) / sc).view(M, C)
    d = (back - y.view(M, C).double()).abs()
    bad = (d > 1e-4).nonzero()
    print("mask", mask, "scale", sc, "max diff", float(d.max()), "n bad", bad.shape[0], "of", M * C)
    if bad.shape[0]:
        r, c = bad[:, 0], bad[:, 1]
        print(" rows%32 hist", torch.bincount(r % 32, minlength=32).tolist())
        print(" cols%32 hist", torch.bincount(c % 32, minlength=32).tolist())
        print(" tiles hist (first 20)", torch.bincount(r // 32)[:20].tolist())
        for i in range(min(5, bad.shape[0])):
            rr, cc = int(r[i]), int(c[i])
            print("  ", rr, cc, "planes", float(back[rr, cc]), "y", float(y.view(M, C)[rr, cc]))
        # is the bad value found elsewhere in the row / neighbouring rows?
        rr, cc = int(r[0]), int(c[0])
        yy = y.view(M, C)
        cand = (yy[max(rr - 16, 0):rr + 17].double() - back[rr, cc]).abs()
        w = (cand < 1e-5).nonzero()
        print("   value found at (drow, col):", [(int(a) - min(rr, 16), int(b)) for a, b in w[:6]])
