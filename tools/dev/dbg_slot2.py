import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wc_gan_amd import ops
torch.manual_seed(0)
N, H, C, groups = 320, 32, 256, 5
M = N * H * H
x = torch.randn(N, H, H, C, device="cuda") * 1.5 + 0.3
G = torch.randn(1, C, C, device="cuda") / 16; B = torch.randn(1, C, device="cuda") * 0.1
st = ops.split(x)
mu, L, W = ops.whiten_split(st, 1e-3, 0.99, 1, None, None, groups)
A, At, plan = ops.color(W, G, st.scale, groups)
center, bias = ops.group_bias(mu.view(groups, C), A, B, groups, 1)
slot = ((torch.arange(N, device="cuda", dtype=torch.int32) // (N // groups))).to(torch.int32).contiguous()
be = ops.split_bias(A, bias, st, center)
y = ops.apply_split(st, None, A, be, slot, plan=plan, relu=True, folded=True)
TR = 32
for trial in range(3):
    for name, kw in (("slot", dict(slot=slot, A=A, be=be, plan=plan)), ("noslot(table 0)", dict(slot=None, A=A[:1].contiguous(), be=be[:1].contiguous(), plan=None))):
        for mask in (False, True):
            rec = ops.out_scale(G, B, C, x.device)
            yref = y if kw["slot"] is not None else ops.apply_split(st, None, kw["A"], kw["be"], None, relu=True, folded=True)
            out = ops.apply_split(st, None, kw["A"], kw["be"], kw["slot"], plan=kw["plan"], relu=True, folded=True, want_mask=mask, oscale=rec)
            torch.cuda.synchronize()
            planes, rec = out[0], out[1]
            back = (planes[0].double() + planes[1].double()) / float(rec[0])
            d = (back.view(M, C) - yref.view(M, C).double()).abs()
            bad = (~(d <= 1e-4 * float(yref.abs().max()))).nonzero()
            tiles = torch.unique(bad[:, 0] // TR) if bad.shape[0] else torch.zeros(0, dtype=torch.long)
            wg = torch.unique(tiles // 40).tolist() if kw["slot"] is not None else torch.unique(tiles % 256).tolist()
            pos = torch.unique(tiles % 40).tolist() if kw["slot"] is not None else torch.unique(tiles // 256).tolist()
            print(trial, name, "mask", mask, "scale", float(rec[0]), "bad", bad.shape[0], "ntiles bad", tiles.numel(), "wgs", wg[:10], len(wg), "tile pos in wg", pos[:12])
