import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wc_gan_amd import ops
torch.manual_seed(0)
for (N, H, C, groups) in ((320, 16, 256, 5), (320, 32, 256, 5), (320, 8, 256, 5), (320, 16, 128, 5)):
    M = N * H * H
    x = torch.randn(N, H, H, C, device="cuda") * 1.5 + 0.3
    G = torch.randn(1, C, C, device="cuda") / 16; B = torch.randn(1, C, device="cuda") * 0.1
    st = ops.split(x)
    mu, L, W = ops.whiten_split(st, 1e-3, 0.99, 1, None, None, groups)
    mu2, L2, W2, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None, groups)
    print(N, H, C, "W planes vs fp32 K1:", float((W - W2).abs().max() / W2.abs().max()))
    A, At, plan = ops.color(W, G, st.scale, groups)
    center, bias = ops.group_bias(mu.view(groups, C), A, B, groups, 1)
    slot = ((torch.arange(N, device="cuda", dtype=torch.int32) // (N // groups))).to(torch.int32).contiguous()
    be = ops.split_bias(A, bias, st, center)
    A32, _, plan32 = ops.color(W, G, cs, groups)
    yref = ops.apply(x, center, A32, bias, slot, plan=plan32, relu=True)
    y = ops.apply_split(st, None, A, be, slot, plan=plan, relu=True, folded=True)
    rec = ops.out_scale(G, B, C, x.device)
    planes, rec = ops.apply_split(st, None, A, be, slot, plan=plan, relu=True, folded=True, oscale=rec)
    torch.cuda.synchronize()
    back = (planes[0].double() + planes[1].double()) / float(rec[0])
    TR = 8192 // C
    for name, v in (("fp32-out", y.double()), ("planes-out", back)):
        d = (v.view(M, C) - yref.view(M, C).double()).abs()
        bad = (d > 1e-4 * float(yref.abs().max())).nonzero()
        tiles = torch.unique(bad[:, 0] // TR).tolist() if bad.shape[0] else []
        print("  ", name, "max diff %.3e" % float(d.max()), "bad", bad.shape[0], "tiles", tiles[:12], "tiles_per_wg", -(-(M // TR) // 256))
