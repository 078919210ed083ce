import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import wc_gan_amd.generator as gen
import wc_gan_amd.functional as WF
from wc_gan_amd import ops
from wc_gan_amd.generator import make_generator
from wc_gan_amd.layers import statistic_groups
torch.manual_seed(11)
F = 256
G = make_generator(block_sizes=(F,) * 3, resamples=("UP",) * 3, first_block_shape=(4, 4, F), block_norm='d', last_norm='d',
                   block_after_norm='uconv', last_after_norm='uconv').cuda()
z = torch.randn(320, 128, device="cuda")
with torch.no_grad():
    G(z[:64])
snap = [t.detach().clone() for t in list(G.parameters()) + list(G.buffers())]
def restore():
    with torch.no_grad():
        for t, s in zip(list(G.parameters()) + list(G.buffers()), snap): t.copy_(s)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
outs = {}
def hook(name):
    def f(mod, inp, out):
        st = WF.split_of(out) if torch.is_tensor(out) else None
        pl = getattr(out, '_wc_planes', None) if torch.is_tensor(out) else None
        if st is not None: v = ops.unsplit(st)
        elif pl is not None: v = (pl[0].float() + pl[1].float()) / pl[2][0]
        else: v = out
        outs.setdefault(cur[0], {})[name] = v.detach().clone()
    return f
cur = [None]
for i, b in enumerate(G.blocks):
    b.register_forward_hook(hook(f"block{i}"))
    b.bn1.register_forward_hook(hook(f"block{i}.bn1"))
    b.bn2.register_forward_hook(hook(f"block{i}.bn2"))
    b.shortcut.register_forward_hook(hook(f"block{i}.shortcut"))
G.final_norm.register_forward_hook(hook("final_norm"))
for mode in ("grouped", "eval"):
    for on in (True, False):
        restore(); gen.SPLIT_PRODUCER = on; cur[0] = (mode, on)
        with torch.no_grad():
            if mode == "grouped":
                G.train()
                with statistic_groups(5): img = G(z)
            else:
                G.eval(); img = G(z[:64]); G.train()
        outs[(mode, on)]["img"] = img
    a, b = outs[(mode, True)], outs[(mode, False)]
    for k in a:
        print(mode, k, tuple(a[k].shape), "rel diff planes vs fp32: %.3e" % rel(a[k], b[k]), "nan" if not bool(torch.isfinite(a[k]).all()) else "")
gen.SPLIT_PRODUCER = True
