"""Times the fused spectral-norm op against torch's parametrisation on the critic's layer shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.spectral import SNConv2d
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for cin, cout, k in [(128, 128, 3), (3, 128, 3), (128, 128, 1), (256, 256, 3)]:
    a = SNConv2d(cin, cout, k, padding=k // 2).cuda()
    b = torch.nn.utils.parametrizations.spectral_norm(torch.nn.Conv2d(cin, cout, k, padding=k // 2).cuda().to(memory_format=torch.channels_last))
    g = torch.randn_like(a.weight)
    def fa():
        w = a.normalized_weight(); w.backward(g)
    def fb():
        w = b.weight; w.backward(g)
    def fa_f():
        with torch.no_grad(): a.normalized_weight()
    print(f"conv {cin}->{cout} k{k}: fused fwd+bwd {t(fa):7.1f} us (fwd only {t(fa_f):6.1f})   torch parametrisation fwd+bwd {t(fb):7.1f} us")
