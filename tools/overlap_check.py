import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
reals = [torch.rand(8, 32, 32, 3, device='cuda') * 2 - 1 for _ in range(2)]
def run(overlap):
    torch.manual_seed(11)
    tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=8, training_ratio=2, seed=77)
    tr.overlap_g_forward = overlap
    for _ in range(2): d, g = tr.step(reals)
    torch.cuda.synchronize()
    return float(d), float(g), torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()])
a = run(True); b = run(False); c = run(False); d = run(True)
for n, (x, y) in {'ovl-seq': (a, b), 'seq-seq': (b, c), 'ovl-ovl': (a, d)}.items():
    df = (x[2] - y[2]).abs()
    print(n, 'loss d/g', abs(x[0]-y[0]), abs(x[1]-y[1]), 'max', float(df.max()), 'frac>2e-5', float((df > 2e-5).float().mean()))
