"""GPU time of the phases of one G+D step (events on the main stream; the G forward of the update runs on the side stream)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
reals = [torch.rand(64, 32, 32, 3, generator=g).cuda() * 2 - 1 for _ in range(5)]
for _ in range(5): tr.step(reals)
torch.cuda.synchronize()
def timed(f, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
t_gen, (fakes, clss) = timed(lambda: tr.generate(5))
t_d, _ = timed(lambda: tr.d_step(reals[0], fake=fakes[0], cls=clss[0]))
def gfwd():
    z, cls = tr._noise(128); return tr.G(z, cls), cls
t_gf, gen = timed(gfwd)
t_g, _ = timed(lambda: tr.g_step(None))
tr.overlap_g_forward = False
t_seq, _ = timed(lambda: tr.step(reals))
tr.overlap_g_forward = True
t_ovl, _ = timed(lambda: tr.step(reals))
print(f"generate(5) {t_gen:.2f} ms | one D step {t_d:.2f} ms (x5 = {5 * t_d:.2f}) | G forward alone {t_gf:.2f} | whole G step {t_g:.2f} | step sequential {t_seq:.2f} overlapped {t_ovl:.2f}")
