"""The class-conditional CIFAR-10 recipe (scripts/cifar10_resnet_sn_cond.sh) for some replayed steps: finite, and the
throughput for the record (not the headline config)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_COND, build_trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tr = build_trainer(CIFAR10_COND, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
reals = [torch.rand(64, 32, 32, 3, generator=g).cuda() * 2 - 1 for _ in range(5)]
labels = [torch.randint(0, 10, (64, 1), generator=g, dtype=torch.int32).cuda() for _ in range(5)]
replay = tr.capture(reals, labels)
for _ in range(5): replay()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n): d, gl = replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
w = torch.cat([p.detach().reshape(-1) for p in list(tr.G.parameters()) + list(tr.D.parameters())])
print(f"cond recipe: {dt * 1e3:.2f} ms/step ({64 / dt:.0f} images/sec), finite {bool(torch.isfinite(w).all())}, d {float(d):.3f} g {float(gl):.3f}")
