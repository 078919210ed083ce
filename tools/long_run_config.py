"""N graph-replayed G+D steps of any BASELINE configuration on synthetic data: losses and weights stay finite."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CONFIGS, build_trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cifar10_cond"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = CONFIGS[name]
tr = build_trainer(cfg, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
H, W, Ci = cfg['image_shape']
reals = [(torch.rand(64, H, W, Ci, generator=g) * 2 - 1).cuda() for _ in range(5)]
K = cfg['generator']['number_of_classes']
labels = [torch.randint(0, K, (64, 1), generator=g, dtype=torch.int32).cuda() for _ in range(5)] if cfg['conditional'] else None
replay = tr.capture(reals, labels)
hist = []
for i in range(n):
    d, gl = replay()
    if i % 20 == 0 or i == n - 1: hist.append((i, float(d), float(gl)))
torch.cuda.synchronize()
w = torch.cat([p.detach().reshape(-1) for p in list(tr.G.parameters()) + list(tr.D.parameters())])
print(name, "finite weights:", bool(torch.isfinite(w).all()), " ".join(f"[{i}: d {d:.3f} g {gl:.3f}]" for i, d, gl in hist))
