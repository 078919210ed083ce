"""Runs warm-up, a 0.5 s pause, then N G+D steps -- the target of `rocprofv3 --kernel-trace`; summarize_profile.py
with `gap` as its window argument then keeps only what follows the last pause (the steady-state steps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
from wc_gan_amd.train import CONFIGS, build_trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cifar10_uncond"]
tr = build_trainer(cfg, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
H, W, Ci = cfg['image_shape']
reals = [torch.rand(64, H, W, Ci, generator=g).cuda() * 2 - 1 for _ in range(5)]
K = cfg['generator']['number_of_classes']
labels = [torch.randint(0, K, (64, 1), generator=g, dtype=torch.int32).cuda() for _ in range(5)] if cfg['conditional'] else None
for _ in range(5): tr.step(reals, labels)
torch.cuda.synchronize(); time.sleep(0.5)
t0 = time.perf_counter()
for _ in range(n): tr.step(reals, labels)
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms/step over {n} steps")
