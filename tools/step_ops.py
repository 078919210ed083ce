"""One eager G+D step under torch.profiler: which aten / autograd ops own the small torch kernels of the step (device time per op name)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
from torch.profiler import profile, ProfilerActivity
from wc_gan_amd.train import CONFIGS, build_trainer
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cifar10_uncond"]
tr = build_trainer(cfg, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
H, W, Ci = cfg['image_shape']
reals = [torch.rand(64, H, W, Ci, generator=g).cuda() * 2 - 1 for _ in range(5)]
for _ in range(4): tr.step(reals, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(reals, None)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None: dt = getattr(e, 'self_cuda_time_total', 0)
    if dt > 0: rows.append((dt, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"device time {tot / 1e3:.2f} ms")
rows = [r for r in rows if r[2].startswith('aten::') or 'Backward' in r[2] or r[2][0].isupper() or r[2].startswith('_')]
for dt, n, k, sh in rows[:60]:
    print(f"{dt / 1e3:7.3f} ms {n:4d} x {k[:60]:60s} {sh}")
