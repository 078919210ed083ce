"""Many G+D steps (graph replay) on synthetic data: everything stays finite, the losses stay in a sane band."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
# "real" images with structure: smooth random fields, so that the critic has something to learn
base = torch.nn.functional.interpolate(torch.randn(5 * 64, 3, 8, 8, generator=g), scale_factor=4, mode='bilinear').clamp(-1, 1)
reals = [base[i * 64:(i + 1) * 64].permute(0, 2, 3, 1).contiguous().cuda() for i in range(5)]
replay = tr.capture(reals)
hist = []
dall = []
for i in range(n):
    d, gl = replay()
    dall.append(d.detach().clone() if torch.is_tensor(d) else torch.tensor(float(d)))
    if i % 25 == 0 or i == n - 1:
        hist.append((i, float(d), float(gl)))
torch.cuda.synchronize()
dall = torch.stack([t.reshape(()).float().cpu() for t in dall])
print("critic loss over the %d steps: min %.6f, steps with d < 5e-4 (every hinge margin met or nearly: exactly-zero gradients reach the splits): %d %s"
      % (n, float(dall.min()), int((dall < 5e-4).sum()), (dall < 5e-4).nonzero().reshape(-1)[:12].tolist()))
# the gated second passes the history-scaled splits took (round 6): per role, summed over every convolution of both networks
from wc_gan_amd import conv as C
tot = {'x': [0, 0], 'g': [0, 0]}
for net in (tr.G, tr.D):
    for m in net.modules():
        book = m.__dict__.get('_wc_split_hist')
        if book:
            for role, h in book.items():
                tot[role][0] += 1; tot[role][1] += int(h[0].view(torch.int32)[C.HIST_REDO])
print("history-scaled splits: %d input sites (second launch: %s) / %d output-gradient sites; second passes taken: inputs %d, output gradients %d"
      % (tot['x'][0], "on" if C._guarded('x') else "off", tot['g'][0], tot['x'][1], tot['g'][1]))
w = torch.cat([p.detach().reshape(-1) for p in list(tr.G.parameters()) + list(tr.D.parameters())])
mm = torch.cat([b.detach().reshape(-1) for b in tr.G.buffers() if b.dtype == torch.float32])
print("finite weights:", bool(torch.isfinite(w).all()), " finite buffers:", bool(torch.isfinite(mm).all()), " max |w|", float(w.abs().max()))
print(" ".join(f"[{i}: d {d:.3f} g {gl:.3f}]" for i, d, gl in hist))
tr.G.eval()
with torch.no_grad():
    img = tr.G(torch.randn(16, 128, device='cuda'), torch.zeros(16, 1, dtype=torch.int32, device='cuda'))
print("eval images finite:", bool(torch.isfinite(img).all()), " range", float(img.min()), float(img.max()), " std", float(img.std()))
