"""Which parameters does one G+D step leave untouched?  (development check behind tests/test_configs_gpu.py)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CONFIGS, build_trainer
name = sys.argv[1] if len(sys.argv) > 1 else 'cifar10_uncond'
cfg = CONFIGS[name]
torch.manual_seed(5)
tr = build_trainer(cfg, "cuda", training_ratio=2)
H, W, Ci = cfg['image_shape']
g = torch.Generator(device="cpu"); g.manual_seed(6)
reals = [(torch.rand(64, H, W, Ci, generator=g) * 2 - 1).cuda() for _ in range(2)]
K = cfg['generator']['number_of_classes']
labels = [torch.randint(0, K, (64, 1), generator=g, dtype=torch.int32).cuda() for _ in range(2)] if cfg['conditional'] else None
for net, nm in ((tr.G, 'G'), (tr.D, 'D')):
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    if nm == 'G':
        d, gl = tr.step(reals, labels)
        print('losses', float(d), float(gl))
    else:
        tr.step(reals, labels)
    for n, p in net.named_parameters():
        if torch.equal(before[n], p.detach()):
            print(nm, 'UNMOVED', n, tuple(p.shape), 'grad', None if p.grad is None else float(p.grad.abs().max()))
print('done')
