import os, sys, torch
sys.path.insert(0, os.getcwd())
from wc_gan_amd.train import CIFAR10_UNCOND, CIFAR10_COND, build_trainer
for cfg in (CIFAR10_UNCOND, CIFAR10_COND):
    tr = build_trainer(cfg, 'cuda', batch_size=8, training_ratio=1)
    real = torch.rand(8, 32, 32, 3, device='cuda') * 2 - 1
    rc = torch.randint(0, 10, (8, 1), device='cuda', dtype=torch.int32)
    tr.d_step(real, rc); tr.step([real]) if not tr.conditional else None
    tr.G.eval()
    with torch.no_grad():
        z = torch.randn(16, 128, device='cuda'); c = torch.randint(0, 10, (16, 1), device='cuda', dtype=torch.int32)
        a = tr.G(z, c); b = tr.G(z, c)
    print(a.shape, float((a - b).abs().max()), bool(torch.isfinite(a).all()))
    assert a.shape == (16, 32, 32, 3) and float((a - b).abs().max()) < 1e-5 and torch.isfinite(a).all()
    tr.G.train()
    x = tr.G(z, c)          # train mode with grad
    x.mean().backward()
    print(cfg['conditional'], 'ok', float(a.abs().max()))
