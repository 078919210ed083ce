import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
reals = [torch.rand(8, 32, 32, 3, device='cuda') * 2 - 1 for _ in range(2)]
g = torch.Generator(device='cuda'); g.manual_seed(5)
noise = {n: (torch.randn(n, 128, device='cuda', generator=g), torch.randint(0, 10, (n, 1), device='cuda', dtype=torch.int32, generator=g)) for n in (16, 8)}
flat = len(sys.argv) > 1 and sys.argv[1] == 'flat'
def make(ovl=True):
    torch.manual_seed(21)
    tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=8, training_ratio=2, seed=9, flat_buckets=flat)
    tr._noise = lambda n: noise[n]
    tr.overlap_g_forward = ovl
    return tr
def weights(tr):
    torch.cuda.synchronize()
    return torch.cat([p.detach().reshape(-1).clone() for p in list(tr.G.parameters()) + list(tr.D.parameters())])
def eager(steps, ovl=True):
    tr = make(ovl)
    for _ in range(steps): tr.step(reals)
    return weights(tr)
def graph(kind, ovl=True):
    tr = make(ovl)
    rp = (tr.capture_segments if kind == 'seg' else tr.capture)(reals, warmup=1)
    for _ in range(2): rp()
    return weights(tr)
spread = lambda x, y: (round(float((x - y).abs().max()), 6), round(float(((x - y).abs() > 2e-5).float().mean()), 4))
e1, e2 = eager(3), eager(3)
print("eager vs eager", spread(e1, e2))
print("eager(no overlap) vs eager", spread(eager(3, False), e1))
print("whole graph vs eager", spread(graph('whole'), e1))
print("segments vs eager", spread(graph('seg'), e1))
print("segments (no overlap) vs eager", spread(graph('seg', False), e1))
print("whole (no overlap) vs eager", spread(graph('whole', False), e1))
print("--- which eager step count does warmup 1 + 2 replays match?")
gw = graph('whole')
for k in (1, 2, 3, 4, 5):
    print(k, spread(gw, eager(k)))
tr = make(); rp = tr.capture(reals, warmup=1); w_after_capture = weights(tr)
print("after capture (1 warm-up, 0 replays) vs eager(1)", spread(w_after_capture, eager(1)), "vs eager(2)", spread(w_after_capture, eager(2)))
print("--- eager with every cache rebuilt at every call vs plain eager")
from wc_gan_amd import _state, conv
orig = conv._cached_image
def nocache(w, key, geom, k_axis, n_axis): return conv.weight_image(w, geom, k_axis, n_axis)
conv._cached_image = nocache
e_nc = eager(3)
conv._cached_image = orig
print("no image cache vs eager", spread(e_nc, e1), " no image cache vs whole graph", spread(e_nc, gw))
p = next(tr.G.parameters()); v0 = p._version; tr.step(reals); print("param version bump per step:", p._version - v0)
