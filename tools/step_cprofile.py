import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=64, training_ratio=5)
g = torch.Generator(device='cpu'); g.manual_seed(0)
reals = [torch.rand(64, 32, 32, 3, generator=g).cuda() * 2 - 1 for _ in range(5)]
for _ in range(5): tr.step(reals)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(5): tr.step(reals)
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / 5:.2f} ms/step (enqueue only, profiled)")
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
