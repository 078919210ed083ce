"""The history-scaled split (wc_conv_split_hist_f32) against the two-launch form, per call, at the tensor sizes of the CIFAR-10 step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C
class Site: pass
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for shape in ((128, 4, 4, 256), (128, 8, 8, 256), (128, 16, 16, 256), (128, 32, 32, 128), (128, 32, 32, 256), (320, 32, 32, 256)):
    x = torch.randn(*shape, device='cuda')
    site = Site(); C.split_planes(x, site=site)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): C.split_planes(x, site=site)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(10): C.split_planes(x)
    th, tc = timeit(g.replay, 50) / 10, timeit(g2.replay, 50) / 10
    print(f"{str(shape):22s} hist {th:6.1f} us   two-launch {tc:6.1f} us per call (inside a graph of 10)", flush=True)
