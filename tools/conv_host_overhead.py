"""Host time to ENQUEUE one fast_conv forward / forward+backward (no synchronisation inside the timed region)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C
import torch.nn.functional as F
x = torch.randn(128, 8, 8, 128, device='cuda', requires_grad=True)
w = (torch.randn(128, 128, 3, 3, device='cuda') * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_(True)
b = torch.zeros(128, device='cuda', requires_grad=True)
def fwd(): return C.fast_conv(x, w, b, 'same')
def fb():
    y = C.fast_conv(x, w, b, 'same'); y.backward(x.detach())   # any gradient of the right shape
def mi():
    y = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1); y.backward(x.detach().permute(0, 3, 1, 2))
for name, f, n in (('forward', fwd, 300), ('forward+backward', fb, 300), ('torch conv2d fwd+bwd', mi, 300)):
    for _ in range(10): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: enqueue {1e6 * (t1 - t) / n:.1f} us/call, drained after {1e6 * (t2 - t) / n:.1f} us/call")
