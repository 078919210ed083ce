"""One WC site through the layer path (functional.whiten_color with autograd, ReLU epilogue; the grouped critic-phase form; and the
K3 -> convolution hand-off form, whose forward writes the next convolution's planes): run under rocprofv3 --kernel-trace;
tools/site_timeline_print.py lists the launches of the last call of each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import functional as F
C = 256
gamma = (torch.randn(1, C, C, device='cuda') / 16).requires_grad_(True); beta = torch.zeros(1, C, device='cuda', requires_grad=True)
mm = torch.zeros(C, device='cuda'); mc = torch.eye(C, device='cuda')
x = torch.randn(128, 32, 32, C, device='cuda', requires_grad=True); gy = torch.randn(128, 32, 32, C, device='cuda')
for _ in range(6):
    y = F.whiten_color(x, gamma, beta, None, mm, mc, True, relu=True)
    y.backward(gy)
torch.cuda.synchronize()
xg = torch.randn(320, 32, 32, C, device='cuda')
with torch.no_grad():
    for _ in range(6): F.whiten_color_grouped(xg, 5, gamma.detach(), beta.detach(), None, mm, mc, relu=True)
torch.cuda.synchronize()
x2 = torch.randn(128, 32, 32, C, device='cuda', requires_grad=True)
for _ in range(6):
    h = F.whiten_color(x2, gamma, beta, None, mm, mc, True, relu=True, planes=True)
    assert getattr(h, '_wc_planes', None) is not None
    h.backward(gy)
torch.cuda.synchronize()
# the same K3 (ReLU + bit mask) in a loop of its own, under the same profiler: the "isolated" figure beside the in-flow one above
from wc_gan_amd import ops
M = 128 * 32 * 32
with torch.no_grad():
    xd = x.detach()
    mu_, L_, W_, cs_ = ops.whiten(xd.view(M, C), 1e-3, 0.99, 1, None, None)
    A_, At_, plan_ = ops.color(W_, gamma.detach(), cs_)
    yb = torch.empty_like(xd)
    torch.cuda.synchronize()
    for _ in range(20):
        ops.apply(xd, mu_, A_, beta.detach(), None, out=yb, plan=plan_, relu=True, want_mask=True)
torch.cuda.synchronize()
