"""Round 5: one WC site as the generator runs it -- producer
(wc_resadd_split_f32) -> whiten_color on the handle (K1 + K2 on planes, K3 on planes with the ReLU bit mask) -> backward (K4 / K6
reading x from the same planes); the grouped critic-phase form; and the form whose K3 writes the next convolution's planes.
Run under rocprofv3 --kernel-trace; tools/site_timeline_print.py <trace> resadd_sample lists the launches of the last call of each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import functional as F
C = 256
gamma = (torch.randn(1, C, C, device='cuda') / 16).requires_grad_(True); beta = torch.zeros(1, C, device='cuda', requires_grad=True)
mm = torch.zeros(C, device='cuda'); mc = torch.eye(C, device='cuda')
h = torch.randn(128, 32, 32, C, device='cuda', requires_grad=True); s = torch.randn(128, 16, 16, C, device='cuda', requires_grad=True)
gy = torch.randn(128, 32, 32, C, device='cuda')
for _ in range(6):
    x = F.residual_add(h, s, True, planes=True, x32=False, stat_groups=1)
    y = F.whiten_color(x, gamma, beta, None, mm, mc, True, relu=True)
    y.backward(gy)
    for t in (gamma, beta, h, s):
        t.grad = None                      # no accumulation adds in the trace: a site's gradients are written, not summed, in the generator's flow
torch.cuda.synchronize()
hg = torch.randn(320, 32, 32, C, device='cuda'); sg = torch.randn(320, 16, 16, C, device='cuda')
with torch.no_grad():
    for _ in range(6):
        xg = F.residual_add(hg, sg, True, planes=True, stat_groups=5)
        F.whiten_color_grouped(xg, 5, gamma.detach(), beta.detach(), None, mm, mc, relu=True)
torch.cuda.synchronize()
for _ in range(6):
    x = F.residual_add(h, s, True, planes=True, x32=False, stat_groups=1)
    hh = F.whiten_color(x, gamma, beta, None, mm, mc, True, relu=True, planes=True)
    assert getattr(hh, '_wc_planes', None) is not None
    hh.backward(gy)
    for t in (gamma, beta, h, s):
        t.grad = None
torch.cuda.synchronize()
