"""Where does the K3 -> convolution hand-off pay?  Per site size, replayed from a hipGraph (no host time): the fp32 route
(K3 + mask, then the convolution's absmax + split) against the planes route (out_scale, K3 writing the planes + mask, the gated launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops, conv
def graph_time(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
C = 256
for N, H in ((128, 8), (128, 16), (128, 32), (320, 8), (320, 16), (320, 32)):
    M = N * H * H
    x = torch.randn(N, H, H, C, device='cuda'); gamma = torch.randn(1, C, C, device='cuda') / 16; b = torch.zeros(1, C, device='cuda')
    mu, L, W, cs = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
    A, At, plan = ops.color(W, gamma, cs)
    y = torch.empty_like(x)
    def fp32_route():
        yy, m = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True, out=y)
        conv.split_planes(yy)
    def planes_route():
        rec = ops.out_scale(gamma, b, C, x.device)
        ops.apply_planes(x, mu, A, b, None, plan, rec, relu=True, want_mask=True)
    print("N %3d H %2d: fp32 route %.1f us, planes route %.1f us" % (N, H, graph_time(fp32_route), graph_time(planes_route)), flush=True)
