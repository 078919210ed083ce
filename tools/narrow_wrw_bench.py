"""wc_conv_wrw_narrow_f32 against MIOpen's fp32 weight gradient (+ torch's bias reduction) at the critic's image-reading layers: eager loops,
to be run under rocprofv3 --kernel-trace (kernel durations, not the host's): tools/gpu_job_narrow_wrw.sh prints the per-kernel medians.
usage: narrow_wrw_bench.py new|miopen"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C
which = sys.argv[1] if len(sys.argv) > 1 else "new"
for (N, H, W, Ci, Co, k) in ((128, 32, 32, 3, 128, 3), (128, 16, 16, 3, 128, 1)):
    x = torch.randn(N, H, W, Ci, device='cuda'); gy = torch.randn(N, H, W, Co, device='cuda')
    w = torch.randn(Co, Ci, k, k, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.zeros(Co, device='cuda', requires_grad=True)
    y = C.narrow_in_conv(x, w, b)
    xn, gn = x.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)
    for _ in range(20):
        if which == "new":
            torch.autograd.grad(y, (w, b), gy, retain_graph=True)
        else:
            torch.ops.aten.convolution_backward(gn, xn, w, [Co], [1, 1], [k // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, True])
    torch.cuda.synchronize()
