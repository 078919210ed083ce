"""Round 6: every WC site of the generator's update pass, one after the other, as the generator runs it -- for the
CIFAR-10 unconditional recipe its 7 sites (N = 128 = batch 64 x generator_batch_multiple 2, C = 256, H = 4 .. 32) and the final sites of the
other three configurations (C = 128 at 32 x 32 with 10 classes' tables in the blocks / final uconv, 48 x 48 x 256, 64 x 64 x 128).
A site on the planes route is fed by the block's residual add (its producer, listed on its own line); the others take fp32.
Run under rocprofv3 --kernel-trace (tools/gpu_job_sites_all.sh); tools/sites_all_print.py reads the trace: per site forward us, backward us,
launches, and the time during which no kernel that streams the activation tensor runs ("HBM idle": the small-matrix chain and the gaps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import functional as F
SITES = [  # (label, N, H, C, relu epilogue)
    ("cifar10 uncond  block1.bn1   4x4", 128, 4, 256, True), ("cifar10 uncond  block1.bn2   8x8", 128, 8, 256, True),
    ("cifar10 uncond  block2.bn1   8x8", 128, 8, 256, True), ("cifar10 uncond  block2.bn2  16x16", 128, 16, 256, True),
    ("cifar10 uncond  block3.bn1  16x16", 128, 16, 256, True), ("cifar10 uncond  block3.bn2  32x32", 128, 32, 256, True),
    ("cifar10 uncond  final       32x32", 128, 32, 256, True),
    ("cifar10 cond    final       32x32 C=128", 128, 32, 128, True), ("stl10           final       48x48 C=256", 128, 48, 256, True),
    ("tiny-imagenet   final       64x64 C=128", 128, 64, 128, True)]
# which sites the block's residual add feeds (bn1 of blocks 2.. and the final site); bn2 sites read a convolution's fp32 output, block 1's
# bn1 the dense layer's
FED = {2, 4, 6, 7, 8, 9}
mark = torch.arange(4096, device='cuda', dtype=torch.float32)
def gap():
    # a marker launch the trace can be cut at (a scan kernel: nothing else in this script launches one), between idle stretches
    torch.cuda.synchronize(); time.sleep(0.002); torch.cumsum(mark, 0); torch.cuda.synchronize(); time.sleep(0.002)
for k, (label, N, H, C, relu) in enumerate(SITES):
    torch.manual_seed(k)
    gamma = (torch.randn(1, C, C, device='cuda') / C ** 0.5).requires_grad_(True); beta = torch.zeros(1, C, device='cuda', requires_grad=True)
    mm = torch.zeros(C, device='cuda'); mc = torch.eye(C, device='cuda')
    planes = k in FED and F.split_route_supported((N, H, H, C), True)
    if planes:
        h = torch.randn(N, H, H, C, device='cuda', requires_grad=True); s = torch.randn(N, H // 2, H // 2, C, device='cuda', requires_grad=True)
    else:
        x32 = torch.randn(N, H, H, C, device='cuda', requires_grad=True)
    gy = torch.randn(N, H, H, C, device='cuda')
    def fwd():
        x = F.residual_add(h, s, True, planes=True, x32=False, stat_groups=1) if planes else x32
        return F.whiten_color(x, gamma, beta, None, mm, mc, True, relu=relu)
    for _ in range(3):                    # warm-up (workspaces, first-call paths)
        fwd().backward(gy)
        for t in (gamma, beta) + ((h, s) if planes else (x32,)): t.grad = None
    gap()
    with torch.no_grad():
        for _ in range(3): fwd()          # cluster A: forward only, three calls
    gap()
    for _ in range(3):                    # cluster B: forward + backward, three calls
        fwd().backward(gy)
        for t in (gamma, beta) + ((h, s) if planes else (x32,)): t.grad = None
    gap()
    print("site %d: %s  route %s" % (k, label, "planes (producer = residual add)" if planes else "fp32"), flush=True)
