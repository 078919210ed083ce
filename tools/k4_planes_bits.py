"""Development: K4 on planes (wc_bwd_reduce_xsplit_f32) at 128 x 32 x 32 x 256 and 128 x 16 x 16 x 256 with the library in WC_LIB: prints a digest of R, gsum
and the scales -- two builds whose digests agree produce the same bits."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if os.environ.get("WC_LIB"): _lib.LIB_PATH = os.environ["WC_LIB"]
from wc_gan_amd import ops
for N, H, C in ((128, 32, 256), (128, 16, 256), (320, 16, 256), (128, 32, 128), (64, 64, 128)):
    M = N * H * H
    g = torch.Generator(device="cpu"); g.manual_seed(77)
    x = (torch.randn(M, C, generator=g) * (1 + 2 * torch.rand(C, generator=g)) + 0.3).view(N, H, H, C).cuda()
    gy = torch.randn(N, H, H, C, generator=g).cuda()
    mu = x.view(M, C).mean(0)
    xs = ops.split(x)
    mask = (torch.rand(M // 32, C, generator=g) * 2 ** 32).to(torch.int64).to(torch.int32).cuda() if False else None
    y = torch.empty_like(x)
    s, xtx = ops.stats(x.view(M, C))
    mu2, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
    A, At, plan = ops.color(W, torch.eye(C, device="cuda").view(1, C, C).contiguous(), cs)
    _, mask = ops.apply(x, mu2, A, torch.zeros(1, C, device="cuda"), None, plan=plan, relu=True, want_mask=True, out=y)
    for m in ((None, mask) if C == 256 else (None,)):
        out = ops.bwd_reduce_xsplit(xs, mu, gy, None, 1, relu_mask=m)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in out[:2]: h.update(t.cpu().numpy().tobytes())
        print((N, H, C), "planes x,", "mask" if m is not None else "no mask", h.hexdigest()[:16], flush=True)
        out = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=m, write_masked=False) if m is not None else ops.bwd_reduce(x, mu, gy, None, 1)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in out[:2]: h.update(t.cpu().numpy().tobytes())
        print((N, H, C), "fp32 x,  ", "mask" if m is not None else "no mask", h.hexdigest()[:16], flush=True)
