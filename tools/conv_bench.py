"""fast_conv (split-fp16 MFMA implicit GEMM) against torch/MIOpen fp32 and an fp64 reference: accuracy and time."""
import sys, time
import torch
import torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C


def ref(x, w, b, kind, dtype):
    xn, wn = x.permute(0, 3, 1, 2).to(dtype), w.to(dtype)
    bb = None if b is None else b.to(dtype)
    if kind == 'same':
        y = F.conv2d(xn, wn, bb, padding=w.shape[2] // 2)
    elif kind == 'down':
        y = F.conv2d(xn, wn, bb, stride=2, padding=1)
    else:
        y = F.conv_transpose2d(xn, wn, bb, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6


def case(N, H, W, Ci, Co, kind, k=3, bench=True):
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Ci, device='cuda') * 1.7 + 0.2
    if kind == 'up':
        w = (torch.randn(Ci, Co, 4, 4, device='cuda') / (Ci * 4) ** 0.5).contiguous(memory_format=torch.channels_last)
    elif kind == 'down':
        w = (torch.randn(Co, Ci, 4, 4, device='cuda') / (Ci * 16) ** 0.5).contiguous(memory_format=torch.channels_last)
    else:
        w = (torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Co, device='cuda') * 0.1
    x.requires_grad_(True); w.requires_grad_(True)
    if not C.supported(x, w, kind):
        print(f"{kind} {N}x{H}x{W} {Ci}->{Co}: unsupported"); return
    y = C.fast_conv(x, w, b, kind)
    gy = torch.randn_like(y)
    dx, dw = torch.autograd.grad(y, (x, w), gy)
    y64 = ref(x, w, b, kind, torch.float64)
    dx64, dw64 = torch.autograd.grad(y64, (x, w), gy.double())
    y32 = ref(x, w, b, kind, torch.float32)
    dx32, = torch.autograd.grad(y32, (x,), gy)
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    msg = (f"{kind:5s} N{N} {H}x{W} {Ci}->{Co} k{w.shape[2]}: y {rel(y, y64):.1e} (miopen {rel(y32, y64):.1e})  "
           f"dx {rel(dx, dx64):.1e} (miopen {rel(dx32, dx64):.1e})  dw {rel(dw, dw64.to(dw.dtype).double()):.1e}")
    if bench:
        xd = x.detach(); wd = w.detach()
        planes = C.split_planes(xd)
        (gf, kf, nf), (gb, kb, nb) = C._geoms(kind, N, H, W, wd)
        img = C.weight_image(wd, gf, kf, nf)
        t_kernel = timeit(lambda: C.run(planes, img, gf, b))
        t_split = timeit(lambda: C.split_planes(xd))
        t_img = timeit(lambda: C.weight_image(wd, gf, kf, nf))
        t_mi = timeit(lambda: ref(xd, wd, b, kind, torch.float32))
        gpl = C.split_planes(gy); imgb = C.weight_image(wd, gb, kb, nb)
        t_bk = timeit(lambda: C.run(gpl, imgb, gb))
        xn, gn = xd.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)
        stride = [1, 1] if kind == 'same' else [2, 2]; pad = [w.shape[2] // 2] * 2 if kind == 'same' else [1, 1]
        t_mb = timeit(lambda: torch.ops.aten.convolution_backward(gn, xn, wd, None, stride, pad, [1, 1], kind == 'up', [0, 0], 1, [True, False, False]))
        xpl = C.split_planes(xd)
        kf_, nf_ = kf, nf
        t_wk = timeit(lambda: C.weight_gradient(xpl, gpl, gf, wd, kf_, nf_))
        t_mw = timeit(lambda: torch.ops.aten.convolution_backward(gn, xn, wd, None, stride, pad, [1, 1], kind == 'up', [0, 0], 1, [False, True, False]))
        flop = 2.0 * gf.N * gf.H * gf.W * gf.nphase * gf.ntaps * gf.Cin * gf.Cout
        msg += (f"\n      fwd {t_kernel:7.1f} us ({flop / t_kernel / 1e6:5.0f} TF) + split {t_split:5.1f} + image {t_img:5.1f} | miopen {t_mi:7.1f} us ({flop / t_mi / 1e6:4.0f} TF)"
                f" | bwd-data {t_bk:7.1f} us vs miopen {t_mb:7.1f} | wrw {t_wk:7.1f} vs miopen {t_mw:7.1f}")
    print(msg, flush=True)


if __name__ == '__main__':
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    case(2, 8, 8, 32, 128, 'same', bench=False)
    case(4, 8, 8, 64, 128, 'same', k=1, bench=False)
    case(2, 8, 8, 128, 128, 'down', bench=False)
    case(2, 8, 8, 128, 128, 'up', bench=False)
    case(3, 12, 12, 96, 256, 'same', bench=False) if (3 * 144) % 128 == 0 else None
    if not quick:
        for N in (128, 320):
            case(N, 32, 32, 256, 256, 'same')
            case(N, 16, 16, 256, 256, 'same')
            case(N, 8, 8, 256, 256, 'same')
            case(N, 16, 16, 256, 256, 'up')
            case(N, 8, 8, 256, 256, 'up')
        case(128, 16, 16, 128, 128, 'same')
        case(128, 8, 8, 128, 128, 'same')
        case(128, 32, 32, 128, 128, 'down')
        case(128, 32, 32, 256, 256, 'same', k=1)
