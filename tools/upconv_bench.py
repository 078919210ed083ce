"""conv3x3(upsample2x(h)) against its sub-pixel form (four 2x2 convolutions on the low-resolution input, interleaved):
same linear map, 16 instead of 36 tap products per low-resolution pixel.  Times forward and forward+backward."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it

def subpixel_kernels(w):
    r0 = torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], dim=2)      # output rows 2y:   low-res rows y-1, y
    r1 = torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], dim=2)      # output rows 2y+1: low-res rows y, y+1
    ks = []
    for r in (r0, r1):
        c0 = torch.stack([r[..., 0], r[..., 1] + r[..., 2]], dim=3)
        c1 = torch.stack([r[..., 0] + r[..., 1], r[..., 2]], dim=3)
        ks += [c0, c1]
    return ks                                                           # (dy, dx) = (0,0), (0,1), (1,0), (1,1)

def upconv_subpixel(h, w, b):
    N, C, H, W = h.shape
    out = torch.empty(N, w.shape[0], 2 * H, 2 * W, device=h.device, dtype=h.dtype).contiguous(memory_format=torch.channels_last)
    for i, k in enumerate(subpixel_kernels(w)):
        dy, dx = i // 2, i % 2
        y = F.conv2d(h, k.contiguous(memory_format=torch.channels_last), b, padding=1)      # (H+1, W+1)
        out[:, :, dy::2, dx::2] = y[:, :, dy:dy + H, dx:dx + W]
    return out

def transposed_kernel(w):
    # (Cout, Cin, 3, 3) -> (Cin, Cout, 4, 4): output row 2y-1+a receives x[y] times the taps that read an upsampled copy of y
    rows = torch.stack([w[:, :, 2], w[:, :, 1] + w[:, :, 2], w[:, :, 0] + w[:, :, 1], w[:, :, 0]], dim=2)          # (Cout,Cin,4,3)
    k = torch.stack([rows[..., 2], rows[..., 1] + rows[..., 2], rows[..., 0] + rows[..., 1], rows[..., 0]], dim=3)   # (Cout,Cin,4,4)
    return k.transpose(0, 1)

def upconv_transposed(h, w, b):
    return F.conv_transpose2d(h, transposed_kernel(w).contiguous(memory_format=torch.channels_last), b, stride=2, padding=1)

for (N, C, H) in [(128, 256, 16), (320, 256, 16), (128, 256, 8), (320, 256, 8), (128, 256, 4)]:
    h = torch.randn(N, C, H, H, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3, device='cuda') / 48).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.zeros(C, device='cuda', requires_grad=True)
    ref = F.conv2d(F.interpolate(h, scale_factor=2, mode='nearest'), w, b, padding=1)
    g = torch.randn_like(ref)
    got = upconv_subpixel(h, w, b)
    err = ((ref - got).abs().max() / ref.abs().max()).item()
    got_t = upconv_transposed(h, w, b)
    err_t = ((ref - got_t).abs().max() / ref.abs().max()).item()
    def f_tr(): return upconv_transposed(h, w, b)
    def fb_tr():
        y = f_tr(); y.backward(torch.randn_like(ref) if False else g); h.grad = None; w.grad = None; b.grad = None
    def f_ref(): return F.conv2d(F.interpolate(h, scale_factor=2, mode='nearest'), w, b, padding=1)
    def f_sub(): return upconv_subpixel(h, w, b)
    def fb_ref():
        y = f_ref(); y.backward(g); h.grad = None; w.grad = None; b.grad = None
    def fb_sub():
        y = f_sub(); y.backward(g); h.grad = None; w.grad = None; b.grad = None
    with torch.no_grad():
        a, s, tr = t(f_ref), t(f_sub), t(f_tr)
    print(f"N={N} C={C} {H}->{2*H}: forward upsample+3x3 {a:.3f} ms, sub-pixel {s:.3f}, transposed 4x4/s2 {tr:.3f} ms; fwd+bwd {t(fb_ref):.3f} / {t(fb_sub):.3f} / {t(fb_tr):.3f} ms; max rel diff {err:.1e} / {err_t:.1e}", flush=True)
