"""avg_pool2(conv3x3(x)) against its fused form, one 4x4 stride-2 convolution (taps = quarter-sums of the 3x3 taps):
same linear map, 16 instead of 36 tap products per low-resolution output.  Times forward and forward+backward."""
import os, sys, torch, torch.nn.functional as F
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it

def pooled_kernel(w):
    k = w.new_zeros(w.shape[0], w.shape[1], 4, 4)
    for dy in (0, 1):
        for dx in (0, 1):
            k[:, :, dy:dy + 3, dx:dx + 3] += w
    return 0.25 * k

for (N, C, H) in [(128, 128, 32), (128, 128, 16), (256, 128, 32)]:
    x = torch.randn(N, C, H, H, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3, device='cuda') / 34).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(C, device='cuda', requires_grad=True)
    def f_ref(): return F.avg_pool2d(F.conv2d(x, w, b, padding=1), 2)
    def f_fus(): return F.conv2d(x, pooled_kernel(w).contiguous(memory_format=torch.channels_last), b, stride=2, padding=1)
    ref, got = f_ref(), f_fus()
    err = ((ref - got).abs().max() / ref.abs().max()).item()
    g = torch.randn_like(ref)
    def fb(f):
        def run():
            y = f(); y.backward(g); x.grad = None; w.grad = None; b.grad = None
        return run
    with torch.no_grad():
        a, s = t(f_ref), t(f_fus)
    print(f"N={N} C={C} {H}->{H//2}: forward conv3x3+pool {a:.3f} ms, 4x4/s2 {s:.3f} ms; fwd+bwd {t(fb(f_ref)):.3f} vs {t(fb(f_fus)):.3f} ms; max rel diff {err:.1e}", flush=True)
