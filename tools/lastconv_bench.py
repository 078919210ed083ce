"""The generator's last convolution (3x3, 256 -> 3 channels) as one GEMM + col2im against MIOpen's direct kernels:
y[p] = sum_taps Z[p + offset][tap] with Z = x @ W_all (M x 256 @ 256 x 27), then F.fold -- the same map."""
import os, sys, torch, torch.nn.functional as F
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it

def conv_as_gemm_fold(x_nhwc, w, b):
    N, H, W, C = x_nhwc.shape
    O = w.shape[0]
    wall = w.flip(2, 3).permute(1, 0, 2, 3).reshape(C, O * 9)            # [c, (o, a, b)] = w[o, c, 2-a, 2-b]
    z = x_nhwc.reshape(N * H * W, C) @ wall                              # (M, 27)
    cols = z.view(N, H * W, O * 9).transpose(1, 2)
    return F.fold(cols, (H, W), kernel_size=3, padding=1) + b.view(1, O, 1, 1)      # NCHW (N, 3, H, W)

for N in (128, 320):
    x = torch.randn(N, 32, 32, 256, device='cuda', requires_grad=True)
    w = (torch.randn(3, 256, 3, 3, device='cuda') / 48).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(3, device='cuda', requires_grad=True)
    def f_ref(): return F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)
    def f_new(): return conv_as_gemm_fold(x, w, b)
    ref, got = f_ref(), f_new()
    err = ((ref - got).abs().max() / ref.abs().max()).item()
    g = torch.randn_like(ref)
    def fb(f):
        def run():
            y = f(); y.backward(g); x.grad = None; w.grad = None; b.grad = None
        return run
    with torch.no_grad():
        a, s = t(f_ref), t(f_new)
    print(f"N={N}: forward MIOpen {a:.3f} ms, GEMM+fold {s:.3f} ms; fwd+bwd {t(fb(f_ref)):.3f} vs {t(fb(f_new)):.3f} ms; max rel diff {err:.1e}", flush=True)
