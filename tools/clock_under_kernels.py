"""Round 6: the shader clock the chip holds under each streaming kernel of the WC path (VERDICT r5 item 3's kernels and K3), 128 x 32 x 32 x 256.
A one-wave probe (tools/probe/clock_probe.hip) samples (s_memtime, s_memrealtime) on a side stream while the kernel under test runs `reps` times back
to back on the main stream between two s_memrealtime stamps; clock = shader cycles per 10-ns tick over the samples inside the window.
The probe's wave lives on ONE XCD: it reads that XCD's clock (the dies share one power budget)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wc_gan_amd import ops, functional as F
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "bin", "libclock_probe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.clock_stamp_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()
gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
y = torch.empty_like(x); y2 = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
xs = ops.split(x)
A, At, plan = ops.color(W, gamma, xs.scale)
be = ops.split_bias(A, b, xs, mu)
mk = torch.empty(M // 32, C, dtype=torch.int32, device='cuda')
ws = ops.apply_split_workspace(C, 1, x.device)
A32, At32, plan32 = ops.color(W, gamma, cs)
S = torch.randn(C, C, generator=g) * 1e-4; S = ((S + S.t()) / 2).cuda()
gm = torch.zeros(C, device='cuda')
scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)[-1]
h = torch.randn(N, H, H, C, generator=g).cuda(); sh = torch.randn(N, H // 2, H // 2, C, generator=g).cuda()
kernels = [
    ("stream copy (read + write, 268 MB)", lambda: ops.stream_copy(x, y2)),
    ("K3 apply_split_kernel<256,false,true,false> (planes in, ReLU + mask)", lambda: ops.apply_split(xs, None, A, be, None, plan=plan, out=y, folded=True, relu=True, want_mask=True, _mask_out=mk, ws=ws)),
    ("K3 affine_ring_kernel (fp32 in, fp32 out)", lambda: ops.apply(x, mu, A32, b, None, out=y, plan=plan32)),
    ("K1 xty_f16x3_kernel<256,false> (wc_stats_f32, fp32 in)", lambda: ops.stats(x.view(M, C))),
    ("producer resadd_xtx_kernel<256> (add + planes + K1's partials)", lambda: F.residual_add(h, sh, True, planes=True, x32=False, stat_groups=1)),
    ("K4 xty_f16x3_kernel<256,true> (wc_bwd_reduce_f32)", lambda: ops.bwd_reduce(x, mu, gy, None, 1)),
    ("K6 onepass_ring_kernel (wc_bwd_apply_scaled_f32)", lambda: ops.bwd_apply(gy, x, mu, At32, S, gm, None, scales=scales)),
]
side = torch.cuda.Stream()
_w = torch.zeros(64, dtype=torch.int64, device='cuda')
with torch.cuda.stream(side):
    probe.clock_probe_launch(_w.data_ptr(), 16, 1, side.cuda_stream)              # the module load, off the first measurement
probe.clock_stamp_launch(_w.data_ptr() + 256, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
NS, reps = 40000, 10
for name, fn in [k for k in kernels for _ in range(2)]:
    with torch.no_grad():
        for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = torch.zeros(2 * NS, dtype=torch.int64, device='cuda'); st = torch.zeros(2, dtype=torch.int64, device='cuda')
    with torch.cuda.stream(side):
        probe.clock_probe_launch(buf.data_ptr(), NS, 4, side.cuda_stream)          # several ms of samples
    main = torch.cuda.current_stream().cuda_stream
    torch.cuda._sleep(200000)                                                       # the chip idle in front (as bench.py's isolated timing)
    probe.clock_stamp_launch(st.data_ptr(), main)
    with torch.no_grad():
        for _ in range(reps): fn()
    probe.clock_stamp_launch(st.data_ptr() + 8, main)
    torch.cuda.synchronize()
    t = buf.view(NS, 2).cpu().double(); w0, w1 = (float(v) for v in st.cpu())
    tk, rt = t[:, 0], t[:, 1]
    ins = (rt >= w0) & (rt <= w1)
    idx = ins.nonzero().reshape(-1)
    if idx.numel() < 8:
        print("%-74s window %.1f us at %.0f: the probe (%.0f .. %.0f) did not overlap it" % (name, (w1 - w0) / 100, w0 / 100, float(rt[0]) / 100, float(rt[-1]) / 100)); continue
    i0, i1 = int(idx[0]), int(idx[-1])
    ghz = (tk[i1] - tk[i0]) / ((rt[i1] - rt[i0]) * 10.0)
    # idle clock: the samples in front of the window
    pre = (rt < w0).nonzero().reshape(-1)
    ghz_pre = float((tk[pre[-1]] - tk[pre[0]]) / ((rt[pre[-1]] - rt[pre[0]]) * 10.0)) if pre.numel() > 8 else float('nan')
    # the lowest clock over any 20-us stretch inside the window
    lo = 9.9
    j = i0
    for i in range(i0, i1):
        while j < i1 and rt[j] - rt[i] < 2000: j += 1
        if rt[j] - rt[i] >= 2000: lo = min(lo, float((tk[j] - tk[i]) / ((rt[j] - rt[i]) * 10.0)))
    print("%-74s %d launches in %.1f us (%.1f each): clock %.3f GHz over the window, lowest 20-us stretch %.3f, idle in front %.3f" %
          (name, reps, (w1 - w0) / 100, (w1 - w0) / 100 / reps, float(ghz), lo, ghz_pre), flush=True)
