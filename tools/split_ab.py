"""Development: in-process A/B of wc_apply_split_f16x2 over every library under csrc/build/var/ (same tensors, same box,
same thermal state; the variants alternate round by round).  usage: split_ab.py [rounds] [N H C]"""
import ctypes, glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wc_gan_amd import _lib, ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, H, C = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (128, 32, 256)
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
if os.environ.get("ILL"):      # the SURVEY section 8d kernel-bench input (cond ~ 1e6), as bench.py's roofline uses it
    z = torch.randn(M, C, generator=g)
    mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
    x = (z @ mix + 0.2).view(N, H, H, C).cuda(); gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda()
else:
    x = torch.randn(N, H, H, C, generator=g).cuda(); gamma = (torch.randn(1, C, C, generator=g) / 16).cuda()
b = torch.zeros(1, C).cuda(); y = torch.empty_like(x); y2 = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
xs = ops.split(x)
A, At, plan = ops.color(W, gamma, xs.scale)
A1, At1, plan1 = ops.color(W, gamma, cs)
be = ops.split_bias(A, b, xs, mu)
yref = ops.apply(x, mu, A, b, None, fast=False)
ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
st = torch._C._cuda_getCurrentRawStream(0)
libs = {}
for p in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_*.so"))):
    if "STAMPS" in p or "ABL" in p: continue
    l = ctypes.CDLL(p)
    l.wc_apply_split_f16x2.restype = ctypes.c_int
    l.wc_apply_split_f16x2.argtypes = _lib.SIGNATURES["wc_apply_split_f16x2"][1]
    libs[os.path.basename(p)[4:-3]] = l
zp = torch.zeros_like(xs.planes)
def run(l, planes=None):
    return l.wc_apply_split_f16x2((planes if planes is not None else xs.planes).data_ptr(), None, xs.scale.data_ptr(), None, A.data_ptr(), be.data_ptr(), None, N, H * H, C, 1, 0,
                                  y.data_ptr(), plan.data_ptr(), ws.data_ptr(), ws.numel(), st)
def timed(fn, it=20):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, l in libs.items():
    assert run(l) == 0
    torch.cuda.synchronize()
    print(name, "err %.2e" % float((y - yref).abs().max() / yref.abs().max()))
for _ in range(20): timed(lambda: ops.stream_copy(x, y2))          # clocks up
res = {k: [] for k in list(libs) + ["copy", "k3_fp32", "base_zero_input"]}
for r in range(rounds):
    for name, l in libs.items(): res[name].append(timed(lambda: run(l)))
    res["copy"].append(timed(lambda: ops.stream_copy(x, y2)))
    res["base_zero_input"].append(timed(lambda: run(libs["base"], zp)))
    res["k3_fp32"].append(timed(lambda: ops.apply(x, mu, A1, b, None, out=y2, plan=plan1)))
# the same kernels launched ONE AT A TIME behind another kernel (a stream copy of the same tensors), events around the single
# launch: what a launch costs inside a layer's flow, without the overlap of one launch's tail with the next one's head
def single(fn, n=30):
    out = []
    for _ in range(n):
        ops.stream_copy(x, y2)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    out.sort(); return out
print("single launches behind a copy (min / median / max us):")
for name, l in libs.items():
    v = single(lambda: run(l)); print("  %-38s %.1f %.1f %.1f" % (name, v[0], v[len(v) // 2], v[-1]))
v = single(lambda: ops.apply(x, mu, A1, b, None, out=y2, plan=plan1)); print("  %-38s %.1f %.1f %.1f" % ("k3_fp32", v[0], v[len(v) // 2], v[-1]))
v = single(lambda: ops.stream_copy(x, y)); print("  %-38s %.1f %.1f %.1f" % ("copy", v[0], v[len(v) // 2], v[-1]))
cp = sorted(res["copy"])[len(res["copy"]) // 2]
for k, v in res.items():
    v = sorted(v)
    print("%-40s min %.1f med %.1f max %.1f   copy/med %.3f" % (k, v[0], v[len(v) // 2], v[-1], cp / v[len(v) // 2]))
