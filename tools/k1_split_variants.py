"""Development: times wc_stats_split_f16x2's kernel with every library under csrc/build/var/ in one process (HIP events around
the ABI call: kernel + the two tail launches); stamp builds print where a wave's time went."""
import ctypes, glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wc_gan_amd import _lib, ops
N, H, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 32, 256)
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
x = torch.randn(N, H, H, C, generator=g).cuda(); y = torch.empty_like(x)
xs = ops.split(x)
s0, x0 = ops.stats(x.view(M, C))
lib0 = _lib.load()
ws = torch.empty(int(lib0.wc_stats_split_workspace_bytes(M, C, 1)), dtype=torch.uint8, device='cuda')
s = torch.empty(C, dtype=torch.float64, device='cuda'); xtx = torch.empty(C, C, dtype=torch.float64, device='cuda')
st = torch._C._cuda_getCurrentRawStream(0)
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device='cuda')
def timed(fn, it=20):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for _ in range(10): timed(lambda: ops.stream_copy(x, y))
libs = {}
for p in sorted(glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_*.so"))):
    l = ctypes.CDLL(p)
    l.wc_stats_split_f16x2.restype = ctypes.c_int
    l.wc_stats_split_f16x2.argtypes = _lib.SIGNATURES["wc_stats_split_f16x2"][1]
    libs[os.path.basename(p)[4:-3]] = l
def run(l):
    return l.wc_stats_split_f16x2(xs.planes.data_ptr(), xs.center.data_ptr(), xs.scale.data_ptr(), M, C, 1, s.data_ptr(), xtx.data_ptr(),
                                  ws.data_ptr(), ws.numel(), st)
res = {k: [] for k in libs}
for name, l in libs.items():
    if "STAMPS" in name:
        l.wc_dev_split_xtx_dbg.argtypes = [ctypes.c_void_p]; l.wc_dev_split_xtx_dbg(dbg.data_ptr())
    assert run(l) == 0
    torch.cuda.synchronize()
    print(name, "xtx err vs fp32 path %.2e" % float((xtx - x0).abs().max() / x0.abs().max()))
for r in range(7):
    for name, l in libs.items(): res[name].append(timed(lambda: run(l)))
for k, v in res.items():
    v = sorted(v); print("%-30s min %.1f med %.1f max %.1f" % (k, v[0], v[len(v) // 2], v[-1]))
d = dbg.view(256, 8, 4).double().cpu()
if d.sum() > 0:
    f = lambda t: "mean %.0f min %.0f max %.0f" % (t.mean(), t.min(), t.max())
    ty = (torch.arange(256) >> 3) % 3
    for t in range(3):
        dt = d[ty == t]; dt = dt[dt[:, 0, 3] > 0]
        for w in (0, 4):
            m = dt[:, w, :].mean(dim=0)
            print("  type %d wave %d: wait %.0f valu %.0f mfma %.0f total %.0f" % (t, w, m[0], m[1], m[2], m[3]))
    print("stamps per wave (s_memtime ticks): wait+barrier+dma", f(d[..., 0]), "| valu", f(d[..., 1]), "| reads+mfma+flush", f(d[..., 2]), "| total", f(d[..., 3]))
