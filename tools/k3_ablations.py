"""K3 as the layers run it -- apply_split_kernel<256, false, true, false>: planes in, ReLU + one-bit mask, 128 x 32 x 32 x 256 -- taken apart by
ablation: every library under csrc/build/var/lib_k3*.so (tools/build_var.py wc_split k3base= k3abl<bits>=-DWC_SPLIT_ABL=<bits> ...;
bits: 1 no stores, 2 no MFMA, 8 no table loads, 16 no mask words) timed ONE launch at a time behind a register-only
spin (bench.py's time_isolated rule), next to the hand-written stream copy of the same 268 MB timed by the same rule in the same process.
Ablated builds compute wrong results: only their times mean anything."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "as shipped", 1: "no stores", 2: "no MFMA", 8: "no table loads (zero table)", 16: "no mask words",
         3: "no stores, no MFMA", 10: "no MFMA, no table", 26: "loads + stores only (no MFMA / table / mask; the epilogue's scale-and-add stays)",
         27: "loads only (and the hand-off counters)", 18: "no MFMA, no mask", 24: "no table, no mask",
         64: "NOT an ablation: tile 3 requested in the prologue (-DWC_SPLIT_PRE3=1)",
         128: "table walked from a per-workgroup k-step (WRONG results)", 130: "no MFMA, table walked from a per-workgroup k-step"}
child = r'''
import sys, torch
sys.path.insert(0, %r)
from wc_gan_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from wc_gan_amd import ops
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device='cpu'); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()                      # SURVEY section 8d kernel-bench input
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
y = torch.empty_like(x); y2 = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
xs = ops.split(x)
A, At, plan = ops.color(W, gamma, xs.scale)
be = ops.split_bias(A, b, xs, mu)
mk = torch.empty(M // 32, C, dtype=torch.int32, device='cuda')
ws = ops.apply_split_workspace(C, 1, x.device)
def isolated(fn, n=25):
    for _ in range(5): fn()
    ts = []
    for rep in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400000); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]
ops.TRACE = []
ops.apply_split(xs, None, A, be, None, plan=plan, out=y, folded=True, relu=True, want_mask=True, _mask_out=mk, ws=ws)
raw = ops.TRACE[0][2]; ops.TRACE = None                             # the raw C-ABI call of the layers' launch
k = isolated(raw); c = isolated(lambda: ops.stream_copy(x, y2))
def b2b(fn, n=20):           # the same launch n times back to back as ONE hipGraph (bench.py's back_to_back_us), median of 3 replays
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(n): fn()
    g_.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g_.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[1]
kb, cb = b2b(raw), b2b(lambda: ops.stream_copy(x, y2))
print("K3 min %%5.1f median %%5.1f us | stream copy (same rule, same process) min %%5.1f median %%5.1f us | K3 / copy %%.3f | back to back in a graph: K3 %%5.1f copy %%5.1f" %% (k[0], k[1], c[0], c[1], k[1] / c[1], kb, cb))
''' % ROOT
libs = []
for lib in glob.glob(os.path.join(ROOT, "wc_gan_amd", "csrc", "build", "var", "lib_k3*.so")):
    tag = os.path.basename(lib)[len("lib_k3"):-3]
    trot = 128 if tag.endswith("trot") else 0           # (round 6: -DWC_SPLIT_TROT=1, the table walked from a different k-step by every workgroup)
    tag = tag[:-4] if trot else tag
    libs.append((trot + (0 if tag in ("base", "") else 64 if tag == "pre3" else int(tag.replace("abl", ""))), lib))
for bits, lib in sorted(libs):
    try:
        r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True, timeout=150)
        out = [l for l in r.stdout.splitlines() if l.startswith("K3")]
        line = out[0] if out else "FAILED " + r.stderr.strip()[-300:]
    except subprocess.TimeoutExpired:
        line = "TIMED OUT (a hang: stop here)"
        print(f"ABL {bits:2d}  {NAMES.get(bits, '?'):80s} {line}", flush=True)
        break
    print(f"ABL {bits:2d}  {NAMES.get(bits, '?'):80s} {line}", flush=True)
