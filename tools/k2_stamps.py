"""Per-phase cycle counts of cholesky_fused_kernel from a -DCF_STAMPS=1 build (development).
usage (GPU box): WC_EXTRA_FLAGS=-DCF_STAMPS=1 python -m wc_gan_amd.build --force && python tools/k2_stamps.py"""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib, ops
if len(sys.argv) > 2: _lib.LIB_PATH = sys.argv[2]      # a -DCF_STAMPS=1 library built with tools/build_var.py wc_small stamps=-DCF_STAMPS=1
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 16384
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(M, C, generator=g).cuda()
s, xtx = ops.stats(x)
mu = torch.empty(C, device='cuda'); L = torch.empty(C, C, dtype=torch.float64, device='cuda'); W = torch.empty_like(L)
ws = torch.zeros(max(lib.wc_factor_workspace_bytes(C, 1), 8 * (8192 + 16 * 128 + 64 * 64)), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.wc_factor_f64(s.data_ptr(), xtx.data_ptr(), M, C, 1, 1e-3, 0.99, 1, 1, None, None, mu.data_ptr(), None,
                                 L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel(), None), "factor")
torch.cuda.synchronize()
st = ws.view(torch.int64)[8192:8192 + 16 * 128].cpu().numpy().reshape(16, 128)
t0 = min(int(r[0]) for r in st if int(r[127]) > 0)
# every wave stamps before and after each of its barriers / counter waits: per wave the list (arrive, leave) relative to the first stamp
for w, row in enumerate(st):
    n = int(row[127]); t = (row[:n].astype(np.int64) - t0)
    print("wave %2d (SIMD %d): " % (w, w & 3) + " ".join("%d>%d" % (t[k], t[k + 1]) for k in range(0, min(n - 1, 100), 2)))

# wave 0's phases per step j >= 1 (slots 40 + 5 j + {0: next diagonal block solved + updated, in LDS; 1: its 16 columns loaded;
# 2: factored + inverted; 3: stores issued}), relative to the release of barrier (A) of step j - 1
row = st[0]; n = int(row[127]); t = row[:n].astype(np.int64) - t0
for j in range(1, min(C // 16, 16)):
    if 2 * (j - 1) + 1 >= n: break
    rel = int(t[2 * (j - 1) + 1]); ph = [int(row[40 + 5 * j + q]) - t0 - rel for q in range(4)]
    arrive = int(t[2 * j]) - rel if 2 * j < n else -1
    print("step %2d: update done +%d, columns loaded +%d, leaf done +%d, stores issued +%d, at barrier +%d" % (j, ph[0], ph[1], ph[2], ph[3], arrive))

# the helpers (cholesky_phased_kernel): per panel round of a helper wave [poll starts, poll matched, operands loaded], then stores landed,
# barrier passed, done counted -- on wave 0's clock; and wave 0's takeover: poll matched, block loaded
hs = ws.view(torch.int64)[8192 + 16 * 128: 8192 + 16 * 128 + 64 * 64].cpu().numpy().reshape(64, 64)
split = C // 16 - 10 if C // 16 >= 14 else C // 16
if split < C // 16 and hs.any():
    for hw in (0, 1, 17, 35, 54):
        r = hs[hw]
        if not r[60]: continue
        print("helper wave %2d: " % hw + " ".join("[%d %d %d]" % tuple(int(r[4 * j + q]) - t0 for q in range(3)) for j in range(split - 2)) +
              "  stores landed %d, barrier %d, counted %d (was %d)" % (int(r[60]) - t0, int(r[61]) - t0, int(r[62]) - t0 if r[62] else -1, int(r[63])))
    for wg in range(4):
        rows = [(w, hs[16 * wg + w]) for w in range(16) if hs[16 * wg + w][60]]
        if not rows: continue
        jl = split - 3
        base = min(int(r[4 * jl + 1]) for _, r in rows if r[4 * jl + 1])
        print("helper workgroup %d, last panel, cycles after its first wave saw the flag: " % wg +
              " ".join("w%d[%s saw %d, loaded %d, landed %d, barrier %d]" % (w, "blk" if r[4 * jl + 1] else "---", int(r[4 * jl + 1]) - base if r[4 * jl + 1] else -1,
                       int(r[4 * jl + 2]) - base if r[4 * jl + 2] else -1, int(r[60]) - base, int(r[61]) - base) for w, r in rows))
    print("wave 0 takeover: done seen %d, block loaded %d" % (int(st[0][124]) - t0, int(st[0][125]) - t0))

if split < C // 16:
    for w in (1, 4, 13, 15):
        r = st[w]
        print("owner wave %2d at the takeover: done seen %d, blocks loaded %d, saved panel applied %d" % (w, int(r[100]) - t0, int(r[101]) - t0, int(r[102]) - t0))
