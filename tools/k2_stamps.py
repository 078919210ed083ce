"""Per-phase cycle counts of K2's factorising workgroups from a -DCF_STAMPS=1 build (development).
usage (GPU box): tools/build_var.py wc_small stamps=-DCF_STAMPS=1 ; python tools/k2_stamps.py 256 wc_gan_amd/csrc/build/var/lib_stamps.so
Every wave stamps (s_memtime) before and after each of its barriers / counter waits.  The relay kernel (cholesky_phased_kernel) has two
factorising workgroups; their clocks are those of different XCDs and do not compare: each is printed on its own first stamp."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib, ops
if len(sys.argv) > 2: _lib.LIB_PATH = sys.argv[2]
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 16384
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(M, C, generator=g).cuda()
s, xtx = ops.stats(x)
mu = torch.empty(C, device='cuda'); L = torch.empty(C, C, dtype=torch.float64, device='cuda'); W = torch.empty_like(L)
ws = torch.zeros(max(lib.wc_factor_workspace_bytes(C, 1), 8 * (8192 + 64 * 128)), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.wc_factor_f64(s.data_ptr(), xtx.data_ptr(), M, C, 1, 1e-3, 0.99, 1, 1, None, None, mu.data_ptr(), None,
                                 L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel(), None), "factor")
torch.cuda.synchronize()
nb = C // 16
bounds = [0] + ([int(v) for v in os.environ["WC_K2_BOUNDS"].split(",")] if os.environ.get("WC_K2_BOUNDS") else ([nb // 4, nb * 9 // 16] if nb >= 14 else [])) + [nb]
allst = ws.view(torch.int64)[8192:8192 + 64 * 128].cpu().numpy().reshape(4, 16, 128)
for part, st in enumerate(allst):
    if not any(int(r[127]) > 0 for r in st): continue
    if part + 1 >= len(bounds): break
    jbeg, jend = bounds[part], bounds[part + 1]
    print("---- factoriser %d (steps %d .. %d)%s" % (part + 1, jbeg, jend - 1,
          ": its first %d barrier pairs are the passive steps' (panel published / staged)" % (2 * jbeg) if part else ""))
    t0 = min(int(r[0]) for r in st if int(r[127]) > 0)
    for w in (0, 1, 4, 13, 15):
        row = st[w]; n = int(row[127]); t = (row[:n].astype(np.int64) - t0)
        print("wave %2d (SIMD %d): " % (w, w & 3) + " ".join("%d>%d" % (t[k], t[k + 1]) for k in range(0, min(n - 1, 100), 2)))
    for w in (1, 5, 9, 13):
        r = st[w]
        if r[106]: print("wave %2d, trailing update of its first active step: slots entered at " % w + " ".join(str(int(r[100 + q]) - t0) for q in (4, 3, 2, 1, 0) if r[100 + q]) + ", done %d, last MFMA landed %d" % (int(r[106]) - t0, int(r[107]) - t0))
    # wave 0's phases per look-ahead (slots 40 + 5 (j - jbeg) + {0: next diagonal block solved + updated, in LDS; 1: its 16 columns loaded;
    # 2: factored + inverted; 3: LDS copies written}), relative to the release of barrier (A) of the step before
    row = st[0]; n = int(row[127]); t = row[:n].astype(np.int64) - t0
    base = 2 * 2 * jbeg          # wave 0's stamps of the passive steps come first
    for k in range(1, jend - jbeg):
        if base + 2 * (k - 1) + 1 >= n or not row[40 + 5 * k + 2]: break
        rel = int(t[base + 2 * (k - 1) + 1]); ph = [int(row[40 + 5 * k + q]) - t0 - rel for q in range(4)]
        arrive = int(t[base + 2 * k]) - rel if base + 2 * k < n else -1
        print("step %2d: update done +%d, columns loaded +%d, leaf done +%d, LDS copies written +%d, at barrier +%d" % (jbeg + k, ph[0], ph[1], ph[2], ph[3], arrive))
