"""Per-step cycle counts of tri_inverse_cols_kernel from a -DCF_STAMPS=1 build (development): W must be followed by scratch."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib, ops
lib = _lib.load()
C = 256; M = 16384
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(M, C, generator=g).cuda()
s, xtx = ops.stats(x)
mu = torch.empty(C, device='cuda'); L = torch.empty(C, C, dtype=torch.float64, device='cuda')
Wbig = torch.zeros(C * C + 512, dtype=torch.float64, device='cuda')
ws = torch.zeros(max(lib.wc_factor_workspace_bytes(C, 1), 8 * (8192 + 512)), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.wc_factor_f64(s.data_ptr(), xtx.data_ptr(), M, C, 1, 1e-3, 0.99, 1, 1, None, None, mu.data_ptr(), None,
                                 L.data_ptr(), Wbig.data_ptr(), ws.data_ptr(), ws.numel(), None), "factor")
torch.cuda.synchronize()
st = Wbig[C * C:C * C + 256].view(torch.int64).cpu().numpy().reshape(2, 128)
for name, row in zip(("wave 0 (column 0)", "wave 4 (column 15)"), st):
    n = int(row[127]); t = row[:n].astype(np.int64)
    print(name, "stamps", n, "total", int(t[-1] - t[0]) if n else 0)
    print("  deltas:", np.diff(t).tolist())
