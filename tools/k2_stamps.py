"""Per-phase cycle counts of cholesky_fused_kernel from a -DCF_STAMPS=1 build (development).
usage (GPU box): WC_EXTRA_FLAGS=-DCF_STAMPS=1 python -m wc_gan_amd.build --force && python tools/k2_stamps.py"""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib, ops
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 16384
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(M, C, generator=g).cuda()
s, xtx = ops.stats(x)
mu = torch.empty(C, device='cuda'); L = torch.empty(C, C, dtype=torch.float64, device='cuda'); W = torch.empty_like(L)
ws = torch.zeros(max(lib.wc_factor_workspace_bytes(C, 1), 8 * (8192 + 512)), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.wc_factor_f64(s.data_ptr(), xtx.data_ptr(), M, C, 1, 1e-3, 0.99, 1, 1, None, None, mu.data_ptr(), None,
                                 L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel(), None), "factor")
torch.cuda.synchronize()
st = ws.view(torch.int64)[8192:8192 + 384].cpu().numpy().reshape(3, 128)
for name, row in zip(("wave 0 (factor)", "wave 1 (solver)", "wave 5 (owner)"), st):
    n = int(row[127]); t = row[:n].astype(np.int64)
    d = np.diff(t)
    print(name, "stamps", n, "total cycles", int(t[-1] - t[0]))
    # stamps alternate: before barrier, after barrier; so d[0::2] = wait inside a barrier, d[1::2] = work between barriers
    print("  barrier waits:", d[0::2][:60].tolist())
    print("  work segments:", d[1::2][:60].tolist())
