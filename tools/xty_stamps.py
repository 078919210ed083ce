"""Per-stage cycle counts of xty_f16x3_kernel<256,false> (wave 0 of workgroup 0) from a -DXTY_STAMPS=1 build (development)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib, ops
lib = _lib.load()
N, H, C = 128, 32, 256
M = N * H * H
x = torch.randn(M, C, device='cuda')
s = torch.empty(C, dtype=torch.float64, device='cuda'); xtx = torch.empty(C, C, dtype=torch.float64, device='cuda')
nb = lib.wc_stats_workspace_bytes(M, C, 1)
ws = torch.zeros(nb + 4096, dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.wc_stats_f32(x.data_ptr(), M, C, 1, s.data_ptr(), xtx.data_ptr(), ws.data_ptr(), ws.numel(), None), "stats")
torch.cuda.synchronize()
# P is the last carve of the workspace: its end = 256-aligned end of the buffer
w64 = ws.view(torch.int64)
for off in (nb // 8,):
    n = int(w64[off + 255])
    if 0 < n < 256:
        t = w64[off:off + n].cpu().numpy().astype(np.int64)
        d = np.diff(t)
        print("stamps", n, "total", int(t[-1] - t[0]))
        k = 5
        rows = d[: (len(d) // k) * k].reshape(-1, k)
        print("per stage [write, load-issue, mfma loop, flush, barrier]:")
        print(rows)
        break
else:
    print("no stamps found")
