import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C
N, H, Ci, Co, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
x = torch.randn(N, H, H, Ci, device='cuda')
shape = (Ci, Co, 4, 4) if kind == 'up' else (Co, Ci, 4, 4) if kind == 'down' else (Co, Ci, 3, 3)
w = torch.randn(*shape, device='cuda').contiguous(memory_format=torch.channels_last)
(gf, kf, nf), _ = C._geoms(kind, N, H, H, w)
y = C.run(C.split_planes(x), C.weight_image(w, gf, kf, nf), gf)
xpl, gpl = C.split_planes(x), C.split_planes(torch.randn_like(y))
for _ in range(5): C.weight_gradient(xpl, gpl, gf, w, kf, nf)
torch.cuda.synchronize()
