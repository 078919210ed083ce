"""Six launches of the headline block convolution (3x3 'same' 256->256 at 128x32x32): the target of the rocprofv3 passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import conv as C
N, H, Cc = 128, 32, 256
g = torch.Generator(device="cpu"); g.manual_seed(1234)
x = torch.randn(N, H, H, Cc, generator=g).cuda()
w = (torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5).cuda().contiguous(memory_format=torch.channels_last)
(gf, kf, nf), _ = C._geoms('same', N, H, H, w)
planes, img = C.split_planes(x), C.weight_image(w, gf, kf, nf)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    y = C.run(planes, img, gf)
torch.cuda.synchronize()
