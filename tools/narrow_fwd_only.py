"""Loops wc_conv_fwd_narrow_f32 at the critic's first layer (128x32x32, 3 -> 128, 3x3) -- target of rocprofv3 --kernel-trace / --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if os.environ.get("WC_LIB"):          # development: another build of the library (tools/build_var.py)
    _lib.LIB_PATH = os.environ["WC_LIB"]
from wc_gan_amd import conv as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = torch.randn(128, 32, 32, 3, device='cuda')
w = torch.randn(128, 3, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last)
b = torch.randn(128, device='cuda')
for _ in range(n):
    y = C.narrow_forward(x, w, b)
torch.cuda.synchronize()
