"""Dumps K1's (sum, xtx) of the seed-sweep inputs (tools/seed_sweep.py) at 128x32x32x256 for the analysis of the dx error on the CPU
(tools/k1_err_structure.py): gpurun_out/k1_moments_seed<seed>.npz."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
from wc_gan_amd import ops
shape = (128, 32, 32, 256)
os.makedirs("gpurun_out", exist_ok=True)
for seed in (int(a) for a in sys.argv[1:]):
    rng = np.random.default_rng(seed)
    x = o.synth_activation(rng, shape, "ill").astype(np.float32)
    s, xtx = ops.stats(torch.from_numpy(x).cuda().view(-1, 256))
    np.savez("gpurun_out/k1_moments_seed%d.npz" % seed, s=s.cpu().numpy(), xtx=xtx.cpu().numpy())
    print("seed", seed, "done")
