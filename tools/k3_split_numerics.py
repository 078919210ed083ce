"""CPU emulation of the split-fp16 apply y = f A at the headline site: how far can the product count go down?
(a 2-product variant gated at the contract's 1e-4.)  float64 arithmetic on fp16-rounded operands."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
rng = np.random.default_rng(11)
C = 256
M = 32768                      # a quarter of the headline rows is enough for the statistics of the error
for cond in ("ill", "well"):
    x = o.synth_activation(rng, (M, 1, 1, C), cond).astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1)
    y_ref, cache = o.wc_forward(x, G, B)
    f = cache['f']; A = cache['A'][0]
    sig = np.sqrt(np.diag(cache['sigma']))
    s = 2.0 ** (3 - np.floor(np.log2(sig)) - 1)                  # chan_scale: std*s in [4, 8)
    g = (f * s).astype(np.float32)
    Bp = A / s[:, None]
    cs = 2.0 ** np.ceil(np.log2(np.abs(Bp).max(axis=0)))        # colscale: |B'/cs| <= 1
    Bp = (Bp / cs).astype(np.float32)
    gh = g.astype(np.float16); gl = (g - gh.astype(np.float32)).astype(np.float16)
    Bh = Bp.astype(np.float16); Bl = (Bp - Bh.astype(np.float32)).astype(np.float16)
    d = lambda a: a.astype(np.float64)
    yr = y_ref.reshape(M, C) - B[0]
    rel = lambda y: float(np.abs(y * cs - yr).max() / np.abs(y_ref).max())
    print(cond, "cond(T) %.1e" % np.linalg.cond((1 - 1e-3) * cache['sigma'] + 1e-3 * np.eye(C)))
    print("  3 products  hi*Hi + lo*Hi + hi*Lo : %.2e" % rel(d(gh) @ d(Bh) + d(gl) @ d(Bh) + d(gh) @ d(Bl)))
    print("  2 products  (hi + lo)*Hi           : %.2e   (table rounded to fp16)" % rel((d(gh) + d(gl)) @ d(Bh)))
    print("  2 products  hi*(Hi + Lo)           : %.2e   (activation rounded to fp16)" % rel(d(gh) @ (d(Bh) + d(Bl))))
    print("  1 product   hi*Hi                  : %.2e" % rel(d(gh) @ d(Bh)))
    # bf16 x 3 for comparison
    def bf(a):
        u = a.astype(np.float32).view(np.uint32); u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32)
    gbh = bf(g); gbl = bf(g - gbh); Bbh = bf(Bp); Bbl = bf(Bp - Bbh)
    print("  bf16 x 3                           : %.2e" % rel(d(gbh) @ d(Bbh) + d(gbl) @ d(Bbh) + d(gbh) @ d(Bbl)))
