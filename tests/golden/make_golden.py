"""Regenerates tests/golden/wc_golden.npz from the float64 oracle (oracle/wc_oracle.py).

The reference holds no golden vectors for this path and cannot be imported here,
so these fixtures pin the BUILD's semantics -- parity with upstream itself stays unpinned.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wc_oracle as o  # noqa: E402

CASES = [  # name, shape (N,H,W,C), Kc, conditioning, training
    ("tiny_uncond", (4, 4, 4, 32), 1, "ill", True),
    ("odd_uncond", (3, 5, 7, 32), 1, "well", True),
    ("cond_k4", (8, 3, 3, 64), 4, "ill", True),
    ("whiten_wide", (6, 3, 3, 96), 1, "well", True),
    ("cond_k10", (12, 2, 2, 32), 10, "well", True),
    ("eval_cond", (4, 3, 3, 64), 3, "ill", False),
]


def main():
    out = {}
    rng = np.random.default_rng(20190506)
    for name, shape, Kc, cond, training in CASES:
        N, C = shape[0], shape[-1]
        x = o.synth_activation(rng, shape, cond).astype(np.float32)
        G, B = o.synth_coloring(rng, C, Kc)
        G = G.astype(np.float32); B = B.astype(np.float32)
        slot = rng.integers(0, Kc, N).astype(np.int32)
        gy = rng.standard_normal(shape).astype(np.float32)
        if training:
            mm0, mc0 = np.zeros(C, np.float32), np.eye(C, dtype=np.float32)
        else:
            mm0, mc0 = o.moments_to_stats(*o.batch_moments(o.synth_activation(rng, (16 * C, C), cond)))
            mm0, mc0 = mm0.astype(np.float32), mc0.astype(np.float32)
        y, cache = o.wc_forward(x, G, B, slot, training=training, moving_mean=mm0, moving_cov=mc0)
        dx, dG, dB = o.wc_backward(gy, cache)
        d = dict(x=x, gamma=G, beta=B, slot=slot, gy=gy, mm0=mm0, mc0=mc0, training=np.array(training),
                 y=y.astype(np.float32), dx=dx.astype(np.float32), dgamma=dG.astype(np.float32), dbeta=dB.astype(np.float32),
                 mu=cache['mu'].astype(np.float32), W=cache['W'].astype(np.float32))
        if training:
            d.update(mm1=cache['moving_mean'].astype(np.float32), mc1=cache['moving_cov'].astype(np.float32))
        for k, v in d.items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wc_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
