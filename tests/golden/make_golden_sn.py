"""Regenerates tests/golden/sn_golden.npz (spectral normalisation) from the float64 oracle.
As for wc_golden.npz these pin the BUILD's semantics; the reference holds no vectors for its SN layers.
Run from the repo root:  python tests/golden/make_golden_sn.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wc_oracle as o  # noqa: E402

CASES = [("conv3x3", 32, 9 * 16, 1), ("conv1x1", 24, 40, 2), ("dense_row", 1, 64, 1), ("embedding", 10, 48, 3), ("eval", 16, 27, 0)]


def main():
    rng = np.random.default_rng(20181126)
    out = {}
    for name, R, K, it in CASES:
        W = (rng.standard_normal((R, K)) / np.sqrt(K)).astype(np.float32)
        u = rng.standard_normal(R); u = (u / np.linalg.norm(u)).astype(np.float32)
        v = rng.standard_normal(K); v = (v / np.linalg.norm(v)).astype(np.float32)
        g = rng.standard_normal((R, K)).astype(np.float32)
        w_sn, sigma, u1, v1 = o.spectral_normalize(W, u, v, it)
        d = dict(W=W, u0=u, v0=v, g=g, iterations=np.array(it), w_sn=w_sn.astype(np.float32), sigma=np.array(sigma, np.float32),
                 u1=u1.astype(np.float32), v1=v1.astype(np.float32),
                 dW_full=o.spectral_normalize_backward(g, w_sn, u1, v1, sigma, True).astype(np.float32),
                 dW_const=o.spectral_normalize_backward(g, w_sn, u1, v1, sigma, False).astype(np.float32))
        for k, val in d.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sn_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
