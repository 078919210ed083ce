"""Convert a Keras 2.0.8 weights file of the reference (`generator.load_weights(...)`, run.py:79-83) to the flat .npz that
wc_gan_amd.checkpoint.load_keras_named() reads -- and back.  h5py is imported lazily: this image has none, the tool is meant to
run wherever an upstream `generator.h5` lives.

    python tools/h5_to_npz.py generator.h5 generator.npz        # upstream -> this build
    python tools/h5_to_npz.py generator.npz generator.h5        # this build -> upstream, for `load_weights(path, by_name=True)`

The npz -> h5 direction writes the layers in this build's module order, which is NOT Keras' `model.layers` order: the reference's
positional `generator.load_weights(path)` (run.py:79, by_name=False) would mis-assign same-shaped layers, so such a file must be
loaded with `by_name=True`.  This build's extra tensors (the right power-iteration vector `/v:0`, which upstream's SN layers do
not have) are left out of the h5 so that the per-layer weight counts match upstream's.

A Keras weights file: root attribute `layer_names`; one group per layer with attribute `weight_names` (entries such as
`Generator.0.conv1/kernel:0`) and one dataset per weight under that path.  The .npz holds the same arrays under the same
`<layer>/<weight>:0` keys, untouched (layouts are converted by load_keras_named / keras_named_state).  A file written by
`model.save(...)` keeps the same tree under `model_weights/`: handled.
"""
import sys

import numpy as np


EXTRA_SUFFIXES = ("/v:0",)      # wc_gan_amd.checkpoint.OPTIONAL_SUFFIXES: tensors upstream's layers do not hold


def h5_to_npz(src, dst):
    import h5py
    out = {}
    with h5py.File(src, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = [n.decode() if isinstance(n, bytes) else n for n in root.attrs["layer_names"]]
        for ln in names:
            g = root[ln]
            for wn in g.attrs.get("weight_names", []):
                wn = wn.decode() if isinstance(wn, bytes) else wn
                out[wn] = np.asarray(g[wn])
    np.savez(dst, **out)
    return sorted(out)


def npz_to_h5(src, dst):
    import h5py
    state = {k: v for k, v in dict(np.load(src)).items() if not k.endswith(EXTRA_SUFFIXES)}
    layers = {}
    for k in state:
        layers.setdefault(k.split("/")[0], []).append(k)
    with h5py.File(dst, "w") as f:
        f.attrs["layer_names"] = [n.encode() for n in layers]
        f.attrs["backend"] = b"tensorflow"
        f.attrs["keras_version"] = b"2.0.8"
        for ln, keys in layers.items():
            g = f.create_group(ln)
            g.attrs["weight_names"] = [k.encode() for k in keys]
            for k in keys:
                g.create_dataset(k, data=state[k])
    return sorted(state)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    a, b = sys.argv[1:]
    keys = h5_to_npz(a, b) if a.endswith((".h5", ".hdf5")) else npz_to_h5(a, b)
    print(f"{len(keys)} tensors: {a} -> {b}")
