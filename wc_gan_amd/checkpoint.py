"""Keras-named weight interchange for the WHOLE generator.

The reference saves / loads Keras HDF5 weights by LAYER NAME (`run.py:79-83`; names built at `generator.py:85-86, 36-38,
55-58, 127, 145, 154-156`).  h5py is not available in this image, so the container here is a flat `.npz` with exactly the
dataset paths a Keras 2.0.8 file holds -- `<layer_name>/<weight_name>:0`, arrays in KERAS' layouts -- and
`tools/h5_to_npz.py` (h5py imported lazily, runs wherever h5py exists) converts an upstream `generator.h5` to that `.npz` and
back.  Weight names / shapes of the un-vendored layers and the block's layer names are [UPSTREAM-RECALL] (the `gan/`
submodule is empty); everything in-tree is cited.

    dense_1/kernel:0                       (128, 4*4*C)   Dense, generator.py:127 (unnamed: Keras' automatic name); /bias:0
    Generator.<i>.conv1|conv2/kernel:0     (3, 3, Cin, Cout)  Conv2D of resblock `Generator.<i>` (generator.py:145); /bias:0
    Generator.<i>.shortcut/kernel:0        (1, 1, Cin, Cout)
    Generator.Final/kernel:0               (3, 3, C, 3)   generator.py:154-155; /bias:0
    ... /u:0 (1, Cout)                     spectrally normalised variants (SNConv2D / SNDense, generator.py:104-113); this build
                                           also keeps the right vector under /v:0 (absent upstream: rebuilt on load when missing)
    Generator.<i>.bn1_npart/moving_mean:0  (C, 1)         DecorelationNormalization (generator.py:24, 85-86); /moving_cov:0 (C, C)
    Generator.<i>.bn1_repart/kernel:0      (1, 1, C, C)   Conv2D 1x1 (uconv, generator.py:49-51)           + /bias:0 (C,)
    ..._repart_c/kernel:0                  (K, C, C)      ConditionalConv11 (generator.py:52-60)          + /bias:0 (K, C)
    ..._repart_c/kernel:0, /class_matrix:0                FactorizedConv11 (E, C, C), (K, E) (generator.py:69-78)
    ..._repart[_u|_c]/gamma:0, /beta:0                    CenterScale / ConditionalCenterScale (generator.py:28-40)
    Generator.BN.Final_npart/..., Generator.BN.Final_repart/...   the last site (generator.py:154)

Layout conversions (torch here <-> Keras in the file): Conv2D kernel (Cout, Cin, kh, kw) <-> (kh, kw, Cin, Cout); Dense kernel
(out, in) <-> (in, out); everything else is stored as it is held.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .layers import _Coloring, DecorelationNormalization


def _dense_name(index, spectral):
    return f"{'sn_dense' if spectral else 'dense'}_{index}"


def _entries(module):
    """(key, tensor, to_keras, from_keras) for every tensor of the generator that a Keras checkpoint names."""
    from .generator import Conv2D
    ident = (lambda a: a, lambda a: a)
    conv = (lambda a: np.transpose(a, (2, 3, 1, 0)), lambda a: np.transpose(a, (3, 2, 0, 1)))
    dense = (lambda a: a.T, lambda a: a.T)
    n_dense = 0
    for m in module.modules():
        name = getattr(m, 'layer_name', None)
        if isinstance(m, (DecorelationNormalization, _Coloring)) and name is not None:
            for wn, t in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
                if not wn.startswith('_'):
                    yield (f"{name}/{wn}:0", t) + ident
        elif isinstance(m, Conv2D) and name is not None:
            c = m.conv
            yield (f"{name}/kernel:0", c.weight) + conv
            if c.bias is not None:
                yield (f"{name}/bias:0", c.bias) + ident
            if hasattr(c, 'sn_u'):
                yield (f"{name}/u:0", c.sn_u, lambda a: a.reshape(1, -1), lambda a: a.reshape(-1))
                yield (f"{name}/v:0", c.sn_v, lambda a: a.reshape(1, -1), lambda a: a.reshape(-1))
        elif isinstance(m, nn.Linear):
            n_dense += 1
            sn = hasattr(m, 'sn_u')
            name = _dense_name(n_dense, sn)
            yield (f"{name}/kernel:0", m.weight) + dense
            if m.bias is not None:
                yield (f"{name}/bias:0", m.bias) + ident
            if sn:
                yield (f"{name}/u:0", m.sn_u, lambda a: a.reshape(1, -1), lambda a: a.reshape(-1))
                yield (f"{name}/v:0", m.sn_v, lambda a: a.reshape(1, -1), lambda a: a.reshape(-1))


def keras_named_state(module):
    """{'<layer_name>/<weight>:0': ndarray in Keras' layout} for every tensor of `module` (the generator)."""
    return {k: np.ascontiguousarray(to_k(t.detach().cpu().numpy())).copy() for k, t, to_k, _ in _entries(module)}


def keras_key_list(module):
    """[(key, Keras shape)]: the documented key list of INTEGRATION.md."""
    return [(k, tuple(v.shape)) for k, v in keras_named_state(module).items()]


def save_keras_named(module, path):
    np.savez(path, **keras_named_state(module))


OPTIONAL_SUFFIXES = ("/v:0",)        # this build's additions: an upstream file does not hold them


def load_keras_named(module, state, strict=True):
    """Copy arrays from a name -> ndarray mapping (or an .npz path) into the matching tensors of `module`.
    strict: every tensor of the module must be in the state (except OPTIONAL_SUFFIXES) and the state may hold nothing else."""
    if isinstance(state, str):
        state = dict(np.load(state))
    seen = set()
    for k, t, _, from_k in _entries(module):
        if k in state:
            a = torch.as_tensor(np.ascontiguousarray(from_k(np.asarray(state[k]))), dtype=t.dtype)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"{k}: shape {tuple(np.asarray(state[k]).shape)} does not fit the layer's {tuple(t.shape)}")
            with torch.no_grad():
                t.copy_(a.to(t.device))          # (copy_ keeps the tensor's own strides: channels_last kernels stay channels_last)
            seen.add(k)
        elif strict and not k.endswith(OPTIONAL_SUFFIXES):
            raise KeyError(f"missing weight {k}")
    extra = set(state) - seen
    if strict and extra:
        raise KeyError(f"unexpected weights {sorted(extra)[:4]}...")
    return sorted(seen)
