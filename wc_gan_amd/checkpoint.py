"""Keras-named weight interchange for the WC sites.

The reference saves/loads Keras HDF5 weights by LAYER NAME (`run.py:79-83`; names built at `generator.py:85-86,
36-38, 55-58, 145, 154`).  h5py is not available in this environment, so the container here is a flat `.npz` with
exactly the dataset paths a Keras 2.0.8 file would hold -- `<layer_name>/<weight_name>:0` -- which a one-line h5py
loop converts either way once h5py is at hand.  Weight names/shapes of the un-vendored layers are [UPSTREAM-RECALL].

    Generator.0.bn1_npart/moving_mean:0   (C, 1)        DecorelationNormalization
    Generator.0.bn1_npart/moving_cov:0    (C, C)
    Generator.0.bn1_repart/kernel:0       (1, 1, C, C)  Conv2D 1x1 (uconv)            + /bias:0 (C,)
    ..._repart_c/kernel:0                 (K, C, C)     ConditionalConv11             + /bias:0 (K, C)
    ..._repart_c/kernel:0, /class_matrix:0              FactorizedConv11 (E,C,C), (K,E)
    ..._repart[_u|_c]/gamma:0, /beta:0                  CenterScale / ConditionalCenterScale
"""
from __future__ import annotations

import numpy as np
import torch

from .layers import _Coloring, DecorelationNormalization


def _named_tensors(module):
    for m in module.modules():
        name = getattr(m, 'layer_name', None)
        if name is None or not isinstance(m, (DecorelationNormalization, _Coloring)):
            continue
        for wn, t in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
            if wn.startswith('_'):
                continue
            yield f"{name}/{wn}:0", t


def keras_named_state(module):
    """{'<layer_name>/<weight>:0': ndarray} for every WC-site layer of `module` (e.g. the generator)."""
    return {k: t.detach().cpu().numpy().copy() for k, t in _named_tensors(module)}


def save_keras_named(module, path):
    np.savez(path, **keras_named_state(module))


def load_keras_named(module, state, strict=True):
    """Copy arrays from a name -> ndarray mapping (or an .npz path) into the matching WC-site tensors."""
    if isinstance(state, str):
        state = dict(np.load(state))
    seen = set()
    for k, t in _named_tensors(module):
        if k in state:
            a = torch.as_tensor(np.asarray(state[k]), dtype=t.dtype)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"{k}: shape {tuple(a.shape)} does not match {tuple(t.shape)}")
            with torch.no_grad():
                t.copy_(a.to(t.device))
            seen.add(k)
        elif strict:
            raise KeyError(f"missing weight {k}")
    extra = set(state) - seen
    if strict and extra:
        raise KeyError(f"unexpected weights {sorted(extra)[:4]}...")
    return sorted(seen)
