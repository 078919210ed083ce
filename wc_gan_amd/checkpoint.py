"""Keras-named weight interchange for the WHOLE generator.

The reference saves / loads Keras HDF5 weights by LAYER NAME (`run.py:79-83`; names built at `generator.py:85-86, 36-38,
55-58, 127, 145, 154-156`).  h5py is not available in this image, so the container here is a flat `.npz` with exactly the
dataset paths a Keras 2.0.8 file holds -- `<layer_name>/<weight_name>:0`, arrays in KERAS' layouts -- and
`tools/h5_to_npz.py` (h5py imported lazily, runs wherever h5py exists) converts an upstream `generator.h5` to that `.npz` and
back.  Weight names / shapes of the un-vendored layers and the block's layer names are [UPSTREAM-RECALL] (the `gan/`
submodule is empty); everything in-tree is cited.

    dense_1/kernel:0                       (128, 4*4*C)   Dense, generator.py:127 (unnamed: Keras' automatic name; any dense_N loads); /bias:0
    embedding_1/embeddings:0               (K, C)         Embedding of concat_cls generators (generator.py:120-121, run.py:175)
    <name>_npart/moving_mean:0, /moving_variance:0        BatchNormalization(center=False, scale=False) of norm == 'b' (generator.py:22)
    Generator.<i>.conv1|conv2/kernel:0     (3, 3, Cin, Cout)  Conv2D of resblock `Generator.<i>` (generator.py:145); /bias:0
    Generator.<i>.shortcut/kernel:0        (1, 1, Cin, Cout)
    Generator.Final/kernel:0               (3, 3, C, 3)   generator.py:154-155; /bias:0
    ... /u:0 (1, Cout)                     spectrally normalised variants (SNConv2D / SNDense, generator.py:104-113); this build
                                           also keeps the right vector under /v:0 (absent upstream: load_keras_named rebuilds it
                                           as normalize(W^T u) when the file has none)
    Generator.<i>.bn1_npart/moving_mean:0  (C, 1)         DecorelationNormalization (generator.py:24, 85-86); /moving_cov:0 (C, C)
    Generator.<i>.bn1_repart/kernel:0      (1, 1, C, C)   Conv2D 1x1 (uconv, generator.py:49-51)           + /bias:0 (C,)
    ..._repart_c/kernel:0                  (K, C, C)      ConditionalConv11 (generator.py:52-60)          + /bias:0 (K, C)
    ..._repart_c/kernel:0, /class_matrix:0                FactorizedConv11 (E, C, C), (K, E) (generator.py:69-78)
    ..._repart[_u|_c]/gamma:0, /beta:0                    CenterScale / ConditionalCenterScale (generator.py:28-40)
    Generator.BN.Final_npart/..., Generator.BN.Final_repart/...   the last site (generator.py:154)

Two conventions that differ and are NOT converted (no shipped recipe uses either; stated so that nobody is surprised): Keras'
BatchNormalization stores the BIASED batch variance in moving_variance, torch's running_var the unbiased one (n / (n - 1) apart while
training resumes: norm == 'b' generators only); and a spectral concat_cls generator upstream wraps its Embedding in SNEmbeding with a
`/u:0` weight, where this build's Generator keeps a plain nn.Embedding (the key is then `embedding_<n>`, not `sn_embeding_<n>`).

Layout conversions (torch here <-> Keras in the file): Conv2D kernel (Cout, Cin, kh, kw) <-> (kh, kw, Cin, Cout); Dense kernel
(out, in) <-> (in, out); everything else is stored as it is held.
"""
from __future__ import annotations

import re

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .layers import _Coloring, DecorelationNormalization


def _dense_name(index, spectral):
    return f"{'sn_dense' if spectral else 'dense'}_{index}"


def _sn_pair(name, m):
    as_row = (lambda a: a.reshape(1, -1), lambda a: a.reshape(-1))
    return [(f"{name}/u:0", m.sn_u) + as_row, (f"{name}/v:0", m.sn_v) + as_row]


def _entries(module):
    """(key, tensor, to_keras, from_keras) for every tensor of the generator that a Keras checkpoint names.  Unnamed Keras layers
    (Dense generator.py:127, Embedding generator.py:120-121 with concat_cls, run.py:175) carry Keras' automatic names
    `dense_<n>` / `embedding_<n>`, n = the order of construction."""
    from .generator import Conv2D, _BatchNormNoAffine
    ident = (lambda a: a, lambda a: a)
    conv = (lambda a: np.transpose(a, (2, 3, 1, 0)), lambda a: np.transpose(a, (3, 2, 0, 1)))
    dense = (lambda a: a.T, lambda a: a.T)
    n_dense = n_emb = 0
    for m in module.modules():
        name = getattr(m, 'layer_name', None)
        if isinstance(m, (DecorelationNormalization, _Coloring)) and name is not None:
            for wn, t in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
                if not wn.startswith('_'):
                    yield (f"{name}/{wn}:0", t) + ident
        elif isinstance(m, _BatchNormNoAffine) and name is not None:
            # BatchNormalization(center=False, scale=False) (generator.py:22): Keras keeps moving_mean / moving_variance only
            if m.bn is not None:
                yield (f"{name}/moving_mean:0", m.bn.running_mean) + ident
                yield (f"{name}/moving_variance:0", m.bn.running_var) + ident
        elif isinstance(m, Conv2D) and name is not None:
            c = m.conv
            yield (f"{name}/kernel:0", c.weight) + conv
            if c.bias is not None:
                yield (f"{name}/bias:0", c.bias) + ident
            if hasattr(c, 'sn_u'):
                yield from _sn_pair(name, c)
        elif isinstance(m, nn.Linear):
            n_dense += 1
            sn = hasattr(m, 'sn_u')
            name = _dense_name(n_dense, sn)
            yield (f"{name}/kernel:0", m.weight) + dense
            if m.bias is not None:
                yield (f"{name}/bias:0", m.bias) + ident
            if sn:
                yield from _sn_pair(name, m)
        elif isinstance(m, nn.Embedding):
            n_emb += 1
            sn = hasattr(m, 'sn_u')
            # upstream's class is SNEmbeding (one d: generator.py:111), which Keras snake-cases to `sn_embeding_<n>`
            name = f"{'sn_embeding' if sn else 'embedding'}_{n_emb}"
            yield (f"{name}/embeddings:0", m.weight) + ident
            if sn:
                yield from _sn_pair(name, m)


# tensors a module may hold that are NOT weights of the Keras model: scratch of this build (names with a leading underscore are
# skipped as well) and torch's step counter of BatchNorm
_NOT_WEIGHTS = ('num_batches_tracked',)


def _assert_covered(module):
    """Every parameter and persistent buffer of `module` is named by _entries -- a tensor the walk does not know would otherwise be
    lost silently by save + load (strict=True cannot notice what is never enumerated)."""
    # by object identity, not by data_ptr(): zero-element / unmaterialised tensors all report data_ptr() == 0 and a view shares its
    # base's pointer, so either could pass for a tensor the walk does not name
    covered = {id(t) for _, t, _, _ in _entries(module)}
    persistent = set(module.state_dict(keep_vars=True).keys())
    missing = []
    for n, t in list(module.named_parameters()) + list(module.named_buffers()):
        leaf = n.rsplit('.', 1)[-1]
        if leaf.startswith('_') or leaf in _NOT_WEIGHTS or n not in persistent or t.numel() == 0:
            continue
        if id(t) not in covered:
            missing.append(n)
    if missing:
        raise NotImplementedError("the Keras-named checkpoint has no entry for " + ", ".join(sorted(missing)[:6]) +
                                  (" ..." if len(missing) > 6 else "") + " (wc_gan_amd/checkpoint.py:_entries)")


def keras_named_state(module):
    """{'<layer_name>/<weight>:0': ndarray in Keras' layout} for every tensor of `module` (the generator); raises when the module
    holds a weight that no entry covers."""
    _assert_covered(module)
    return {k: np.ascontiguousarray(to_k(t.detach().cpu().numpy())).copy() for k, t, to_k, _ in _entries(module)}


def keras_key_list(module):
    """[(key, Keras shape)]: the documented key list of INTEGRATION.md."""
    return [(k, tuple(v.shape)) for k, v in keras_named_state(module).items()]


def save_keras_named(module, path):
    np.savez(path, **keras_named_state(module))


OPTIONAL_SUFFIXES = ("/v:0",)        # this build's additions: an upstream file does not hold them


_AUTO_NAME = re.compile(r"^(dense|sn_dense|embedding|sn_embeding|sn_embedding)_(\d+)/(.+)$")     # (sn_embedding: files written by rounds 3-4 of this build)


_LEGACY_KINDS = {'sn_embedding': 'sn_embeding'}


def _renumber_auto_names(module, state):
    """Keras numbers unnamed layers with a process-global counter (`dense_1`, `dense_7` ... depending on what was built before the
    generator), so a file's `dense_N` need not be this build's `dense_1`: the n-th distinct index of a kind in the file (ascending)
    is taken for the n-th such layer of the module."""
    # the kind as THIS build spells it: rounds 3-4 wrote `sn_embedding_N`, upstream's class SNEmbeding snake-cases to `sn_embeding_N`
    # (the legacy spelling matched _AUTO_NAME but was never rewritten, so such a file failed with `missing weight` or, with
    # strict=False, loaded nothing)
    canon = lambda kind: _LEGACY_KINDS.get(kind, kind)
    want = {}
    for k, _, _, _ in _entries(module):
        m = _AUTO_NAME.match(k)
        if m:
            want.setdefault(canon(m.group(1)), set()).add(int(m.group(2)))
    have = {}
    for k in state:
        m = _AUTO_NAME.match(k)
        if m:
            have.setdefault(canon(m.group(1)), set()).add(int(m.group(2)))
    out = {}
    for k, v in state.items():
        m = _AUTO_NAME.match(k)
        if m:
            kind, idx = canon(m.group(1)), int(m.group(2))
            if kind in want and sorted(have[kind]) != sorted(want[kind]) and len(have[kind]) == len(want[kind]):
                idx = sorted(want[kind])[sorted(have[kind]).index(idx)]
            k = f"{kind}_{idx}/{m.group(3)}"
        out[k] = v
    return out


def _rebuild_sn_v(owner):
    """An upstream file holds u only (SNConv2D / SNDense keep one vector, generator.py:104-113): v = normalize(W^T u), the
    vector the next power-iteration step would compute -- in evaluation mode the op runs no iteration and takes sigma = u^T W v
    from the stored pair (include/wc_hip.h, wc_spectral_norm_f32), so a stale v would give a wrong sigma."""
    with torch.no_grad():
        wm = owner._as_matrix(owner.weight.detach())
        owner.sn_v.copy_(F.normalize(wm.t().mv(owner.sn_u), dim=0, eps=owner.sn_eps))


def load_keras_named(module, state, strict=True):
    """Copy arrays from a name -> ndarray mapping (or an .npz path) into the matching tensors of `module`.
    strict: every tensor of the module must be in the state (except OPTIONAL_SUFFIXES) and the state may hold nothing else.
    `dense_N` / `embedding_N` of the file may carry any N (Keras' automatic names).  A spectrally normalised layer whose
    `/v:0` is absent gets v rebuilt from the loaded weight and u."""
    if isinstance(state, str):
        state = dict(np.load(state))
    _assert_covered(module)
    state = _renumber_auto_names(module, state)
    owners = {id(m.sn_v): m for m in module.modules() if hasattr(m, 'sn_v')}
    seen = set()
    rebuild = []
    for k, t, _, from_k in _entries(module):
        if k in state:
            a = torch.as_tensor(np.ascontiguousarray(from_k(np.asarray(state[k]))), dtype=t.dtype)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"{k}: shape {tuple(np.asarray(state[k]).shape)} does not fit the layer's {tuple(t.shape)}")
            with torch.no_grad():
                t.copy_(a.to(t.device))          # (copy_ keeps the tensor's own strides: channels_last kernels stay channels_last)
            seen.add(k)
        elif k.endswith('/v:0') and id(t) in owners:
            rebuild.append(owners[id(t)])
        elif strict and not k.endswith(OPTIONAL_SUFFIXES):
            raise KeyError(f"missing weight {k}")
    for m in rebuild:                            # (after the loop: the layer's weight and u are loaded by now)
        _rebuild_sn_v(m)
    extra = set(state) - seen
    if strict and extra:
        raise KeyError(f"unexpected weights {sorted(extra)[:4]}...")
    return sorted(seen)
