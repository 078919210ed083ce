"""CPU tests of the Python host mirror: create_norm alphabets, sub-layer naming, coloring tables."""
import os

import numpy as np
import pytest
import torch

from oracle import wc_oracle as o
from wc_gan_amd.generator import AFTER_NORMS, NORMS, create_norm, make_generator
from wc_gan_amd.layers import (ConditionalConv11, Conv11, DecorelationNormalization, FactorizedConv11,
                               WhiteningColoring)
from wc_gan_amd.train import CIFAR10_COND, CIFAR10_UNCOND, FlatGradBucket


def test_alphabets_match_reference():
    # generator.py:16-17
    assert NORMS == ['n', 'b', 'd', 'dr']
    assert AFTER_NORMS == ['ucs', 'ccs', 'uccs', 'uconv', 'fconv', 'ufconv', 'cconv', 'ucconv', 'ccsuconv', 'n']
    with pytest.raises(AssertionError):
        create_norm('x', 'uconv')
    with pytest.raises(AssertionError):
        create_norm('d', 'cs')


@pytest.mark.parametrize("after_norm", AFTER_NORMS)
def test_fused_stack_builds_for_every_after_norm(after_norm):
    stack = create_norm('d', after_norm, number_of_classes=7, filters_emb=3)(axis=-1, name='Generator.0.bn1', channels=32)
    assert isinstance(stack, WhiteningColoring)
    assert stack.npart.layer_name == 'Generator.0.bn1_npart'          # generator.py:85
    assert stack.npart.moving_mean.shape == (32, 1) and stack.npart.moving_cov.shape == (32, 32)
    names = [b.layer_name for b in stack.branches]
    assert all(n.startswith('Generator.0.bn1_repart') for n in names)  # generator.py:86
    if after_norm in ('ucconv', 'ufconv', 'uccs', 'ccsuconv'):
        assert names == ['Generator.0.bn1_repart_c', 'Generator.0.bn1_repart_u']   # generator.py:55-57


@pytest.mark.parametrize("after_norm", AFTER_NORMS)
def test_coloring_table_equals_oracle_table(after_norm):
    C, K, E, N = 32, 5, 3, 9
    stack = create_norm('d', after_norm, number_of_classes=K, filters_emb=E)(axis=-1, name='s', channels=C)
    rng = np.random.default_rng(0)
    params = {}
    for br in stack.branches:
        for pn, p in br.named_parameters():
            with torch.no_grad():
                p.copy_(torch.tensor(rng.standard_normal(tuple(p.shape)), dtype=torch.float32))
        kind = type(br).__name__
        if kind == 'Conv11':
            params['u_kernel'] = br.kernel.detach().numpy().reshape(C, C).astype(np.float64)
            params['u_bias'] = br.bias.detach().numpy().astype(np.float64)
        elif kind == 'ConditionalConv11':
            params['c_kernel'] = br.kernel.detach().numpy().astype(np.float64)
            params['c_bias'] = br.bias.detach().numpy().astype(np.float64)
        elif kind == 'FactorizedConv11':
            params['f_kernel'] = br.kernel.detach().numpy().astype(np.float64)
            params['f_alpha'] = br.class_matrix.detach().numpy().astype(np.float64)
        elif kind == 'CenterScale':
            params['u_gamma'] = br.gamma.detach().numpy().astype(np.float64)
            params['u_beta'] = br.beta.detach().numpy().astype(np.float64)
        elif kind == 'ConditionalCenterScale':
            params['c_gamma'] = br.gamma.detach().numpy().astype(np.float64)
            params['c_beta'] = br.beta.detach().numpy().astype(np.float64)
    x = torch.zeros(N, 2, 2, C)
    cls = torch.tensor(rng.integers(0, K, (N, 1)), dtype=torch.int32)
    gamma, beta, slot, _ps = stack.coloring_table(x, cls)
    G_ref, B_ref = o.coloring_table(after_norm, C, params, K)
    if after_norm == 'n':
        assert gamma is None and beta is None
        return
    assert np.abs(gamma.detach().numpy() - G_ref).max() < 1e-5
    if beta is not None:
        assert np.abs(beta.detach().numpy() - B_ref).max() < 1e-6
    else:
        assert np.abs(B_ref).max() == 0
    if gamma.shape[0] > 1:
        assert slot.dtype == torch.int32 and torch.equal(slot, cls.reshape(-1))


def test_more_classes_than_samples_switches_to_per_sample_slots():
    C, K, N = 32, 50, 4
    stack = create_norm('d', 'ucconv', number_of_classes=K)(axis=-1, name='s', channels=C)
    cls = torch.tensor([[3], [49], [3], [0]], dtype=torch.int32)
    gamma, beta, slot, _ps = stack.coloring_table(torch.zeros(N, 1, 1, C), cls)
    assert gamma.shape == (N, C, C) and beta.shape == (N, C)
    assert torch.equal(slot, torch.arange(N, dtype=torch.int32))
    full = stack.branches[0].kernel + stack.branches[1].kernel.view(1, C, C)
    assert torch.allclose(gamma, full[cls.reshape(-1).long()])


def test_generator_matches_recipe_shapes():
    G = make_generator(**CIFAR10_UNCOND['generator'])
    nparams = sum(p.numel() for p in G.parameters())
    assert abs(nparams - 4.73e6) < 0.02e6                  # SURVEY.md section 8e estimate
    sites = [m for m in G.modules() if isinstance(m, WhiteningColoring)]
    assert len(sites) == 7 and all(s.npart.channels == 256 for s in sites)     # row a2 site list
    assert sites[-1].npart.layer_name == 'Generator.BN.Final_npart'            # generator.py:154
    Gc = make_generator(**CIFAR10_COND['generator'])
    kinds = [type(b).__name__ for b in Gc.blocks[0].bn1.branches]
    assert kinds == ['ConditionalConv11', 'Conv11']                             # ucconv, generator.py:52-60
    assert [type(b).__name__ for b in Gc.final_norm.branches] == ['Conv11']     # last site is always uconv


def test_layer_constructor_surface():
    d = DecorelationNormalization(name='x_npart', renorm=True)
    assert d.renorm and d.decomposition == 'cholesky' and d.momentum == 0.99 and d.epsilon == 1e-3
    with pytest.raises(ValueError):
        DecorelationNormalization(decomposition='svd')
    c = ConditionalConv11(filters=32, number_of_classes=10, name='c')
    assert c.kernel.shape == (10, 32, 32) and c.bias.shape == (10, 32)
    f = FactorizedConv11(number_of_classes=10, filters=32, filters_emb=4, use_bias=False, name='f')
    assert f.kernel.shape == (4, 32, 32) and f.class_matrix.shape == (10, 4) and f.bias is None
    u = Conv11(kernel_size=(1, 1), filters=32, name='u')
    assert u.kernel.shape == (1, 1, 32, 32) and u.bias.shape == (32,)
    lazy = Conv11(kernel_size=(1, 1), name='lazy')
    assert lazy.channels is None                                                # Keras-style deferred build


def test_flat_grad_bucket_views_track_backward():
    lin = torch.nn.Linear(4, 3)
    conv = torch.nn.Conv2d(2, 2, 3).to(memory_format=torch.channels_last)
    params = list(lin.parameters()) + list(conv.parameters())
    b = FlatGradBucket(params)
    assert conv.weight.grad.stride() == conv.weight.stride()
    (lin(torch.ones(2, 4)).sum() + conv(torch.ones(1, 2, 5, 5)).sum()).backward()
    assert b.flat.abs().sum() > 0
    assert all(p.grad.data_ptr() >= b.flat.data_ptr() for p in params)
    b.zero()
    assert float(lin.weight.grad.abs().sum()) == 0.0


def test_keras_named_checkpoint_roundtrip(tmp_path):
    """N4: EVERY tensor of the generator under its Keras layer name and in Keras' layout (run.py:79-83 loads by name)."""
    from wc_gan_amd.checkpoint import keras_named_state, load_keras_named, save_keras_named
    G = make_generator(**CIFAR10_COND['generator'])
    st = keras_named_state(G)
    # names follow generator.py:85-86 / 55-58 / 154
    assert 'Generator.BN.Final_npart/moving_cov:0' in st and st['Generator.BN.Final_npart/moving_mean:0'].shape == (128, 1)
    assert st['Generator.BN.Final_repart/kernel:0'].shape == (1, 1, 128, 128)
    assert st['Generator.0.bn1_repart_c/kernel:0'].shape == (10, 128, 128)
    assert st['Generator.0.bn1_repart_u/bias:0'].shape == (128,)
    # the dense layer (generator.py:127, Keras' automatic name), the block convolutions (generator.py:145) and the last
    # convolution (generator.py:154-155), kernels in Keras' (kh, kw, Cin, Cout) / (in, out) layouts
    assert st['dense_1/kernel:0'].shape == (128, 4 * 4 * 128) and st['dense_1/bias:0'].shape == (4 * 4 * 128,)
    assert st['Generator.0.conv1/kernel:0'].shape == (3, 3, 128, 128) and st['Generator.2.shortcut/kernel:0'].shape == (1, 1, 128, 128)
    assert st['Generator.Final/kernel:0'].shape == (3, 3, 128, 3) and st['Generator.Final/bias:0'].shape == (3,)
    w = G.blocks[1].conv2.conv.weight.detach().numpy()                       # torch (Cout, Cin, kh, kw)
    assert np.array_equal(st['Generator.1.conv2/kernel:0'], np.transpose(w, (2, 3, 1, 0)))
    assert np.array_equal(st['dense_1/kernel:0'], G.dense.weight.detach().numpy().T)
    # every parameter and every persistent buffer of the generator is covered
    n_tensors = len(list(G.parameters())) + sum(1 for n, _ in G.named_buffers() if 'moving_' in n)
    assert len(st) == n_tensors
    p = str(tmp_path / "g.npz")
    save_keras_named(G, p)
    torch.manual_seed(123)
    G2 = make_generator(**CIFAR10_COND['generator'])
    assert not np.array_equal(keras_named_state(G2)['Generator.0.conv1/kernel:0'], st['Generator.0.conv1/kernel:0'])
    loaded = load_keras_named(G2, p)
    assert len(loaded) == len(st)
    for k, v in keras_named_state(G2).items():
        assert np.array_equal(v, st[k])
    assert G2.blocks[0].conv1.conv.weight.is_contiguous(memory_format=torch.channels_last)       # strides kept
    with pytest.raises(KeyError):
        load_keras_named(G2, {k: v for k, v in list(st.items())[:-1]})
    with pytest.raises(ValueError):
        bad = dict(st); bad['Generator.Final/kernel:0'] = bad['Generator.Final/kernel:0'][:, :, :, :2]
        load_keras_named(G2, bad)
    # spectrally normalised generator (generator.py:104-113): u under the layer's name, this build's v optional on load
    Gs = make_generator(block_sizes=(32,), resamples=("UP",), first_block_shape=(4, 4, 32), block_norm='d', block_after_norm='uconv',
                        last_norm='d', last_after_norm='uconv', spectral=True)
    ss = keras_named_state(Gs)
    assert ss['sn_dense_1/u:0'].shape == (1, 4 * 4 * 32) and ss['Generator.0.conv1/u:0'].shape == (1, 32)
    Gs2 = make_generator(block_sizes=(32,), resamples=("UP",), first_block_shape=(4, 4, 32), block_norm='d', block_after_norm='uconv',
                         last_norm='d', last_after_norm='uconv', spectral=True)
    load_keras_named(Gs2, {k: v for k, v in ss.items() if not k.endswith('/v:0')})                    # an upstream file has no v
    assert torch.equal(Gs2.blocks[0].conv1.conv.sn_u, Gs.blocks[0].conv1.conv.sn_u)
    # ... so v is rebuilt from the LOADED weight and u (the random-init v would give a wrong sigma in eval mode, where
    # the op runs no iteration and takes sigma = u^T W v as stored): v = normalize(W^T u), sigma as the source model's
    for a, b in ((Gs2.blocks[0].conv1.conv, Gs.blocks[0].conv1.conv), (Gs2.dense, Gs.dense)):
        wm = a._as_matrix(a.weight.detach())
        assert torch.allclose(a.sn_v, torch.nn.functional.normalize(wm.t().mv(a.sn_u), dim=0), atol=1e-6)
        sig = lambda m: float(m.sn_u @ m._as_matrix(m.weight.detach()) @ m.sn_v)
        # (one half-step of the power iteration further than the source's stored v: equal at convergence, 2e-3 apart after the
        # constructor's 15 warm-up steps; the random-init v this replaces gave sigma ~ 0.1 of it)
        assert abs(sig(a) - sig(b)) < 1e-2 * abs(sig(b)) and sig(a) >= sig(b) * (1 - 1e-6)


def test_keras_named_checkpoint_covers_embedding_batchnorm_and_any_dense_index():
    """`Generator.emb` (concat_cls, generator.py:120-121 / run.py:175) and the moving statistics of norm == 'b'
    (generator.py:22) travel too; a tensor that no entry names raises instead of being dropped; `dense_N` loads for any N."""
    from wc_gan_amd.checkpoint import keras_named_state, load_keras_named
    kw = dict(block_sizes=(32,), resamples=("UP",), first_block_shape=(4, 4, 32), number_of_classes=7, concat_cls=True,
              block_norm='b', block_after_norm='ucs', last_norm='d', last_after_norm='uconv', gan_type='AC_GAN')
    G = make_generator(**kw)
    z, cls = torch.randn(4, 128), torch.randint(0, 7, (4, 1))
    # the BatchNorm layers build on their first call (Keras-style): before it there is nothing to save for them
    st0 = keras_named_state(G)
    assert 'embedding_1/embeddings:0' in st0 and st0['embedding_1/embeddings:0'].shape == (7, 32)
    for blk in G.blocks:                                     # build the lazily created BatchNorm state without running HIP layers
        for bn in (blk.bn1, blk.bn2):
            bn.norm_layer(torch.randn(2, 4, 4, 32))
    st = keras_named_state(G)
    assert st['Generator.0.bn1_npart/moving_variance:0'].shape == (32,) and 'Generator.0.bn2_npart/moving_mean:0' in st
    torch.manual_seed(5)
    G2 = make_generator(**kw)
    for blk in G2.blocks:
        for bn in (blk.bn1, blk.bn2):
            bn.norm_layer(torch.randn(2, 4, 4, 32) * 3 + 1)
    renamed = {k.replace('dense_1/', 'dense_7/').replace('embedding_1/', 'embedding_3/'): v for k, v in st.items()}
    load_keras_named(G2, renamed)                            # Keras' automatic names count per process: any index loads
    for k, v in keras_named_state(G2).items():
        assert np.array_equal(v, st[k]), k
    assert torch.equal(G2.emb.weight, G.emb.weight)
    # a weight the walk does not know: loud, both ways
    G2.extra = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(NotImplementedError):
        keras_named_state(G2)
    with pytest.raises(NotImplementedError):
        load_keras_named(G2, st)


def test_keras_named_checkpoint_loads_the_legacy_sn_embedding_spelling():
    """rounds 3-4 of this build wrote a spectrally normalised embedding as `sn_embedding_<n>` (two d); upstream's class is
    SNEmbeding and Keras names it `sn_embeding_<n>`, which is what _entries emits since round 5.  A file with the old spelling -- and any
    index -- must still load (it raised `missing weight`, or loaded nothing with strict=False)."""
    from wc_gan_amd.checkpoint import keras_named_state, load_keras_named
    from wc_gan_amd.spectral import SNEmbedding

    class D(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = SNEmbedding(7, 12)

    torch.manual_seed(2)
    a, b = D(), D()
    st = keras_named_state(a)
    assert any(k.startswith('sn_embeding_1/') for k in st) and not any(k.startswith('sn_embedding_') for k in st)
    legacy = {k.replace('sn_embeding_1/', 'sn_embedding_5/'): v for k, v in st.items()}
    seen = load_keras_named(b, legacy)
    assert len(seen) == len(st)
    for k, v in keras_named_state(b).items():
        assert np.array_equal(v, st[k]), k
    c = D()
    assert len(load_keras_named(c, legacy, strict=False)) == len(st) and torch.equal(c.emb.weight, a.emb.weight)


def test_h5_converter_round_trip_when_h5py_is_available(tmp_path):
    """tools/h5_to_npz.py (runs wherever h5py exists; skipped here when it does not)."""
    pytest.importorskip("h5py")
    import importlib.util, os
    from wc_gan_amd.checkpoint import keras_named_state, save_keras_named
    spec = importlib.util.spec_from_file_location("h5_to_npz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "h5_to_npz.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    G = make_generator(**CIFAR10_COND['generator'])
    a, h, b = str(tmp_path / "a.npz"), str(tmp_path / "g.h5"), str(tmp_path / "b.npz")
    save_keras_named(G, a)
    mod.npz_to_h5(a, h); mod.h5_to_npz(h, b)
    sa, sb = dict(np.load(a)), dict(np.load(b))
    assert sorted(sa) == sorted(sb) and all(np.array_equal(sa[k], sb[k]) for k in sa)


def test_all_four_gpu_configs_and_their_site_lists():
    """BASELINE.json:configs 2-5 as wc_gan_amd.train.CONFIGS, shapes per run.py:147-237 and the four scripts."""
    from wc_gan_amd.train import CONFIGS, wc_sites
    assert sorted(CONFIGS) == ['cifar10_cond', 'cifar10_uncond', 'stl10_uncond', 'tinyimagenet_cond_sa']
    hw = lambda name, n: [(h, w, c) for _, _, h, w, c in wc_sites(CONFIGS[name], n)]
    assert hw('cifar10_uncond', 128) == [(4, 4, 256), (8, 8, 256), (8, 8, 256), (16, 16, 256), (16, 16, 256), (32, 32, 256), (32, 32, 256)]
    # run.py:152 first_block_w = 6 for stl10, run.py:333 48 x 48 images
    assert hw('stl10_uncond', 128) == [(6, 6, 256), (12, 12, 256), (12, 12, 256), (24, 24, 256), (24, 24, 256), (48, 48, 256), (48, 48, 256)]
    # run.py:155-158 four UP blocks, run.py:335 64 x 64 images
    assert hw('tinyimagenet_cond_sa', 128)[-1] == (64, 64, 128) and len(hw('tinyimagenet_cond_sa', 128)) == 9
    t = CONFIGS['tinyimagenet_cond_sa']
    assert t['generator']['number_of_classes'] == 200 and t['generator']['filters_emb'] == 15          # run.py:172-173, script line 7
    assert t['discriminator']['block_sizes'] == (256, 512, 1024, 1024, 1024)                              # run.py:205-208 at filters 1024
    assert t['discriminator']['resamples'] == ('DOWN', 'DOWN', 'DOWN', 'SAME', 'SAME')
    G = make_generator(**t['generator'])
    kinds = [type(b).__name__ for b in G.blocks[0].bn1.branches]
    assert kinds == ['FactorizedConv11', 'Conv11']                                                       # ufconv, generator.py:69-78
    assert G.blocks[0].bn1.branches[0].class_matrix.shape == (200, 15)
    sites = [m for m in G.modules() if isinstance(m, WhiteningColoring)]
    assert [s.npart.layer_name for s in sites][-1] == 'Generator.BN.Final_npart' and len(sites) == 9


def test_per_sample_tables_are_decided_per_statistic_group():
    """200 classes against a grouped batch of 5 x 64: 200 > 64 samples per group -> per-sample tables (320), not 5 x 200."""
    from wc_gan_amd.layers import statistic_groups
    C, K, N = 32, 200, 320
    stack = create_norm('d', 'ufconv', number_of_classes=K, filters_emb=4)(axis=-1, name='s', channels=C)
    cls = torch.randint(0, K, (N, 1), dtype=torch.int32)
    gamma, beta, slot, per_sample = stack.coloring_table(torch.zeros(N, 1, 1, C), cls)
    assert not per_sample and gamma.shape[0] == K                    # ungrouped: 200 classes <= 320 samples
    with statistic_groups(5):
        gamma, beta, slot, per_sample = stack.coloring_table(torch.zeros(N, 1, 1, C), cls)
    assert per_sample and gamma.shape == (N, C, C) and torch.equal(slot, torch.arange(N, dtype=torch.int32))


def test_supports_statistic_groups_and_refusals():
    from wc_gan_amd.layers import statistic_groups, supports_statistic_groups
    from wc_gan_amd.train import CONFIGS
    assert supports_statistic_groups(make_generator(**CONFIGS['cifar10_uncond']['generator']))
    assert not supports_statistic_groups(make_generator(block_sizes=(32,), resamples=("UP",), first_block_shape=(4, 4, 32),
                                                        block_norm='dr', block_after_norm='uconv', last_norm='d', last_after_norm='uconv'))
    assert not supports_statistic_groups(make_generator(block_sizes=(48,), resamples=("UP",), first_block_shape=(4, 4, 48),
                                                        block_norm='d', block_after_norm='uconv', last_norm='d', last_after_norm='uconv'))
    G_b = make_generator(block_sizes=(8,), resamples=("UP",), first_block_shape=(4, 4, 8), block_norm='b', block_after_norm='ucs',
                         last_norm='b', last_after_norm='ucs')
    with torch.no_grad():
        G_b(torch.zeros(2, 128), torch.zeros(2, 1, dtype=torch.int32))
    assert not supports_statistic_groups(G_b)
    # a layer without the grouped form raises inside the context instead of pooling the statistics
    layer = DecorelationNormalization(name='r', renorm=True, channels=32)
    with statistic_groups(2), pytest.raises(RuntimeError):
        layer.transform(torch.zeros(4, 2, 2, 32))
    # sync-WC: the grouped forward has no collective, so a layer with a process group refuses it (its critic-phase fakes would
    # be whitened with per-replica statistics and the replicas' moving statistics would drift apart) and the trainer falls
    # back to separate passes
    sync = DecorelationNormalization(name='s', channels=32, process_group=object())
    with statistic_groups(2), pytest.raises(RuntimeError, match="sync-WC"):
        sync.transform(torch.zeros(4, 2, 2, 32))
    G_s = make_generator(**CONFIGS['cifar10_uncond']['generator'])
    for m in G_s.modules():
        if isinstance(m, DecorelationNormalization):
            m.process_group = object()
            m.channels = 256
    assert not supports_statistic_groups(G_s)
    # the setting is per host thread
    import threading
    seen = []
    with statistic_groups(3):
        t = threading.Thread(target=lambda: seen.append(__import__('wc_gan_amd.layers', fromlist=['x'])._stat_groups()))
        t.start(); t.join()
    assert seen == [1]


def test_discriminator_keyword_defaults_are_the_references():
    import inspect
    from wc_gan_amd.discriminator import make_discriminator
    d = {k: v.default for k, v in inspect.signature(make_discriminator).parameters.items()}
    # discriminator.py:15-20
    assert d['type'] == 'AC_GAN' and d['spectral'] is False and d['sum_pool'] is False and d['norm'] == 'n'
    assert d['conv_singular'] is True and d['block_sizes'] == (128, 128, 128, 128)
    D = make_discriminator(input_image_shape=(8, 8, 3), block_sizes=(8, 8), resamples=('DOWN', 'SAME'))
    out, cls_out = D(torch.zeros(2, 8, 8, 3))
    assert out.shape == (2, 1) and cls_out.shape == (2, 10)
    assert type(D.cls_out) is torch.nn.Linear                          # plain Dense class head, discriminator.py:74


def test_ranks_that_share_a_gpu_get_the_k2_form_without_an_in_launch_wait(monkeypatch):
    """_lib.shared_gpu_guard: more local ranks than visible devices -> WC_K2_TWO_LAUNCH=1 before the library reads its environment; one rank per
    device, a single process, or an explicit setting: nothing is touched ("Abstractions": the user no longer has to know the switch)."""
    from wc_gan_amd import _lib
    monkeypatch.delenv("WC_K2_TWO_LAUNCH", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert _lib.shared_gpu_guard(device_count=1) is False and "WC_K2_TWO_LAUNCH" not in os.environ
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _lib.shared_gpu_guard(device_count=8) is False and "WC_K2_TWO_LAUNCH" not in os.environ
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"): monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert _lib.shared_gpu_guard(device_count=1) is False and "WC_K2_TWO_LAUNCH" not in os.environ           # devices masked per rank: no verdict
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert _lib.shared_gpu_guard(device_count=1) is True and os.environ["WC_K2_TWO_LAUNCH"] == "1"
    assert _lib.shared_gpu_guard(device_count=1) is False and os.environ["WC_K2_TWO_LAUNCH"] == "1"      # explicit now: left alone
    monkeypatch.setenv("WC_K2_TWO_LAUNCH", "0")
    assert _lib.shared_gpu_guard(device_count=1) is False and "WC_K2_TWO_LAUNCH" not in os.environ           # "0" = the one-launch form, by request
    monkeypatch.delenv("WC_K2_TWO_LAUNCH", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "two")
    assert _lib.shared_gpu_guard(device_count=1) is False
