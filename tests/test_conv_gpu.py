"""The block convolutions on the split-fp16 MFMA kernel (wc_gan_amd/conv.py, csrc/wc_conv.hip) against torch's
float64 convolutions of the same map (Keras Conv2D 'same', generator.py:142-158; its UpSampling2D / AveragePooling2D
forms as 4x4 stride-2 transposed / strided convolutions).  Tolerance: 1e-5 of the result's max (fp32 convolution level;
north_star asks for 1e-4 relative)."""
import pytest
import torch
import torch.nn.functional as F

TOL = 1e-5


def _ref(x, w, b, kind):
    xn = x.permute(0, 3, 1, 2).double()
    bb = None if b is None else b.double()
    if kind == 'same':
        y = F.conv2d(xn, w.double(), bb, padding=w.shape[2] // 2)
    elif kind == 'down':
        y = F.conv2d(xn, w.double(), bb, stride=2, padding=1)
    else:
        y = F.conv_transpose2d(xn, w.double(), bb, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def _weights(kind, ci, co, k, channels_last=True):
    shape = (ci, co, 4, 4) if kind == 'up' else (co, ci, 4, 4) if kind == 'down' else (co, ci, k, k)
    w = torch.randn(*shape, device='cuda') / (ci * shape[2] * shape[3]) ** 0.5
    return w.contiguous(memory_format=torch.channels_last) if channels_last else w


def _rel(a, r):
    return float(((a.detach().double() - r.detach()).abs().max() / r.detach().abs().max()).item())


CASES = [
    # kind, N, H, W, Cin, Cout, k, taken
    ('same', 2, 8, 8, 128, 128, 3, True),
    ('same', 8, 12, 12, 128, 256, 3, True),          # sizes that are no powers of two (STL-10's 12 x 12)
    ('same', 4, 8, 8, 256, 128, 1, True),
    ('same', 16, 8, 8, 256, 256, 3, True),
    ('same', 1, 16, 16, 128, 384, 3, True),          # three n-tiles of 128
    ('down', 2, 16, 16, 128, 128, 0, True),
    ('down', 32, 12, 12, 128, 256, 0, True),
    ('up', 2, 8, 8, 128, 128, 0, True),
    ('up', 32, 4, 4, 256, 256, 0, True),
    ('same', 128, 16, 16, 256, 256, 3, True),        # the 256-point tile
    ('up', 8, 6, 6, 256, 256, 0, False),             # 288 points: not a multiple of the 128-point tile
    ('same', 2, 8, 8, 32, 128, 3, False),            # the data gradient would produce 32 channels
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,H,W,ci,co,k,taken", CASES)
def test_forward_and_gradients_match_float64(kind, N, H, W, ci, co, k, taken):
    from wc_gan_amd import conv as C
    torch.manual_seed(N + H + ci)
    x = (torch.randn(N, H, W, ci, device='cuda') * 1.7 + 0.3).requires_grad_(True)
    w = _weights(kind, ci, co, k).requires_grad_(True)
    b = (torch.randn(co, device='cuda') * 0.1).requires_grad_(True)
    assert C.supported(x, w, kind) == taken
    if not taken:
        with pytest.raises(RuntimeError):
            C.fast_conv(x, w, b, kind)
        return
    y = C.fast_conv(x, w, b, kind)
    gy = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
    y64 = _ref(x, w, b, kind)
    dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
    assert _rel(y, y64) < TOL
    assert _rel(dx, dx64) < TOL
    assert _rel(dw, dw64) < 2e-5          # MIOpen's fp32 weight gradient
    assert _rel(db, db64) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,H,ci,co", [('down3', 8, 16, 128, 128), ('down3', 32, 12, 128, 256), ('up3', 8, 8, 128, 128),
                                            ('up3', 32, 4, 256, 256), ('up3', 128, 16, 256, 256)])
def test_pooled_and_upsampled_3x3_layers_from_the_3x3_weight(kind, N, H, ci, co):
    """Conv2D 3x3 -> AveragePooling2D and UpSampling2D -> Conv2D 3x3 (generator.py:144-151, discriminator.py:41-54) given
    the 3x3 weight: the 4x4 stride-2 kernels are formed inside the weight image, the weight gradient is folded back."""
    from wc_gan_amd import conv as C
    torch.manual_seed(N + H)
    x = (torch.randn(N, H, H, ci, device='cuda') * 1.3 - 0.2).requires_grad_(True)
    w = _weights('same', ci, co, 3).requires_grad_(True)
    b = (torch.randn(co, device='cuda') * 0.1).requires_grad_(True)
    assert C.supported(x, w, kind)
    y = C.fast_conv(x, w, b, kind)
    xn = x.permute(0, 3, 1, 2).double()
    if kind == 'down3':
        y64 = F.avg_pool2d(F.conv2d(xn, w.double(), b.double(), padding=1), 2).permute(0, 2, 3, 1)
    else:
        y64 = F.conv2d(F.interpolate(xn, scale_factor=2, mode='nearest'), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert y.shape == y64.shape
    gy = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
    dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
    assert _rel(y, y64) < TOL and _rel(dx, dx64) < TOL and _rel(dw, dw64) < TOL and _rel(db, db64) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ['same', 'down3'])
def test_relu_in_the_split_equals_relu_then_layer(kind):
    """conv(relu(x)) with the ReLU done while x is split; the mask comes back in the data gradient"""
    from wc_gan_amd import conv as C
    torch.manual_seed(11)
    x = torch.randn(8, 16, 16, 128, device='cuda').requires_grad_(True)
    w = _weights('same', 128, 128, 3).requires_grad_(True)
    b = (torch.randn(128, device='cuda') * 0.1).requires_grad_(True)
    y = C.fast_conv_or_none(x, w, b, kind, relu_input=True)
    z = F.conv2d(F.relu(x.permute(0, 3, 1, 2).double()), w.double(), b.double(), padding=1)
    y64 = (F.avg_pool2d(z, 2) if kind == 'down3' else z).permute(0, 2, 3, 1)
    gy = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
    dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
    assert _rel(y, y64) < TOL and _rel(dx, dx64) < TOL and _rel(dw, dw64) < TOL and _rel(db, db64) < TOL
    assert bool(((dx == 0) | (x > 0)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("ci", [32, 96])
def test_forward_with_reduction_channels_in_multiples_of_32(ci):
    """the kernel itself takes any multiple of 32 reduction channels (the layer entry also wants the data gradient)"""
    from wc_gan_amd import conv as C
    torch.manual_seed(ci)
    x = torch.randn(2, 8, 8, ci, device='cuda')
    w = _weights('same', ci, 128, 3)
    (gf, kf, nf), _ = C._geoms('same', 2, 8, 8, w)
    y = C.run(C.split_planes(x), C.weight_image(w, gf, kf, nf), gf)
    assert _rel(y, _ref(x, w, None, 'same')) < TOL
    y = C.run(C.split_planes(x, relu=True), C.weight_image(w, gf, kf, nf), gf, relu=True)      # ReLU before and after
    assert _rel(y, _ref(x.clamp_min(0), w, None, 'same').clamp_min(0)) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1e-6, 1.0, 3e4])
def test_tensor_scales_keep_small_and_large_magnitudes_accurate(scale):
    """gradients of 1e-6 and activations of 3e4 both go through fp16 planes: the power-of-two tensor scale keeps the
    relative accuracy"""
    from wc_gan_amd import conv as C
    torch.manual_seed(3)
    x = torch.randn(2, 8, 8, 128, device='cuda') * scale
    w = _weights('same', 128, 128, 3) * (1.0 / scale if scale < 1 else 1.0) * 1e-3
    y = C.fast_conv(x, w, None, 'same')
    assert torch.isfinite(y).all()
    assert _rel(y, _ref(x, w, None, 'same')) < TOL


@pytest.mark.gpu
def test_zero_input_and_wide_dynamic_range():
    from wc_gan_amd import conv as C
    w = _weights('same', 128, 128, 3)
    x = torch.zeros(2, 8, 8, 128, device='cuda')
    b = torch.randn(128, device='cuda')
    y = C.fast_conv(x, w, b, 'same')
    assert torch.equal(y, b.expand_as(y))
    torch.manual_seed(5)
    x = torch.randn(2, 8, 8, 128, device='cuda') * torch.logspace(-6, 2, 128, device='cuda')      # channels 8 decades apart
    assert _rel(C.fast_conv(x, w, None, 'same'), _ref(x, w, None, 'same')) < TOL


@pytest.mark.gpu
def test_convolution_follows_in_place_weight_updates_that_keep_the_version_counter():
    """torch's fused Adam (and replayed graphs) change a weight without bumping `_version`: nothing may be cached on it"""
    from wc_gan_amd import conv as C
    torch.manual_seed(7)
    x = torch.randn(2, 8, 8, 128, device='cuda')
    w = _weights('same', 128, 128, 3).requires_grad_(True)
    y1 = C.fast_conv(x, w, None, 'same').detach()
    opt = torch.optim.Adam([w], lr=0.05, fused=True)
    w.grad = torch.ones_like(w)
    v = w._version
    opt.step()
    y2 = C.fast_conv(x, w, None, 'same').detach()
    assert _rel(y2, _ref(x, w.detach(), None, 'same')) < TOL
    assert float((y2 - y1).abs().max()) > 1e-3 or w._version != v


@pytest.mark.gpu
def test_generator_and_critic_blocks_agree_with_miopen_path(monkeypatch):
    """the same networks with the kernel switched off (MIOpen fp32) give the same images and critic outputs"""
    import wc_gan_amd.generator as G
    from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
    tr = build_trainer(CIFAR10_UNCOND, 'cuda', batch_size=8, seed=3)
    torch.manual_seed(0)
    z = torch.randn(16, 128, device='cuda'); cls = torch.zeros(16, 1, dtype=torch.int32, device='cuda')
    tr.G.eval(); tr.D.eval()
    with torch.no_grad():
        monkeypatch.setattr(G, 'FAST_CONV', True)
        img_fast = tr.G(z, cls); out_fast = tr._d(img_fast, None)
        monkeypatch.setattr(G, 'FAST_CONV', False)
        img_ref = tr.G(z, cls); out_ref = tr._d(img_fast, None)
    assert float((img_fast - img_ref).abs().max()) < 2e-5            # tanh outputs in [-1, 1]
    assert float((out_fast - out_ref).abs().max()) < 1e-4 * float(out_ref.abs().max()) + 1e-5


def test_no_cpu_path():
    from wc_gan_amd import conv as C
    x = torch.randn(2, 8, 8, 32)
    w = torch.randn(128, 32, 3, 3)
    assert not C.supported(x, w, 'same')
    with pytest.raises(RuntimeError):
        C.fast_conv(x, w, None, 'same')


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the split's scale from the previous call of the same site (wc_conv_split_hist_f32: one launch, no absmax pass)
# ---------------------------------------------------------------------------------------------------------------------
class _Site:
    """a layer object as conv.split_planes sees it: something with a __dict__ that lives as long as the layer"""


def _back(planes):
    hi, lo, scale = planes[:3]
    return (hi.double() + lo.double()) / float(scale[0])


@pytest.mark.gpu
def test_history_scaled_split_keeps_22_bits_while_the_tensor_moves_by_orders_of_magnitude():
    """First call of a site = the measured maximum (the two-launch form, bit for bit); every later call takes the scale the PREVIOUS call's
    maximum asks for, with 64 x of headroom: the planes carry the tensor to 2^-20 of its maximum through a 100-fold growth and a 1000-fold
    shrink from one call to the next; the column sums that ride along (the bias gradient) are the classic ones."""
    from wc_gan_amd import conv as C
    torch.manual_seed(3)
    site = _Site()
    x = torch.randn(32, 16, 16, 128, device='cuda') * 1.3 + 0.2
    first = C.split_planes(x, site=site)
    classic = C.split_planes(x)
    assert torch.equal(first[0], classic[0]) and torch.equal(first[1], classic[1]) and torch.equal(first[2][:1], classic[2][:1])
    for factor in (1.0, 2.5, 250.0, 0.25, 1.0, 37.0):          # call to call: x 2.5, x 100, / 1000, x 4, x 37
        xx = x * factor
        got = C.split_planes(xx, site=site)
        s = float(got[2][0])
        assert s > 0 and torch.log2(got[2][0]).item() == int(torch.log2(got[2][0]).item())
        assert bool(torch.isfinite(got[0].float()).all()) and bool(torch.isfinite(got[1].float()).all())
        err = float((_back(got) - xx.double()).abs().max() / xx.abs().max())
        assert err < 2.0 ** -20, (factor, err)
    # with the ReLU in the split and the column sums of the tensor as given (the bias gradient's partial rows)
    g = C.split_planes(x, relu=True, colsum=True, site=site, role='g')          # (a fresh role: its first call measures)
    g2 = C.split_planes(x * 3, relu=True, colsum=True, site=site, role='g')
    ref = C.split_planes(x * 3, relu=True, colsum=True)
    assert float((_back(g2) - (x * 3).clamp_min(0).double()).abs().max() / (x * 3).abs().max()) < 2.0 ** -20
    assert torch.equal(g2[3], ref[3]) and g[3].shape == ref[3].shape
    # a layer in eval mode measures every tensor: the planes do not depend on what the site saw before
    site.training = False
    ev, cl = C.split_planes(x * 50, site=site), C.split_planes(x * 50)
    assert torch.equal(ev[0], cl[0]) and torch.equal(ev[1], cl[1]) and torch.equal(ev[2][:1], cl[2][:1])


@pytest.mark.gpu
def test_history_scaled_split_gives_the_same_bits_eagerly_and_from_a_graph():
    """The record lives on the device and is updated by the kernel itself, so the planes of a sequence of calls are a function of the
    sequence of tensors alone: three calls replayed from one hipGraph (twice) equal the same six calls made eagerly on a fresh site."""
    from wc_gan_amd import conv as C
    torch.manual_seed(6)
    xs = [torch.randn(16, 16, 16, 128, device='cuda') * f for f in (1.0, 7.0, 0.05, 3.0, 3.0, 0.5, 11.0)]
    eager_site, graph_site = _Site(), _Site()
    eager = [C.split_planes(x, site=eager_site) for x in xs]
    first = C.split_planes(xs[0], site=graph_site)                  # the measuring call, eager on both sides
    buf = [torch.empty_like(xs[0]) for _ in range(3)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = [C.split_planes(b, site=graph_site) for b in buf]
    assert torch.equal(first[0], eager[0][0])
    for rep in range(2):
        for b, x in zip(buf, xs[1 + 3 * rep: 4 + 3 * rep]):
            b.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            e = eager[1 + 3 * rep + k]
            assert torch.equal(o[0], e[0]) and torch.equal(o[1], e[1]) and torch.equal(o[2][:1], e[2][:1]), (rep, k)


def _redo_count(site, role='x'):
    from wc_gan_amd import conv as C
    return int(site.__dict__['_wc_split_hist'][role][0].view(torch.int32)[C.HIST_REDO])


@pytest.mark.gpu
def test_history_scaled_split_outside_its_window_is_split_again_with_the_measured_scale():
    """Round 6: the previous call's maximum can be wrong in three ways -- the tensor grew more than 255-fold
    (round 5: inf in the planes, NaN in the weights one optimizer step later, no redo), it shrank more than 4096-fold (round 5: the lo plane
    in fp16's subnormals, quietly fewer bits), or the previous tensor was ALL ZERO (round 5: scale 1.0, whatever the next tensor is -- a
    hinge critic whose margins are all met hands back exactly-zero gradients).  The output gradients (role 'g') take a gated second launch
    that sees the tensor's own maximum in the record and splits again with the measured scale: the planes are then the two-launch form's,
    bit for bit, and the site's counter says the second pass ran.  Inside the window it does not run."""
    from wc_gan_amd import conv as C
    torch.manual_seed(4)
    site = _Site()
    x = torch.randn(8, 16, 16, 128, device='cuda')
    sp = lambda t, **kw: C.split_planes(t, site=site, role='g', **kw)
    rc = lambda s=site: _redo_count(s, 'g')
    sp(x)
    assert rc() == 0
    ok = sp(x * 4.0)                                               # inside the window: history scale, no second pass
    assert rc() == 0 and float(ok[2][0]) != float(C.split_planes(x * 4.0)[2][0])
    def exact(t, got):
        ref = C.split_planes(t)
        return torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2][:1], ref[2][:1])
    big = sp(x * 12000.0)                                          # x 3000 from one call to the next
    assert bool(torch.isfinite(big[0].float()).all()) and exact(x * 12000.0, big) and rc() == 1
    small = sp(x * 0.12)                                           # / 100000
    assert exact(x * 0.12, small) and rc() == 2
    assert float((_back(small) - (x * 0.12).double()).abs().max() / (x * 0.12).abs().max()) < 2.0 ** -20
    # zeros, then a tensor of size 1e-4: 2^-20 of its maximum
    z = sp(torch.zeros_like(x))
    assert rc() == 2 and not bool(z[0].any()) and not bool(z[1].any())
    z2 = sp(torch.zeros_like(x))                                   # zeros after zeros: nothing to redo either
    assert rc() == 2 and not bool(z2[0].any())
    tiny = x * 1.0e-4                                              # (0.12 / 1e-4 = 1200-fold below what the site last saw: inside the window)
    got = sp(tiny)
    assert rc() == 2
    assert float((_back(got) - tiny.double()).abs().max() / tiny.abs().max()) < 2.0 ** -20
    sp(torch.zeros_like(x))
    tinier = x * 1.0e-9                                            # ... and 1e5-fold below it: the second pass
    got = sp(tinier)
    assert exact(tinier, got) and rc() == 3
    assert float((_back(got) - tinier.double()).abs().max() / tinier.abs().max()) < 2.0 ** -20
    nxt = sp(tinier * 2.0)                                         # the record now holds the measured maximum: history again
    assert rc() == 3
    assert float((_back(nxt) - (tinier * 2.0).double()).abs().max() / (tinier * 2.0).abs().max()) < 2.0 ** -20
    # with the ReLU in the split: the maximum that counts is the one of what is split (a tensor whose positive part is tiny)
    site2 = _Site()
    y = torch.where(x > 0, x * 1.0e-5, x * 50.0)
    C.split_planes(x, relu=True, site=site2, role='g')
    gr = C.split_planes(y, relu=True, site=site2, role='g')
    yr = y.clamp_min(0)
    assert _redo_count(site2, 'g') == 1
    assert float((_back(gr) - yr.double()).abs().max() / yr.abs().max()) < 2.0 ** -20
    # (the two-launch form takes its scale from max |y| BEFORE the ReLU -- 225 here -- and leaves this positive part 2^-9 of it: the
    # second pass scales for what is actually split)
    ref = C.split_planes(y, relu=True)
    assert float((_back(ref) - yr.double()).abs().max() / yr.abs().max()) > 2.0 ** -20


@pytest.mark.gpu
def test_history_scaled_split_of_a_layer_input_survives_all_zero_tensors_without_the_second_launch():
    """The layer inputs (role 'x') do not take the second launch (+2.3 us per call: every split of the step would give back what the history
    saves): an all-zero tensor leaves the site's scale where it was -- the record carries the maximum a call assumed -- so the tensor after
    it, and after any number of them, is split to 2^-20 of its maximum as long as it lies within the window of the last non-zero one."""
    from wc_gan_amd import conv as C
    torch.manual_seed(9)
    site = _Site()
    x = torch.randn(8, 16, 16, 128, device='cuda')
    C.split_planes(x, site=site)
    for k in range(3):
        z = C.split_planes(torch.zeros_like(x), site=site)
        assert not bool(z[0].any()) and not bool(z[1].any())
        t = x * (0.5 ** k) * 3.0
        got = C.split_planes(t, site=site)
        assert float((_back(got) - t.double()).abs().max() / t.abs().max()) < 2.0 ** -20, k
    assert _redo_count(site) == 0


@pytest.mark.gpu
def test_history_scaled_split_second_pass_inside_a_graph():
    """The gate is on the device: a captured sequence takes the second pass on exactly the replays whose tensors need it."""
    from wc_gan_amd import conv as C
    torch.manual_seed(8)
    site = _Site()
    x = torch.randn(8, 16, 16, 128, device='cuda')
    C.split_planes(x, site=site, role='g')
    buf = torch.empty_like(x)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = C.split_planes(buf, site=site, role='g')
    for factor, redo in ((2.0, 0), (0.0, 0), (1.0e-4, 1), (3.0e-4, 0), (40.0, 1), (40.0, 0)):
        before = _redo_count(site, 'g')
        buf.copy_(x * factor)
        graph.replay()
        torch.cuda.synchronize()
        assert _redo_count(site, 'g') - before == redo, (factor, redo)
        if factor:
            assert float((_back(out) - buf.double()).abs().max() / buf.abs().max()) < 2.0 ** -20, factor
            assert bool(torch.isfinite(out[0].float()).all())


@pytest.mark.gpu
def test_convolution_with_history_scaled_splits_over_several_steps():
    """conv.fast_conv_or_none(site=...) called repeatedly with inputs and output gradients whose magnitudes drift: output and all three
    gradients against float64 at the kernel's tolerance on every call (the first measures, the others use the history); replayed from a
    hipGraph the recorded calls keep working (every node reads the record the node before it left)."""
    from wc_gan_amd import conv as C
    torch.manual_seed(5)
    site = _Site()
    w = _weights('same', 128, 128, 3).requires_grad_(True)
    b = (torch.randn(128, device='cuda') * 0.1).requires_grad_(True)
    for step, (sx, sg) in enumerate(((1.0, 1.0), (1.6, 0.3), (0.4, 5.0), (3.0, 0.01), (1.0, 1.0))):
        x = (torch.randn(16, 8, 8, 128, device='cuda') * sx).requires_grad_(True)
        y = C.fast_conv_or_none(x, w, b, 'same', site=site)
        gy = torch.randn_like(y) * sg
        dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
        y64 = _ref(x, w, b, 'same')
        dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
        assert _rel(y, y64) < TOL and _rel(dx, dx64) < TOL and _rel(dw, dw64) < 2e-5 and _rel(db, db64) < TOL, step
    xs = torch.randn(16, 8, 8, 128, device='cuda')
    out = torch.empty(2, 16, 8, 8, 128, device='cuda')
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with torch.cuda.graph(graph):
            out[0].copy_(C.fast_conv_or_none(xs, w, b, 'same', site=site))
            out[1].copy_(C.fast_conv_or_none(xs * 2, w, b, 'same', site=site))
    for rep in range(3):
        xs.normal_().mul_(1.0 + rep)
        graph.replay()
        torch.cuda.synchronize()
        assert _rel(out[0], _ref(xs, w, b, 'same')) < TOL and _rel(out[1], _ref(xs * 2, w, b, 'same')) < TOL, rep


# ---------------------------------------------------------------------------------------------------------------------
# round 5: weight / bias gradient of the critic's image-reading layers (wc_conv_wrw_narrow_f32)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,k", [((128, 32, 32, 3), 128, 3), ((128, 16, 16, 3), 128, 1), ((5, 12, 12, 1), 256, 3),
                                          ((3, 48, 48, 3), 128, 3), ((64, 64, 64, 3), 128, 1), ((2, 6, 10, 2), 128, 3)])
def test_narrow_input_weight_gradient_against_float64(shape, cout, k):
    """dW, db (and the forward, and dx) of a 'same' convolution on an image-like input through conv.narrow_in_conv against torch in
    float64: one pass over gy on the fp32 matrix pipe, fixed summation order (two calls give the same bits); weights in channels_last
    and in contiguous layout."""
    from wc_gan_amd import conv as C
    torch.manual_seed(11)
    N, H, W, Ci = shape
    x = (torch.randn(*shape, device='cuda') * 0.7 + 0.1).requires_grad_(True)
    for fmt in (torch.channels_last, torch.contiguous_format):
        w = (torch.randn(cout, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5).contiguous(memory_format=fmt).requires_grad_(True)
        b = (torch.randn(cout, device='cuda') * 0.1).requires_grad_(True)
        assert C.narrow_wrw_supported(x, w)
        y = C.narrow_in_conv(x, w, b)
        gy = torch.randn_like(y)
        dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
        dw2, = torch.autograd.grad(C.narrow_in_conv(x, w, b), (w,), gy)
        y64 = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
        assert dw.stride() == w.stride() and torch.equal(dw, dw2)
        assert _rel(y, y64) < 1e-5 and _rel(dx, dx64) < 1e-5
        assert _rel(dw, dw64) < 2e-6 and _rel(db, db64) < 2e-6, (fmt, _rel(dw, dw64), _rel(db, db64))


@pytest.mark.gpu
def test_critic_first_block_takes_the_narrow_weight_gradient():
    """generator.Conv2D on a 3-channel input (the critic's conv1 and shortcut): the layer's output and gradients equal MIOpen's route
    (WC_NARROW_WRW off) to fp32 summation order, with and without spectral normalisation; the data gradient too (the generator update)."""
    from wc_gan_amd import conv as C
    from wc_gan_amd.generator import Conv2D
    torch.manual_seed(12)
    for spectral in (False, True):
        for k in (3, 1):
            layer = Conv2D(3, 128, (k, k), spectral=spectral).cuda()
            x = torch.randn(16, 32, 32, 3, device='cuda', requires_grad=True)
            wts = torch.randn(16, 32, 32, 128, device='cuda')         # (random: a sum that cancels to 1e-4 of its terms would compare two fp32 rounding errors)
            state = {n: v.detach().clone() for n, v in layer.state_dict().items()}
            outs = []
            for on in (True, False):
                layer.load_state_dict(state)
                C.NARROW_WRW = on
                try:
                    y = layer(x)
                    g = torch.autograd.grad((y * wts).sum(), [x] + list(layer.parameters()))
                finally:
                    C.NARROW_WRW = True
                outs.append([y.detach()] + [t.detach() for t in g])
            for a, b in zip(*outs):
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (spectral, k)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((128, 32, 32, 256), 3), ((6, 12, 12, 128), 3), ((4, 8, 8, 128), 1)])
def test_narrow_output_weight_gradient_against_float64(shape, cout):
    """The generator's last layer (256 -> 3, generator.py:155-157): its weight gradient by the narrow-input kernel with the operands
    exchanged and the taps mirrored (conv.narrow_out_weight_gradient), and through generator._NarrowConv3x3, against float64."""
    from wc_gan_amd import conv as C
    from wc_gan_amd.generator import _NarrowConv3x3
    torch.manual_seed(13)
    N, H, W, Ci = shape
    x = (torch.randn(*shape, device='cuda') * 0.8).requires_grad_(True)
    w = (torch.randn(cout, Ci, 3, 3, device='cuda') / (Ci * 9) ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.zeros(cout, device='cuda', requires_grad=True)
    assert C.narrow_out_wrw_supported(x, w)
    y = _NarrowConv3x3.apply(x, w, b)
    gy = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
    y64 = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    dx64, dw64, db64 = torch.autograd.grad(y64, (x, w, b), gy.double())
    assert dw.stride() == w.stride()
    assert _rel(y, y64) < 1e-5 and _rel(dx, dx64) < 1e-5 and _rel(db, db64) < 1e-5
    assert _rel(dw, dw64) < 2e-6, _rel(dw, dw64)
