"""CPU check of the convolution GEOMETRIES (wc_gan_amd/conv.py: virtual grid, phases, taps, source slices) that the HIP
kernel is driven by: a literal numpy interpreter of `wc_conv_geom` must reproduce torch's convolutions -- forward AND data
gradient -- for every layer kind (Keras Conv2D 'same', Conv2D -> AveragePooling2D, UpSampling2D -> Conv2D;
generator.py:142-158, discriminator.py:41-54).  No GPU, no library load: this pins the host logic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from wc_gan_amd import conv as C


def interpret(g, x, w, k_axis, n_axis):
    """y[n, oy, ox, :] per the geometry; x (N, Hin, Win, Cin) float64, w the 4-d weight; k_axis / n_axis as conv._geoms says"""
    y = np.zeros((g.N, g.Hout, g.Wout, g.Cout))
    wt = np.moveaxis(w, (k_axis, n_axis), (0, 1))                 # (k, n, r, s)
    assert wt.shape[0] == g.Cin and wt.shape[1] == g.Cout
    for p in range(g.nphase):
        for t in range(g.ntaps):
            sl = np.zeros((g.Cin, g.Cout))
            for m in range(g.nsrc[p][t]):
                sl += wt[:, :, g.wr[p][t][m], g.ws[p][t][m]]
            sl *= g.wcoef
            for yy in range(g.H):
                iy = yy * g.in_stride + g.dy[p][t]
                if not 0 <= iy < g.Hin:
                    continue
                for xx in range(g.W):
                    ix = xx * g.in_stride + g.dx[p][t]
                    if not 0 <= ix < g.Win:
                        continue
                    y[:, yy * g.out_stride + g.off_y[p], xx * g.out_stride + g.off_x[p], :] += x[:, iy, ix, :] @ sl
    return y


def reference(kind, x, w):
    xn = x.permute(0, 3, 1, 2)
    if kind == 'same':
        y = F.conv2d(xn, w, padding=w.shape[2] // 2)
    elif kind == 'down':
        y = F.conv2d(xn, w, stride=2, padding=1)
    elif kind == 'up':
        y = F.conv_transpose2d(xn, w, stride=2, padding=1)
    elif kind == 'down3':
        y = F.avg_pool2d(F.conv2d(xn, w, padding=1), 2)
    else:
        y = F.conv2d(F.interpolate(xn, scale_factor=2, mode='nearest'), w, padding=1)
    return y.permute(0, 2, 3, 1)


CASES = [('same', 3, (5, 4, 3, 3)), ('same', 1, (5, 4, 1, 1)), ('down', 4, (5, 4, 4, 4)), ('up', 4, (4, 5, 4, 4)),
         ('down3', 3, (5, 4, 3, 3)), ('up3', 3, (5, 4, 3, 3))]


@pytest.mark.parametrize("kind,k,wshape", CASES)
@pytest.mark.parametrize("H,W", [(4, 6), (2, 2)])
def test_geometry_reproduces_the_layer_and_its_data_gradient(kind, k, wshape, H, W):
    torch.manual_seed(H * 10 + W + k)
    N, ci = 2, 4
    x = torch.randn(N, H, W, ci, dtype=torch.float64, requires_grad=True)
    w = torch.randn(*wshape, dtype=torch.float64)
    (gf, kf, nf), (gb, kb, nb) = C._geoms(kind, N, H, W, w)
    y_ref = reference(kind, x, w)
    y = interpret(gf, x.detach().numpy(), w.numpy(), kf, nf)
    assert y.shape == tuple(y_ref.shape)
    assert np.abs(y - y_ref.detach().numpy()).max() < 1e-12
    gy = torch.randn_like(y_ref)
    dx_ref, = torch.autograd.grad(y_ref, x, gy)
    dx = interpret(gb, gy.numpy(), w.numpy(), kb, nb)
    assert dx.shape == tuple(dx_ref.shape)
    assert np.abs(dx - dx_ref.numpy()).max() < 1e-12


def test_every_source_tap_is_named_by_the_forward_geometry():
    """the weight-gradient reduction writes dW tap by tap from the forward geometry's source lists: all 9 (3x3) taps appear"""
    w = torch.zeros(8, 4, 3, 3)
    for kind in ('same', 'down3', 'up3'):
        (gf, _, _), _ = C._geoms(kind, 2, 4, 4, w)
        taps = {(gf.wr[p][t][m], gf.ws[p][t][m]) for p in range(gf.nphase) for t in range(gf.ntaps) for m in range(gf.nsrc[p][t])}
        assert taps == {(r, s) for r in range(3) for s in range(3)}
        assert all(1 <= gf.nsrc[p][t] <= 4 for p in range(gf.nphase) for t in range(gf.ntaps))


def interpret_wrw(g, x, gy, wshape, k_axis, n_axis):
    """dW per the FORWARD geometry: every slice's product sum, folded (x wcoef) onto each source tap it was formed from"""
    dw = np.zeros(np.moveaxis(np.zeros(wshape), (k_axis, n_axis), (0, 1)).shape)      # (k, n, r, s)
    for p in range(g.nphase):
        for t in range(g.ntaps):
            P = np.zeros((g.Cin, g.Cout))
            for yy in range(g.H):
                iy = yy * g.in_stride + g.dy[p][t]
                if not 0 <= iy < g.Hin:
                    continue
                for xx in range(g.W):
                    ix = xx * g.in_stride + g.dx[p][t]
                    if not 0 <= ix < g.Win:
                        continue
                    P += x[:, iy, ix, :].T @ gy[:, yy * g.out_stride + g.off_y[p], xx * g.out_stride + g.off_x[p], :]
            for m in range(g.nsrc[p][t]):
                dw[:, :, g.wr[p][t][m], g.ws[p][t][m]] += g.wcoef * P
    return np.moveaxis(dw, (0, 1), (k_axis, n_axis))


@pytest.mark.parametrize("kind,k,wshape", CASES)
def test_geometry_reproduces_the_weight_gradient(kind, k, wshape):
    torch.manual_seed(k)
    N, H, W, ci = 2, 4, 6, 4
    x = torch.randn(N, H, W, ci, dtype=torch.float64)
    w = torch.randn(*wshape, dtype=torch.float64, requires_grad=True)
    (gf, kf, nf), _ = C._geoms(kind, N, H, W, w)
    y_ref = reference(kind, x, w)
    gy = torch.randn_like(y_ref)
    dw_ref, = torch.autograd.grad(y_ref, w, gy)
    dw = interpret_wrw(gf, x.numpy(), gy.numpy(), wshape, kf, nf)
    assert np.abs(dw - dw_ref.numpy()).max() < 1e-11
