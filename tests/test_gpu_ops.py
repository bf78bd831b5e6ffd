"""Op-level parity of the HIP kernels (through the C ABI) against the numpy oracle.
Tolerance: rel-L2 <= 1e-5 for fp32 ops (north_star bound is 1e-3; fp32 MFMA is an exact fmaf chain)."""
import os

import numpy as np
import pytest

from gan_heightmaps_amd._lib import tuning_env

from oracle import lp as LP
from oracle import ops as O

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.fixture(scope="module")
def gpu():
    from gan_heightmaps_amd import device
    if device.device_count() == 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    dev = device.Device(0)
    yield dev, device.Ops(dev), device
    dev.close()


CONV_CASES = [
    # N, C, H, W, K, k, s, pad
    (2, 16, 12, 12, 32, 5, 1, 2),     # dcgan-style 5x5 same
    (2, 32, 16, 16, 64, 3, 2, 1),     # unet encoder 3x3 s2
    (1, 48, 9, 11, 40, 3, 1, 1),      # odd sizes, C%16==0, ragged K
    (2, 5, 10, 10, 7, 3, 1, 1),       # slow-K path, ragged everything
    (4, 1, 32, 32, 64, 5, 1, 2),      # first layer d_conv1 (C=1)
    (2, 4, 32, 32, 64, 3, 2, 1),      # pd_conv1 (C=4, s2)
    (2, 64, 16, 16, 1, 5, 1, 2),      # g_out (K=1: direct kernel)
    (2, 32, 8, 8, 3, 3, 2, 1),        # small-R, stride 2
    (3, 32, 2, 2, 48, 2, 1, 0),       # conv9: k2 valid 2x2 -> 1x1
    (2, 128, 24, 24, 128, 3, 1, 1),   # 128-wide tiles
    (1, 16, 7, 7, 16, 3, 2, 1),       # odd input with stride 2
    (4, 1000, 1, 1, 96, 1, 1, 0),     # DenseLayer as 1x1 conv (1000 % 16 != 0)
    (4, 1000, 1, 1, 2048, 1, 1, 0),   # the generator's first layer, narrower (dense_smallp_kernel<4>: GEMV + slice sums)
    (7, 200, 1, 1, 1100, 1, 1, 0),    # dense_smallp_kernel<8>, ragged row block (1100 = 4 * 256 + 76), 12 slices
    # geometries that take the LDS-patch kernels (Wo % 16 == 0, C*k*k >= 96)
    (2, 16, 32, 32, 128, 5, 1, 2),    # 5x5 s1, ragged channel tile (16 = 3*5 + 1)
    (1, 40, 32, 48, 64, 3, 1, 1),     # 3x3 s1, rectangular, 64 filters
    (2, 24, 64, 64, 48, 3, 2, 1),     # 3x3 s2 -> 32x32
    (1, 12, 32, 32, 200, 5, 1, 2),    # ragged filter tile (200 = 128 + 72)
    (2, 32, 32, 32, 96, 3, 2, 1),     # 3x3 s2 -> 16x16 (one slab per row)
    (3, 20, 16, 16, 24, 3, 1, 1),     # narrow filter tile (24 -> BN=32)
    (2, 64, 32, 32, 128, 5, 1, 2),    # conv_patch_kernel 5x5, 128-row tile, both directions
    (1, 32, 64, 64, 64, 5, 1, 2),     # conv_patch_kernel 5x5, 64-row tile (8x32 pixel tile)
    (2, 48, 32, 64, 160, 3, 1, 1),    # conv_patch_kernel 3x3, ragged filter tile, rectangular
    (1, 64, 32, 32, 64, 3, 1, 1),     # conv_patch_kernel 3x3, 64-row tile
    (2, 32, 64, 64, 128, 3, 2, 1),    # conv_patch_kernel 3x3 stride 2, 128-row tile (-> 32x32)
    (1, 16, 64, 128, 64, 3, 2, 1),    # conv_patch_kernel 3x3 stride 2, 64-row tile, rectangular
    (2, 128, 64, 64, 40, 3, 2, 1),    # dgrad_s2_patch_kernel, 128 dx channels, ragged filters (40 % 4 == 0)
    (1, 48, 128, 64, 24, 3, 2, 1),    # dgrad_s2_patch_kernel, 64-row tile (48 channels), rectangular
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", ["big", "small"])
def test_conv_fwd_dgrad_wgrad(gpu, case, tile):
    with tuning_env(GHM_FORCE_TILE=tile):
        _check_conv(gpu, case)


# first / last layers at full resolution: <= 4 channels on one side, large maps (conv_thin.hip)
THIN_CASES = [
    # N, C, H, W, K, k, s, pad        fwd / dgrad / dgrad_t variants expected
    ((2, 1, 128, 128, 64, 5, 1, 2), ("fanout_kernel<fwd>", "fanin_s1_kernel<dgrad>", None)),   # d_conv1
    ((2, 4, 256, 256, 64, 3, 2, 1), ("fanout_kernel<fwd>", "fanin_s2_kernel<3>", None)),   # pd_conv1
    ((2, 1, 256, 256, 64, 3, 2, 1), ("fanout_kernel<fwd>", "fanin_s2_kernel<3>", None)),   # unet conv1
    ((2, 3, 256, 256, 128, 2, 2, 0), ("fanout_kernel<fwd>", "fanin_s2_kernel<2>", None)),  # final Deconv2DLayer
    ((2, 64, 128, 128, 1, 5, 1, 2), ("fanin_s1_kernel<fwd>", "fanout_kernel<dgrad>", "fanout_kernel<dgrad_t>")),   # g_out
    ((2, 1, 64, 256, 64, 5, 1, 2), ("fanout_kernel<fwd>", "fanin_s1_kernel<dgrad>", None)),   # d_conv1, rectangular
    ((2, 128, 128, 128, 3, 3, 1, 1), ("fanin_s1_kernel<fwd>", None, None)),          # 3 filters x 9 taps, 128 channels
    ((1, 64, 128, 256, 3, 3, 1, 1), (None, "fanout_kernel<dgrad>", "fanout_kernel<dgrad_t>")),   # 3 filters, 27 rows
    ((2, 3, 128, 256, 64, 3, 1, 1), ("fanout_kernel<fwd>", None, None)),             # 3 channels x 9 taps, rectangular
    ((2, 3, 128, 256, 48, 3, 1, 1), ("igemm_kernel<64,256,fwd>", None, None)),       # ragged filters: general kernel
]


@pytest.mark.parametrize("case,variants", THIN_CASES)
def test_thin_layer_kernels(gpu, case, variants):
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    for kind, want in zip((0, 1, 3), variants):
        if want is not None:
            assert ops.conv_variant(d, kind) == want
    if (K in (64, 128) or C in (64, 128)) and d.Wo >= 256:
        assert ops.conv_variant(d, 2).startswith("thin_wgrad_kernel")
    _check_conv(gpu, case)
    # and they agree with the general kernels they replace
    rng = np.random.RandomState(3)
    x = dev.tensor(rng.randn(N, C, H, W).astype(np.float32))
    dy = dev.tensor(rng.randn(N, K, d.Ho, d.Wo).astype(np.float32))
    w = dev.tensor((rng.randn(C * k * k * K) / np.sqrt(C * k * k)).astype(np.float32))
    b = dev.tensor(rng.randn(K).astype(np.float32))
    y1, y2 = dev.empty((N, K, d.Ho, d.Wo)), dev.empty((N, K, d.Ho, d.Wo))
    dx1, dx2 = dev.empty((N, C, H, W)), dev.empty((N, C, H, W))
    ops.conv2d_fwd(d, x, w, b, y1, act='tanh')
    ops.conv2d_dgrad(d, dy, w, dx1)
    with tuning_env(GHM_NO_THIN="1"):
        ops.conv2d_fwd(d, x, w, b, y2, act='tanh')
        ops.conv2d_dgrad(d, dy, w, dx2)
    assert rel(y1.numpy(), y2.numpy()) < 1e-5 and rel(dx1.numpy(), dx2.numpy()) < 1e-5


@pytest.mark.parametrize("case", [(4, 64, 32, 32, 1, 3, 2, 1),      # pd_out (p2p.py:289): one filter, 3x3 stride 2
                                  (3, 40, 4, 4, 1, 5, 1, 2),        # d_out (dcgan.py:50): one filter, 5x5 on the last 4x4 map
                                  (2, 19, 10, 6, 1, 3, 1, 1)])      # ragged channel group, rectangular
def test_one_filter_data_gradient_kernel(gpu, case):
    """K == 1: a thread owns a pixel and eight channels (smallk1_dgrad_kernel) -- the same products in the same order as the
    one-thread-per-element kernel it replaces (GHM_NO_SMALLK1=1): bit-identical, with and without the producer's
    LeakyRectify backward in the epilogue."""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.conv_variant(d, 1) == "smallk_dgrad_kernel"
    _check_conv(gpu, case)
    rng = np.random.RandomState(5)
    dy = dev.tensor(rng.randn(N, K, d.Ho, d.Wo).astype(np.float32))
    w = dev.tensor((rng.randn(C * k * k * K) / np.sqrt(C * k * k)).astype(np.float32))
    xa = dev.tensor(rng.randn(N, C, H, W).astype(np.float32))
    outs = []
    for env in ({}, {"GHM_NO_SMALLK1": "1"}):
        with tuning_env(**env):
            a, b = dev.empty((N, C, H, W)), dev.empty((N, C, H, W))
            ops.conv2d_dgrad(d, dy, w, a)
            if ops.dgrad_dact_supported(d, 'f32') == 1:          # (form 1: reads the packed weights as they are)
                ops.conv2d_dgrad_dact(d, dy, w, b, xa, 'lrelu', 0.2, 'f32')
            else:
                ops.conv2d_dgrad(d, dy, w, b)
            outs.append((a.numpy(), b.numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_scale_samples(gpu):
    """ghm_scale_samples: x[n] *= num[n] / den[n] on a strided view -- the per-sample factor of the DCGAN generator's gradient
    (step.py).  den[n] == 0 behind a dead final ReLU means the gradient is exactly 0 as well and stays 0; a NON-zero gradient
    behind a zero denominator (a head that is exactly 0 without a dead ReLU) cannot be recovered by any factor and must come out
    NaN, not silently 0.  A denominator at the bottom of the fp32 range must not make inf * 0."""
    dev, ops, D = gpu
    rng = np.random.RandomState(4)
    N, C, H, W = 6, 3, 6, 10
    big = rng.randn(N, C + 2, H, W).astype(np.float32)
    num = np.array([0.5, -0.25, 3.0, 0.7, 0.5, -0.5], np.float32)
    den = np.array([2.0, 0.125, 0.0, -1.5, 1e-42, 0.0], np.float32)   # [2]: dead sample, [4]: denormal, [5]: zero seed, live gradient
    big[4] *= np.float32(1e-40)                                         # the gradient behind a tiny seed is that small too
    big[2, 1:1 + C] = 0.0                                               # dead final ReLU: the gradient is exactly zero
    t = dev.tensor(big)
    view = t.channels(1, 1 + C)
    ops.scale_samples(view, dev.tensor(num.reshape(N, 1, 1, 1)), dev.tensor(den.reshape(N, 1, 1, 1)))
    got = t.numpy()
    want = big.astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        r = np.where(den != 0, num.astype(np.float64) / den.astype(np.float64), 0.0)
    want[:, 1:1 + C] *= r.reshape(N, 1, 1, 1)
    assert np.all(np.isfinite(got[:5]))
    assert np.array_equal(got[:, 0], big[:, 0]) and np.array_equal(got[:, -1], big[:, -1])      # outside the view: untouched
    assert np.all(got[2, 1:1 + C] == 0)
    assert np.all(np.isnan(got[5, 1:1 + C]))                            # loud, not zero
    assert np.allclose(got[:5, 1:1 + C], want[:5, 1:1 + C].astype(np.float32), rtol=1e-6, atol=0)


def _check_conv(gpu, case):
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    y_ref = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, dW_ref, db_ref = O.conv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), dy.astype(np.float64), s, pad)

    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    xd, wd, bd = dev.tensor(x), dev.tensor(D.pack_conv_w(Wt).reshape(1, -1, 1, 1)), dev.tensor(b)
    yd = dev.empty(y_ref.shape)
    ops.conv2d_fwd(d, xd, wd, bd, yd)
    assert rel(yd.numpy(), y_ref) < TOL
    # fused epilogue
    ops.conv2d_fwd(d, xd, wd, bd, yd, act='lrelu', alpha=0.2)
    assert rel(yd.numpy(), O.lrelu_fwd(y_ref, 0.2)) < TOL
    # the other activations of the epilogue (relu shares the piecewise-linear fast path, tanh / sigmoid take the
    # element-by-element one)
    for act, f in (('relu', lambda v: np.maximum(v, 0.0)), ('tanh', np.tanh), ('sigmoid', lambda v: 1.0 / (1.0 + np.exp(-v)))):
        ops.conv2d_fwd(d, xd, wd, bd, yd, act=act)
        assert rel(yd.numpy(), f(y_ref)) < TOL, act
    # dgrad (+ accumulate)
    dyd, dxd = dev.tensor(dy), dev.empty(x.shape)
    ops.conv2d_dgrad(d, dyd, wd, dxd)
    assert rel(dxd.numpy(), dx_ref) < TOL
    ops.conv2d_dgrad(d, dyd, wd, dxd, accumulate=True)
    assert rel(dxd.numpy(), 2 * dx_ref) < TOL
    if ops.dgrad_t_supported(d):
        # data gradient as a forward-form conv on the transposed packed weights
        wtd = dev.zeros((1, C * k * k * K, 1, 1))
        ops.transpose_weights(d, wd, wtd)
        wp_host = D.pack_conv_w(Wt)                                   # [C, T, K]
        assert np.array_equal(wtd.numpy().ravel(), np.ascontiguousarray(wp_host[:, ::-1, :].transpose(2, 1, 0)).ravel())
        ops.conv2d_dgrad_t(d, dyd, wtd, dxd)
        assert rel(dxd.numpy(), dx_ref) < TOL
        ops.conv2d_dgrad_t(d, dyd, wtd, dxd, accumulate=True)
        assert rel(dxd.numpy(), 2 * dx_ref) < TOL
    # wgrad in packed layout
    dwd = dev.zeros((1, C * k * k * K, 1, 1))
    ws = dev.alloc(ops.wgrad_workspace(d))
    ops.conv2d_wgrad(d, xd, dyd, dwd, ws)
    dW = D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k)
    assert rel(dW, dW_ref) < TOL
    dbd = dev.empty((1, K, 1, 1))
    ops.channel_sum(dyd, dbd)
    assert rel(dbd.numpy().ravel(), db_ref) < TOL


def test_wgrad_split_k_large(gpu):
    """enough pixels to force split-K partials + reduce, and accumulate on top"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = 2, 16, 96, 96, 32, 3, 1, 1
    rng = np.random.RandomState(1)
    x = rng.randn(N, C, H, W).astype(np.float32)
    dy = rng.randn(N, K, H, W).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert "splits=1" not in ops.conv_variant(d, 2)
    ref = O.corr2d_bwd_weight(x.astype(np.float64), dy.astype(np.float64), s, pad, k, k)   # [K,C,a,b] unflipped
    ref_packed = ref.transpose(1, 2, 3, 0).reshape(-1)
    dwd = dev.zeros((1, C * k * k * K, 1, 1))
    ws = dev.alloc(ops.wgrad_workspace(d))
    xd, dyd = dev.tensor(x), dev.tensor(dy)
    ops.conv2d_wgrad(d, xd, dyd, dwd, ws)
    assert rel(dwd.numpy().ravel(), ref_packed) < TOL
    ops.conv2d_wgrad(d, xd, dyd, dwd, ws, accumulate=True)
    assert rel(dwd.numpy().ravel(), 2 * ref_packed) < TOL


DECONV_CASES = [(2, 48, 1, 1, 32, 2, 1), (2, 32, 8, 8, 3, 2, 2), (1, 16, 5, 5, 24, 2, 2), (2, 16, 4, 4, 16, 3, 2)]


@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv_fwd_bwd(gpu, case):
    """Deconv2DLayer = adjoint of the true convolution: forward runs the dgrad form, backward the forward form."""
    dev, ops, D = gpu
    N, Cin, h, w, Cout, k, s = case
    rng = np.random.RandomState(7)
    x = rng.randn(N, Cin, h, w).astype(np.float32)
    Wt = (rng.randn(Cin, Cout, k, k) / np.sqrt(Cin)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    y_ref = O.deconv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s)
    Hh, Ww = y_ref.shape[2:]
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, dW_ref, db_ref = O.deconv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), dy.astype(np.float64), s)
    # descriptor of the forward conv this layer is the adjoint of: input = deconv output
    d = D.conv_desc(N, Cout, Hh, Ww, Cin, k, k, s, 0)
    assert (d.Ho, d.Wo) == (h, w)
    xd, wd, bd = dev.tensor(x), dev.tensor(D.pack_conv_w(Wt).reshape(1, -1, 1, 1)), dev.tensor(b)
    yd = dev.empty(y_ref.shape)
    ops.conv2d_dgrad(d, xd, wd, yd, bias=bd, act='tanh')
    assert rel(yd.numpy(), np.tanh(y_ref)) < TOL
    dyd, dxd = dev.tensor(dy), dev.empty(x.shape)
    ops.conv2d_fwd(d, dyd, wd, None, dxd)
    assert rel(dxd.numpy(), dx_ref) < TOL
    dwd = dev.zeros((1, Cout * k * k * Cin, 1, 1))
    ws = dev.alloc(ops.wgrad_workspace(d))
    ops.conv2d_wgrad(d, dyd, xd, dwd, ws)
    assert rel(D.unpack_conv_w(dwd.numpy().ravel(), Cin, Cout, k, k), dW_ref) < TOL


def test_conv_on_channel_slice_views(gpu):
    """ConcatLayer(axis=1) without copies: conv reads / writes channel slices of wider buffers."""
    dev, ops, D = gpu
    rng = np.random.RandomState(3)
    N, C, H, K = 2, 16, 8, 32
    big_in = rng.randn(N, C + 5, H, H).astype(np.float32)
    Wt = rng.randn(K, C, 3, 3).astype(np.float32) * 0.1
    b = rng.randn(K).astype(np.float32)
    tin = dev.tensor(big_in)
    tout = dev.zeros((N, K + 3, H, H))
    xin, yout = tin.channels(5, 5 + C), tout.channels(3, 3 + K)
    d = D.conv_desc(N, C, H, H, K, 3, 3, 1, 1, xin.nstride, yout.nstride)
    ops.conv2d_fwd(d, xin, dev.tensor(D.pack_conv_w(Wt).reshape(1, -1, 1, 1)), dev.tensor(b), yout)
    ref = O.conv2d_fwd(big_in[:, 5:].astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), 1, 1)
    got = tout.numpy()
    assert rel(got[:, 3:], ref) < TOL and np.all(got[:, :3] == 0)


# (N*H*W <= 16384 per channel: the one-launch-per-layer kernels; above: partial / final / apply; "big" forces the latter)
@pytest.mark.parametrize("shape", [(4, 24, 1, 1), (4, 8, 16, 16), (3, 5, 7, 9), (2, 64, 64, 64), (4, 40, 64, 64),
                                   (2, 6, 96, 128), (2, 3, 129, 67)])
@pytest.mark.parametrize("act,alpha", [('lrelu', 0.2), ('linear', 0.0)])
@pytest.mark.parametrize("path", ["auto", "big"])
def test_batchnorm_fwd_bwd(gpu, shape, act, alpha, path):
    with tuning_env(**({"GHM_NO_BN_SMALL": "1"} if path == "big" else {})):
        _check_batchnorm(gpu, shape, act, alpha)


def _check_batchnorm(gpu, shape, act, alpha):
    dev, ops, D = gpu
    rng = np.random.RandomState(5)
    N, C, H, W = shape
    x = (rng.randn(*shape) * 2 + 1).astype(np.float32)
    beta, gamma = rng.randn(C).astype(np.float32), rng.rand(C).astype(np.float32) + 0.5
    rm, ri = rng.randn(C).astype(np.float32), rng.rand(C).astype(np.float32) + 0.5
    x64 = x.astype(np.float64)
    z, mu, inv = O.bn_train_fwd(x64, beta.astype(np.float64), gamma.astype(np.float64))
    y_ref = O.lrelu_fwd(z, alpha) if act == 'lrelu' else z
    dout = rng.randn(*shape).astype(np.float32)
    dz = O.lrelu_vjp(z, alpha, dout.astype(np.float64)) if act == 'lrelu' else dout.astype(np.float64)
    dx_ref, dbeta_ref, dgamma_ref = O.bn_train_vjp(x64, gamma.astype(np.float64), mu, inv, dz)
    rm_ref, ri_ref = O.bn_running_update(rm.astype(np.float64), ri.astype(np.float64), mu, inv)

    ws = dev.alloc(ops.bn_workspace(C))
    xd, yd = dev.tensor(x), dev.empty(shape)
    md, ivd, rmd, rid = dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1)), dev.tensor(rm), dev.tensor(ri)
    gd, bd = dev.tensor(gamma), dev.tensor(beta)
    ops.bn_stats(xd, md, ivd, ws, rmd, rid)
    assert rel(md.numpy().ravel(), mu) < TOL and rel(ivd.numpy().ravel(), inv) < TOL
    assert rel(rmd.numpy().ravel(), rm_ref) < TOL and rel(rid.numpy().ravel(), ri_ref) < TOL
    ops.bn_apply(xd, yd, md, ivd, gd, bd, act, alpha)
    assert rel(yd.numpy(), y_ref) < TOL
    dxd, dgd, dbd = dev.empty(shape), dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1))
    ops.bn_backward(dev.tensor(dout), yd, xd, dxd, md, ivd, gd, dgd, dbd, ws, act, alpha)
    assert rel(dbd.numpy().ravel(), dbeta_ref) < 1e-4
    assert rel(dgd.numpy().ravel(), dgamma_ref) < 1e-4
    assert rel(dxd.numpy(), dx_ref) < 1e-4
    # deterministic path = apply with running stats
    ops.bn_apply(xd, yd, rmd, rid, gd, bd, 'linear', 0.0)
    assert rel(yd.numpy(), O.bn_infer_fwd(x64, beta, gamma, rm_ref, ri_ref)) < TOL
    # the training forward as ONE entry point (what the step issues): same results as stats + apply, bit for bit
    md2, ivd2, rmd2, rid2, yd2 = dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1)), dev.tensor(rm), dev.tensor(ri), dev.empty(shape)
    ops.bn_forward(xd, yd2, md2, ivd2, gd, bd, ws, rmd2, rid2, act=act, alpha=alpha)
    ops.bn_apply(xd, yd, md, ivd, gd, bd, act, alpha)
    assert np.array_equal(md2.numpy(), md.numpy()) and np.array_equal(ivd2.numpy(), ivd.numpy())
    assert np.array_equal(rmd2.numpy(), rmd.numpy()) and np.array_equal(rid2.numpy(), rid.numpy())
    assert np.array_equal(yd2.numpy(), yd.numpy())
    # accumulate on top of existing parameter gradients
    ops.bn_backward(dev.tensor(dout), yd, xd, dxd, md, ivd, gd, dgd, dbd, ws, act, alpha, accumulate=True)
    assert rel(dbd.numpy().ravel(), 2 * dbeta_ref) < 1e-4 and rel(dgd.numpy().ravel(), 2 * dgamma_ref) < 1e-4
    # the form the step issues: no y -- the layer output is recomputed from x with the forward pass's own expression,
    # so every result is bit-identical to the form that reads y (yd holds bn_apply's output for these statistics)
    dx_y = dev.empty(shape)
    dg_y, db_y, dg_x, db_x = (dev.empty((1, C, 1, 1)) for _ in range(4))
    ops.bn_backward(dev.tensor(dout), yd, xd, dx_y, md, ivd, gd, dg_y, db_y, ws, act, alpha)
    ops.bn_backward_x(dev.tensor(dout), xd, dxd, md, ivd, gd, bd, dg_x, db_x, ws, act, alpha)
    assert np.array_equal(dxd.numpy(), dx_y.numpy())
    assert np.array_equal(dg_x.numpy(), dg_y.numpy()) and np.array_equal(db_x.numpy(), db_y.numpy())
    # sample-strided views (a channel slice of a wider buffer, as the concat-in-place layers use)
    wide = dev.tensor(np.concatenate([x, rng.randn(*shape).astype(np.float32)], axis=1))
    view = wide.channels(0, C) if hasattr(wide, 'channels') else None
    if view is not None:
        ops.bn_forward(view, yd2, md2, ivd2, gd, bd, ws, act=act, alpha=alpha)
        assert np.array_equal(yd2.numpy(), yd.numpy())


def test_bn_bottleneck_batch4_single_pixel(gpu):
    """the [4,512,1,1] U-Net bottleneck: statistics over 4 values per channel"""
    dev, ops, D = gpu
    rng = np.random.RandomState(11)
    x = rng.randn(4, 512, 1, 1).astype(np.float32)
    z, mu, inv = O.bn_train_fwd(x.astype(np.float64), np.zeros(512), np.ones(512))
    ws = dev.alloc(ops.bn_workspace(512))
    md, ivd = dev.empty((1, 512, 1, 1)), dev.empty((1, 512, 1, 1))
    ops.bn_stats(dev.tensor(x), md, ivd, ws)
    assert rel(ivd.numpy().ravel(), inv) < TOL


def test_pool_upsample_act(gpu):
    dev, ops, D = gpu
    rng = np.random.RandomState(9)
    x = rng.randn(2, 6, 16, 12).astype(np.float32)
    xd = dev.tensor(x)
    # maxpool + fused lrelu backward
    a = O.lrelu_fwd(x.astype(np.float64), 0.2)
    ad = dev.tensor(a)
    pd_ = dev.empty((2, 6, 8, 6))
    ops.maxpool2_fwd(ad, pd_)
    p_ref = O.maxpool_fwd(a, 2)
    assert rel(pd_.numpy(), p_ref) < TOL
    g = rng.randn(*p_ref.shape).astype(np.float32)
    dxd = dev.empty(x.shape)
    ops.maxpool2_bwd(ad, pd_, dev.tensor(g), dxd, 'lrelu', 0.2)
    ref = O.lrelu_vjp(x.astype(np.float64), 0.2, O.maxpool_vjp(a.astype(np.float32).astype(np.float64),
                                                               p_ref.astype(np.float32).astype(np.float64), g, 2))
    assert rel(dxd.numpy(), ref) < TOL
    # avgpool
    yd = dev.empty((2, 6, 4, 3))
    ops.avgpool_fwd(xd, yd, 4)
    assert rel(yd.numpy(), O.avgpool_fwd(x, 4)) < TOL
    g = rng.randn(2, 6, 4, 3).astype(np.float32)
    ops.avgpool_bwd(dev.tensor(g), dxd, 4)
    assert rel(dxd.numpy(), O.avgpool_vjp(x.shape, g, 4)) < TOL
    # nearest / bilinear up + adjoints (+accumulate)
    ud = dev.empty((2, 6, 32, 24))
    ops.upsample_nearest2_fwd(xd, ud)
    assert np.array_equal(ud.numpy(), O.upscale_nearest_fwd(x))
    g = rng.randn(2, 6, 32, 24).astype(np.float32)
    gd = dev.tensor(g)
    ops.upsample_nearest2_bwd(gd, dxd)
    assert rel(dxd.numpy(), O.upscale_nearest_vjp(g.astype(np.float64))) < TOL
    ops.upsample_bilinear2_fwd(xd, ud)
    assert rel(ud.numpy(), O.bilinear_up2_fwd(x)) < TOL
    ops.upsample_bilinear2_bwd(gd, dxd)
    assert rel(dxd.numpy(), O.bilinear_up2_vjp(g.astype(np.float64))) < TOL
    ops.upsample_bilinear2_bwd(gd, dxd, accumulate=True)
    assert rel(dxd.numpy(), 2 * O.bilinear_up2_vjp(g.astype(np.float64))) < TOL
    # 1x1 and 2x2 bilinear edge cases (U-Net 2x2 -> 4x4)
    for hw in [(1, 1), (2, 2), (1, 3), (3, 4), (8, 2)]:
        xs = rng.randn(2, 3, *hw).astype(np.float32)
        us = dev.empty((2, 3, 2 * hw[0], 2 * hw[1]))
        ops.upsample_bilinear2_fwd(dev.tensor(xs), us)
        assert rel(us.numpy(), O.bilinear_theano_literal(xs.astype(np.float64))) < TOL
        gs = rng.randn(2, 3, 2 * hw[0], 2 * hw[1]).astype(np.float32)      # adjoint: even widths take the two-column kernel
        dxs = dev.empty(xs.shape)
        ops.upsample_bilinear2_bwd(dev.tensor(gs), dxs)
        assert rel(dxs.numpy(), O.bilinear_up2_vjp(gs.astype(np.float64))) < TOL
    # activations fwd/bwd
    for act, alpha, f, df in [('sigmoid', 0, O.sigmoid_fwd, O.sigmoid_vjp_from_out),
                              ('tanh', 0, O.tanh_fwd, O.tanh_vjp_from_out)]:
        yd2 = dev.empty(x.shape)
        ops.act_fwd(xd, yd2, act, alpha)
        y_ref = f(x.astype(np.float64))
        assert rel(yd2.numpy(), y_ref) < TOL
        g = rng.randn(*x.shape).astype(np.float32)
        ops.act_bwd(dev.tensor(g), yd2, dxd, act, alpha)
        assert rel(dxd.numpy(), df(y_ref, g)) < 1e-4


def test_losses_and_optimizers(gpu):
    dev, ops, D = gpu
    rng = np.random.RandomState(2)
    d = rng.randn(8, 1, 16, 16).astype(np.float32)
    dd, gd, ld = dev.tensor(d), dev.empty(d.shape), dev.zeros((1, 1, 1, 1))
    ops.lsgan_loss(dd, 1.0, ld, gd, 1.0)
    loss, grad = O.squared_error_mean(d.astype(np.float64), 1.0)
    assert abs(ld.numpy().item() - loss) < 1e-5 * abs(loss) and rel(gd.numpy(), grad) < TOL
    ops.lsgan_loss(dd, 0.0, ld, None, 1.0, accumulate_loss=True)
    assert abs(ld.numpy().item() - (loss + O.squared_error_mean(d.astype(np.float64), 0.0)[0])) < 1e-4
    p = rng.rand(4, 1, 1, 1).astype(np.float32) * 0.8 + 0.1
    pd_, gp = dev.tensor(p), dev.empty(p.shape)
    ops.bce_loss(pd_, 1.0, ld, gp)
    loss, grad = O.bce_mean(p.astype(np.float64), 1.0)
    assert abs(ld.numpy().item() - loss) < 1e-5 and rel(gp.numpy(), grad) < TOL
    a, b = rng.randn(2, 3, 16, 16).astype(np.float32), rng.randn(2, 3, 16, 16).astype(np.float32)
    ad, bd_, ga = dev.tensor(a), dev.tensor(b), dev.zeros(a.shape)
    ops.recon_loss(ad, bd_, ld, ga, 100.0)
    loss, grad = O.l1_mean(a.astype(np.float64), b.astype(np.float64))
    assert abs(ld.numpy().item() - loss) < 1e-5 and rel(ga.numpy(), 100 * grad) < TOL
    ops.recon_loss(ad, bd_, ld, ga, 1.0, l2=True, accumulate_grad=True)
    l2, g2 = O.l2_mean(a.astype(np.float64), b.astype(np.float64))
    assert abs(ld.numpy().item() - l2) < 1e-5 and rel(ga.numpy(), 100 * grad + g2) < TOL
    # optimisers on a flat buffer with a ragged tail
    n = 1003
    pp, gg = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    acc = np.abs(rng.randn(n)).astype(np.float32)
    pd2, gd2, ad2 = dev.tensor(pp), dev.tensor(gg), dev.tensor(acc)
    hyper = dev.tensor(np.array([1e-2, 0.0], np.float32))
    ops.rmsprop(pd2, gd2, ad2, n, hyper)
    p_ref, a_ref = O.rmsprop_step(pp.astype(np.float64), gg.astype(np.float64), acc.astype(np.float64), 1e-2)
    assert rel(pd2.numpy().ravel(), p_ref) < TOL and rel(ad2.numpy().ravel(), a_ref) < TOL
    m, v = np.zeros(n), np.zeros(n)
    md, vd, pd3 = dev.zeros((1, n, 1, 1)), dev.zeros((1, n, 1, 1)), dev.tensor(pp)
    pcur, t = pp.astype(np.float64), 0
    for _ in range(3):
        ops.adam(pd3, gd2, md, vd, n, hyper)
        ops.adam_tick(hyper)
        pcur, m, v, t = O.adam_step(pcur, gg.astype(np.float64), m, v, t, 1e-2)
    assert rel(pd3.numpy().ravel(), pcur) < TOL
    assert hyper.numpy().ravel()[1] == 3.0


def test_graph_capture_replay(gpu):
    dev, ops, D = gpu
    x = np.arange(64, dtype=np.float32)
    xd, yd = dev.tensor(x), dev.zeros((1, 64, 1, 1))
    dev.capture_begin()
    ops.axpby(2.0, xd, 1.0, yd, 64)
    g = dev.capture_end()
    for _ in range(3):
        dev.graph_launch(g)
    dev.sync()
    assert np.allclose(yd.numpy().ravel(), 6 * x)
    dev.graph_destroy(g)


def test_rccl_single_rank_allreduce(gpu):
    """world=1 communicator: exercises dlopen(librccl), ncclCommInitRank and ncclAllReduce on the ctx stream."""
    import ctypes as C
    from gan_heightmaps_amd._lib import call
    dev, ops, D = gpu
    uid = (C.c_uint8 * 128)()
    call("ghm_comm_unique_id", C.byref(uid))
    call("ghm_comm_init", dev.h, 0, 1, C.byref(uid))
    x = np.arange(1000, dtype=np.float32)
    xd = dev.tensor(x)
    ops.allreduce_sum(xd, 1000)
    ops.allreduce_max(xd, 1000)
    dev.sync()
    assert np.array_equal(xd.numpy().ravel(), x)
    call("ghm_comm_destroy", dev.h)


@pytest.mark.parametrize("case", [(2, 64, 8, 8, 96, 3, 1, 1), (4, 256, 4, 4, 128, 5, 1, 2), (2, 48, 9, 9, 40, 3, 2, 1)])
@pytest.mark.parametrize("splits", ["3", "1000"])
def test_conv_split_k(gpu, case, splits):
    """few output tiles + long reduction: K is split across blocks, partials reduced with bias/act/accumulate"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(42)
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    y_ref = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, _, _ = O.conv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), dy.astype(np.float64), s, pad)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    xd, wd, bd = dev.tensor(x), dev.tensor(D.pack_conv_w(Wt).reshape(1, -1, 1, 1)), dev.tensor(b)
    yd, dyd, dxd = dev.empty(y_ref.shape), dev.tensor(dy), dev.zeros(x.shape)
    with tuning_env(GHM_FORCE_SPLITK=splits):
        ops.conv2d_fwd(d, xd, wd, bd, yd, act='lrelu', alpha=0.01)
        assert rel(yd.numpy(), O.lrelu_fwd(y_ref, 0.01)) < TOL
        ops.conv2d_dgrad(d, dyd, wd, dxd)
        ops.conv2d_dgrad(d, dyd, wd, dxd, accumulate=True)
        assert rel(dxd.numpy(), 2 * dx_ref) < TOL
    # the opt-in form that folds the reduction into the producer (last block to arrive at a tile sums the slices in
    # fixed order): bit-identical to the separate reduction launch, launch after launch (the counters reset themselves)
    y_sep, dx_sep = yd.numpy(), dxd.numpy()
    with tuning_env(GHM_FORCE_SPLITK=splits, GHM_SPLITK_FOLD="1"):
        for _ in range(3):
            ops.conv2d_fwd(d, xd, wd, bd, yd, act='lrelu', alpha=0.01)
            ops.conv2d_dgrad(d, dyd, wd, dxd)
            ops.conv2d_dgrad(d, dyd, wd, dxd, accumulate=True)
            assert np.array_equal(yd.numpy(), y_sep) and np.array_equal(dxd.numpy(), dx_sep)


@pytest.mark.parametrize("case", [(2, 16, 128, 128, 1, 5, 1, 2), (1, 8, 192, 192, 3, 3, 1, 1)])
def test_taps_as_rows_path(gpu, case):
    """stride-1 convs with <=4 channels on one side and many pixels (g_out, d_conv1 dgrad): 1x1 MFMA GEMM with
    (tap, r) rows + shift-add; must agree with the oracle AND with the direct kernels it replaces"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(17)
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.conv_variant(d, 0).startswith("taps_as_rows") and ops.conv_variant(d, 2).startswith("taps_as_rows")
    y_ref = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, dW_ref, _ = O.conv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), dy.astype(np.float64), s, pad)
    xd, wd, bd = dev.tensor(x), dev.tensor(D.pack_conv_w(Wt).ravel()), dev.tensor(b)
    yd = dev.empty(y_ref.shape)
    ops.conv2d_fwd(d, xd, wd, bd, yd, act='sigmoid')
    assert rel(yd.numpy(), O.sigmoid_fwd(y_ref)) < TOL
    dyd = dev.tensor(dy)
    dwd = dev.zeros((1, C * k * k * K, 1, 1))
    ops.conv2d_wgrad(d, xd, dyd, dwd, dev.alloc(ops.wgrad_workspace(d)))
    assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), dW_ref) < TOL
    # the transposed role: a conv with few INPUT channels, data gradient
    d2 = D.conv_desc(N, K, H, W, C, k, k, s, pad)
    assert ops.conv_variant(d2, 1).startswith("taps_as_rows")
    W2 = (rng.randn(C, K, k, k) / np.sqrt(K * k * k)).astype(np.float32)
    g2 = rng.randn(N, C, H, W).astype(np.float32)
    x2 = rng.randn(N, K, H, W).astype(np.float32)
    dx2_ref, _, _ = O.conv2d_vjp(x2.astype(np.float64), W2.astype(np.float64), g2.astype(np.float64), s, pad)
    dx2 = dev.zeros((N, K, H, W))
    ops.conv2d_dgrad(d2, dev.tensor(g2), dev.tensor(D.pack_conv_w(W2).ravel()), dx2)
    assert rel(dx2.numpy(), dx2_ref) < TOL
    ops.conv2d_dgrad(d2, dev.tensor(g2), dev.tensor(D.pack_conv_w(W2).ravel()), dx2, accumulate=True)
    assert rel(dx2.numpy(), 2 * dx2_ref) < TOL


@pytest.mark.parametrize("case", [(2, 16, 8, 8, 32), (1, 8, 16, 32, 4), (2, 64, 32, 32, 64), (3, 5, 4, 6, 1)])
def test_upscale_conv5_collapsed_to_four_3x3(gpu, case):
    """Upscale2DLayer(2) -> Conv2DLayer(5x5, 'same') (dcgan.py:22-31) evaluated as a 3x3 conv with 4K filters on the
    low-res input + parity interleave; forward, data gradient and weight gradient against the oracle's literal
    upscale + 5x5 convolution"""
    dev, ops, D = gpu
    N, C, H, W, K = case
    rng = np.random.RandomState(sum(case))
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, 5, 5) / np.sqrt(C * 25)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    xu = O.upscale_nearest_fwd(x.astype(np.float64), 2)
    y_ref = O.conv2d_fwd(xu, Wt.astype(np.float64), b.astype(np.float64), 1, 2)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dxu, dW_ref, db_ref = O.conv2d_vjp(xu, Wt.astype(np.float64), dy.astype(np.float64), 1, 2)
    dx_ref = O.upscale_nearest_vjp(dxu, 2)

    wp5 = dev.tensor(D.pack_conv_w(Wt).ravel())
    bd = dev.tensor(b)
    wpc, b4 = dev.empty((1, C * 9 * 4 * K, 1, 1)), dev.empty((1, 4 * K, 1, 1))
    ops.upconv_collapse_weights(wp5, bd, wpc, b4, C, K)
    assert np.array_equal(b4.numpy().ravel(), np.tile(b, 4))
    d = D.conv_desc(N, C, H, W, 4 * K, 3, 3, 1, 1)
    xd = dev.tensor(x)
    pp = dev.empty((4 * N, K, H, W))
    ops.conv2d_fwd(d, xd, wpc, b4, pp.reshape((N, 4 * K, H, W)))
    hi = dev.empty((N, K, 2 * H, 2 * W))
    ops.pp_to_hi(pp, hi)
    assert rel(hi.numpy(), y_ref) < TOL
    # backward: dy (hi) -> pp -> 3x3 data gradient (sums the four parities) and collapsed weight gradient -> 5x5
    dyd = dev.tensor(dy)
    dpp = dev.empty((4 * N, K, H, W))
    ops.hi_to_pp(dyd, dpp)
    back = dev.empty((N, K, 2 * H, 2 * W))
    ops.pp_to_hi(dpp, back)
    assert np.array_equal(back.numpy(), dy)                      # the two permutations are inverses
    dxd = dev.empty((N, C, H, W))
    ops.conv2d_dgrad(d, dpp.reshape((N, 4 * K, H, W)), wpc, dxd)
    assert rel(dxd.numpy(), dx_ref) < TOL
    dwc = dev.zeros((1, C * 9 * 4 * K, 1, 1))
    ops.conv2d_wgrad(d, xd, dpp.reshape((N, 4 * K, H, W)), dwc, dev.alloc(ops.wgrad_workspace(d)))
    dw5 = dev.zeros((1, C * 25 * K, 1, 1))
    ops.upconv_expand_wgrad(dwc, dw5, C, K)
    assert rel(D.unpack_conv_w(dw5.numpy().ravel(), K, C, 5, 5), dW_ref) < TOL
    ops.upconv_expand_wgrad(dwc, dw5, C, K, accumulate=True)
    assert rel(D.unpack_conv_w(dw5.numpy().ravel(), K, C, 5, 5), 2 * dW_ref) < TOL
    # bias gradient = per-channel sum over the parity-planar tensor
    dbd = dev.empty((1, K, 1, 1))
    ops.channel_sum(dpp, dbd)
    assert rel(dbd.numpy().ravel(), db_ref) < TOL


@pytest.mark.parametrize("case", [(2, 32, 32, 5, 7), (1, 32, 64, 2, 2), (2, 64, 32, 8, 16), (1, 96, 32, 16, 16), (3, 32, 32, 2, 9),
                                  (4, 64, 64, 32, 32)])
def test_bilinear_conv3_collapsed_onto_the_coarse_grid(gpu, case):
    """BilinearUpsample2DLayer(2) -> Conv2DLayer(3x3, 'same') (p2p.py:204-267, layers.py:13-26) without the up-sampled tensor:
    the collapsed 3x3 convolution with 4K filters on the coarse input (ghm_upconv_collapse_batched mode 1) + the frame the
    zero extension leaves out (ghm_blconv_frame_*): forward, data gradient and weight gradient against the oracle's literal
    bilinear up-sampling + convolution -- odd sizes, 2 x 2 maps, rectangular maps, both borders."""
    dev, ops, D = gpu
    N, C, K, n1, n2 = case
    assert ops.blconv_supported(N, C, K, n1, n2)
    rng = np.random.RandomState(sum(case))
    x = rng.randn(N, C, n1, n2).astype(np.float32)
    Wt = (rng.randn(K, C, 3, 3) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    xu = O.bilinear_up2_fwd(x.astype(np.float64))
    y_ref = O.conv2d_fwd(xu, Wt.astype(np.float64), b.astype(np.float64), 1, 1)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dxu, dW_ref, _ = O.conv2d_vjp(xu, Wt.astype(np.float64), dy.astype(np.float64), 1, 1)
    dx_ref = O.bilinear_up2_vjp(dxu)

    wp3 = dev.tensor(D.pack_conv_w(Wt).ravel())
    bd = dev.tensor(b)
    wpc, b4 = dev.empty((1, C * 9 * 4 * K, 1, 1)), dev.empty((1, 4 * K, 1, 1))
    ops.upconv_collapse_batched(ops.collapse_table([(wp3, bd, wpc, b4, C, K, 1)]))
    assert np.array_equal(b4.numpy().ravel(), np.tile(b, 4))
    # the collapsed taps against the oracle's: [c][rs][pq][k]
    Wc = O.bilinear_conv_collapse(O._flip(Wt.astype(np.float64)))
    want = np.stack([w.reshape(K, C, 9) for w in Wc], 0).transpose(2, 3, 0, 1)       # [c][rs][pq][k]
    assert rel(wpc.numpy().ravel(), want.ravel()) < 1e-6
    d = D.conv_desc(N, C, n1, n2, 4 * K, 3, 3, 1, 1)
    xd = dev.tensor(x)
    pp = dev.empty((4 * N, K, n1, n2))
    y4 = pp.reshape((N, 4 * K, n1, n2))
    ops.conv2d_fwd(d, xd, wpc, b4, y4)
    main = O.bilinear_conv_main(x.astype(np.float64), Wt.astype(np.float64)) + np.tile(b, 4).reshape(1, 4, K, 1, 1)
    assert rel(y4.numpy().reshape(N, 4, K, n1, n2), main) < TOL
    nfl, ndyl = ops.blconv_frame_sizes(N, C, K, n1, n2)
    FL, DYL = dev.empty((1, nfl, 1, 1)), dev.empty((1, ndyl, 1, 1))
    ops.blconv_frame_fwd(xd, wp3, y4, K, FL)
    hi = dev.empty((N, K, 2 * n1, 2 * n2))
    ops.pp_to_hi(pp, hi)
    assert rel(hi.numpy(), y_ref) < TOL
    # backward
    dpp = dev.empty((4 * N, K, n1, n2))
    ops.hi_to_pp(dev.tensor(dy), dpp)
    g4 = dpp.reshape((N, 4 * K, n1, n2))
    dxd = dev.empty((N, C, n1, n2))
    ops.conv2d_dgrad(d, g4, wpc, dxd)
    ops.blconv_frame_gather(g4, C, K, DYL)
    ops.blconv_frame_dgrad(DYL, wp3, dxd, K)
    assert rel(dxd.numpy(), dx_ref) < TOL
    dwc = dev.zeros((1, C * 9 * 4 * K, 1, 1))
    ops.conv2d_wgrad(d, xd, g4, dwc, dev.alloc(ops.wgrad_workspace(d)))
    dw3 = dev.zeros((1, C * 9 * K, 1, 1))
    ops.upconv_expand_batched(ops.expand_table([(dwc, dw3, C, K, 1)]))
    ops.blconv_frame_wgrad(DYL, FL, dw3, N, C, K, n1, n2)
    assert rel(D.unpack_conv_w(dw3.numpy().ravel(), K, C, 3, 3), dW_ref) < TOL
    # into a wider destination (the decoder's output is a channel slice of a ConcatLayer buffer): dx with a sample stride
    wide = dev.zeros((N, C + 32, n1, n2))
    view = wide.channels(32, 32 + C)
    ops.conv2d_dgrad(D.conv_desc(N, C, n1, n2, 4 * K, 3, 3, 1, 1, view.nstride, g4.nstride), g4, wpc, view)
    ops.blconv_frame_dgrad(DYL, wp3, view, K)
    got = wide.numpy()
    assert rel(got[:, 32:], dx_ref) < TOL and not got[:, :32].any()


def test_dropout_hash_mask_and_backward(gpu):
    """DropoutLayer(p): the device mask is bit-identical to oracle.ops.dropout_mask for the same (key, step); the
    backward is the same call on dy; a tick of the device counter changes the mask"""
    dev, ops, D = gpu
    rng = np.random.RandomState(0)
    x = rng.randn(3, 5, 7, 9).astype(np.float32)
    counter = dev.zeros((1, 1, 1, 1))
    xd, yd = dev.tensor(x), dev.empty(x.shape)
    for step, p, key in [(0, 0.5, 0x1234ABCD), (1, 0.5, 0x1234ABCD), (2, 0.25, 7)]:
        ops.dropout(xd, yd, p, key, counter)
        m = O.dropout_mask(x.shape, p, key, step)
        got = yd.numpy()
        assert np.array_equal(got != 0, m & (x != 0))
        assert np.allclose(got, x * m / (1 - p), rtol=1e-6, atol=0)
        assert abs(m.mean() - (1 - p)) < 0.05
        ops.counter_tick(counter)
    # strided destination (a channel slice of a wider buffer) and the backward on a gradient
    wide = dev.zeros((3, 8, 7, 9))
    ops.dropout(xd, wide.channels(2, 7), 0.5, 99, counter)
    m = O.dropout_mask(x.shape, 0.5, 99, 3)
    assert np.allclose(wide.numpy()[:, 2:7], x * m * 2) and not wide.numpy()[:, :2].any()


def test_dropout_networks_against_the_interpreter(gpu):
    """g_unet(dropout=True) (p2p.py:200-223) and default_generator(dropout_p) (dcgan.py:25-26) through the engine:
    forward with fresh masks per pass, backward through the same masks -- against the layer-graph interpreter on the
    oracle's ops (tests/golden/symtheano.py) with the same keys and step counter"""
    import sys
    dev, ops, D = gpu
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import symtheano as ST
    from oracle import tape as TP
    from gan_heightmaps_amd import init as INIT, layers as L
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.engine import NetPlan, ParamStore
    from gan_heightmaps_amd.nonlinearities import tanh
    for which in ("unet", "dcgan"):
        INIT.set_rng(np.random.RandomState(3))
        if which == "unet":
            net = p2p.g_unet(32, True, False, nf=4, act=tanh, dropout=True, bilinear_upsample=True)
            xin = np.random.RandomState(1).rand(3, 1, 32, 32).astype(np.float32)
        else:
            net = dcgan.default_generator(16, True, nch=16, div=[2, 2, 4], final_size=32, dropout_p=0.3)
            xin = np.random.RandomState(1).rand(3, 16).astype(np.float32)
        drops = [l for l in L.get_all_layers(net) if isinstance(l, L.DropoutLayer)]
        assert len(drops) == 3
        in_layer = [l for l in L.get_all_layers(net) if isinstance(l, L.InputLayer)][0]
        store = ParamStore(dev, L.get_all_params(net))
        plan = NetPlan(dev, ops, net, 3, store, name=which, rng_seed=5)
        keys = {id(n.layer): n.aux['key'] for n in plan.dropout_nodes}
        fwd, bwd = [], []
        plan.emit_forward(fwd)
        seed = np.random.RandomState(2).randn(*plan.out.shape).astype(np.float32)
        seed_d = dev.tensor(seed)
        plan.emit_backward(bwd, seed_d)
        outs = []
        for step in (1, 2):                      # the counter is ticked at the start of every forward pass
            plan.input_nodes[0].out.set(xin.reshape(plan.input_nodes[0].shape))
            seed_d.set(seed)
            for e in fwd + bwd:
                e[1]()
            got = plan.out.numpy()
            c = ST.Ctx({'X': xin, '__rng__': lambda l, step=step: (keys[id(l)], step)}, np.float64)
            ref = ST.get_output(net, ST.placeholder('X')).ev(c)
            assert rel(got, ref.v) < 1e-5, (which, step)
            TP.backward(ref, seed.astype(np.float64).reshape(ref.v.shape))
            for p in L.get_all_params(net, trainable=True):
                g_ref = c.param(p).g
                if g_ref is None or np.linalg.norm(g_ref) < 1e-9:
                    continue
                assert rel(store.download_grad(p), g_ref) < 5e-4, (which, step, p.name, p.shape)
            outs.append(got)
        assert not np.array_equal(outs[0], outs[1])          # fresh masks per pass
        det = []
        plan_det_prog = []
        plan.emit_forward(plan_det_prog, deterministic=True)
        for e in plan_det_prog:
            e[1]()
        c = ST.Ctx({'X': xin}, np.float64)
        assert rel(plan.out.numpy(), ST.get_output(net, ST.placeholder('X'), deterministic=True).ev(c).v) < 1e-5


POOL_CASES = [
    # N, C, H, W, K, k        (stride 1, 'same'); which kernel takes it
    (8, 1, 64, 64, 64, 5),     # thin fan-out (d_conv1): one window per lane
    (2, 4, 128, 128, 64, 3),   # thin fan-out, 4 input channels
    (8, 16, 128, 128, 64, 5),  # patch kernel 5x5
    (4, 16, 128, 128, 128, 5),  # patch kernel 5x5, two filter tiles
    (8, 32, 128, 128, 64, 3),  # patch kernel 3x3
]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", POOL_CASES)
def test_conv_lrelu_maxpool_fused(gpu, case, dtype):
    """Conv2DLayer -> LeakyRectify(0.2) -> MaxPool2DLayer(2) (architectures/dcgan.py:42-47) as ONE kernel
    (ghm_conv2d_fwd_pool) + the mask backward (ghm_maxpool2_mask_bwd): the pooled values against the oracle, and --
    because the fused epilogue pools the very accumulators the unfused kernel stores -- BITWISE against conv + max-pool
    as two launches, mask included (checked through the backward, which must equal ghm_maxpool2_bwd on the
    materialised activation bit for bit, ties and LeakyReLU slope included)."""
    from oracle import lp as LP
    dev, ops, D = gpu
    N, C, H, W, K, k = case
    pad = k // 2
    rng = np.random.RandomState(sum(case))
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, 1, pad)
    form = ops.conv_pool_supported(d, 'lrelu', dtype)
    assert form == (1 if (C <= 4 or dtype == 'f32') else 2), (case, dtype, form)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    xd, bd = dev.tensor(x), dev.tensor(b)
    w = wp
    if form == 2:
        w = dev.alloc(ops.lp_weight_bytes(d, False))
        ops.lp_pack_weights(d, wp, w, dtype, False)
    pooled = dev.empty((N, K, H // 2, W // 2))
    mask = dev.alloc(N * K * (H // 2) * (W // 2))
    ops.conv2d_fwd_pool(d, xd, w, bd, pooled, mask, 'lrelu', 0.2, dtype)
    got = pooled.numpy()
    # oracle
    if form == 2:
        y64 = LP.conv2d_fwd(x, Wt, b, 1, pad, dtype)
    else:
        y64 = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), 1, pad)
    ref = O.maxpool_fwd(O.lrelu_fwd(y64, 0.2), 2)
    assert rel(got, ref) < (TOL if form == 1 else 2e-5), rel(got, ref)
    # the same two ops unfused, on the device
    yd = dev.empty((N, K, H, W))
    if form == 2:
        ops.conv2d_fwd_lp(d, xd, w, bd, yd, dtype, 'lrelu', 0.2)
    else:
        ops.conv2d_fwd(d, xd, wp, bd, yd, 'lrelu', 0.2)
    pd = dev.empty((N, K, H // 2, W // 2))
    ops.maxpool2_fwd(yd, pd)
    assert np.array_equal(got, pd.numpy())
    dp = rng.randn(N, K, H // 2, W // 2).astype(np.float32)
    dpd = dev.tensor(dp)
    dx_mask, dx_ref = dev.empty((N, K, H, W)), dev.empty((N, K, H, W))
    ops.maxpool2_mask_bwd(mask, pooled, dpd, dx_mask, 'lrelu', 0.2)
    ops.maxpool2_bwd(yd, pd, dpd, dx_ref, 'lrelu', 0.2)
    assert np.array_equal(dx_mask.numpy(), dx_ref.numpy())
    # the variant that also sums what it writes per channel (the conv's bias gradient), plain and accumulating
    db = dev.zeros((1, K, 1, 1))
    dx2 = dev.empty((N, K, H, W))
    ops.maxpool2_mask_bwd(mask, pooled, dpd, dx2, 'lrelu', 0.2, db)
    assert np.array_equal(dx2.numpy(), dx_ref.numpy())
    want_db = dx_ref.numpy().astype(np.float64).sum(axis=(0, 2, 3))
    assert rel(db.numpy().ravel(), want_db) < 1e-5
    ops.maxpool2_mask_bwd(mask, pooled, dpd, dx2, 'lrelu', 0.2, db, True)
    assert rel(db.numpy().ravel(), 2 * want_db) < 1e-5
    vjp = O.lrelu_vjp(y64, 0.2, O.maxpool_vjp(O.lrelu_fwd(y64, 0.2), ref, dp.astype(np.float64), 2))
    assert rel(dx_mask.numpy(), vjp) < 1e-4
    for t in (xd, bd, pooled, yd, pd, dpd, dx_mask, dx_ref, wp):
        dev.free(t.ptr)
    dev.free(mask)


@pytest.mark.parametrize("case", [(4, 64, 64, 64), (3, 96, 128, 64), (2, 40, 256, 24), (2, 64, 1024, 8), (2, 32, 64, 72)])
def test_conv_pool_backward_from_the_pooled_operands(gpu, case):
    """ghm_conv2d_pool_wgrad_sparse / ghm_conv2d_pool_dgrad_sparse (d_conv1: 1 -> K, 5x5, LeakyRectify, MaxPool 2) against
    (a) the float64 oracle VJP of conv -> lrelu -> max-pool and (b) the materialised device path they replace
    (ghm_maxpool2_mask_bwd_bias + ghm_conv2d_wgrad / ghm_conv2d_dgrad).  The input has constant patches, so windows whose
    maximum is tied (several mask bits set: the gradient goes to every arg-max position) are exercised too."""
    dev, ops, D = gpu
    N, H, W, K = case
    rng = np.random.RandomState(sum(case))
    x = rng.randn(N, 1, H, W).astype(np.float32)
    x[:, :, : H // 4, : W // 4] = 0.25                     # constant patch: conv output constant -> 4-way ties
    x[-1] = 0.0                                            # a whole image of ties (output = bias everywhere)
    Wt = (rng.randn(K, 1, 5, 5) / 5.0).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    d = D.conv_desc(N, 1, H, W, K, 5, 5, 1, 2)
    served = ops.pool_bwd_sparse_supported(d, 'lrelu')
    assert served & 1 and bool(served & 2) == (K <= 64)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    xd, bd = dev.tensor(x), dev.tensor(b)
    pooled = dev.empty((N, K, H // 2, W // 2))
    mask = dev.alloc(N * K * (H // 2) * (W // 2))
    if ops.conv_pool_supported(d, 'lrelu', 'f32'):
        ops.conv2d_fwd_pool(d, xd, wp, bd, pooled, mask, 'lrelu', 0.2, 'f32')
    else:                                                  # geometry the fused forward does not take: build its outputs
        y = O.lrelu_fwd(O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), 1, 2), 0.2).astype(np.float32)
        yp = y.reshape(N, K, H // 2, 2, W // 2, 2).max(axis=(3, 5))
        bits = np.zeros(yp.shape, np.uint8)
        for bb, (r, c) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            bits |= ((y[:, :, r::2, c::2] == yp).astype(np.uint8) << bb)
        bits |= ((yp > 0).astype(np.uint8) << 4)           # the sign bit the fused forward kernels leave
        pooled.set(yp)
        dev.h2d(mask, bits)
    gp = rng.randn(N, K, H // 2, W // 2).astype(np.float32)
    gpd = dev.tensor(gp)
    # (b) the materialised path
    Gf = dev.empty((N, K, H, W))
    db_ref = dev.zeros((1, K, 1, 1))
    ops.maxpool2_mask_bwd(mask, pooled, gpd, Gf, 'lrelu', 0.2, db_ref)
    assert (np.count_nonzero(Gf.numpy().reshape(N, K, H // 2, 2, W // 2, 2), axis=(3, 5)) > 1).any(), "no tied window in the test"
    dw_ref = dev.zeros((1, 25 * K, 1, 1))
    ws = dev.alloc(max(ops.wgrad_workspace(d), 16))
    ops.conv2d_wgrad(d, xd, Gf, dw_ref, ws)
    # (a) the oracle on the materialised gradient (itself checked against the oracle VJP in test_conv_lrelu_maxpool_fused)
    dx64, dW64, db64 = O.conv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), Gf.numpy().astype(np.float64), 1, 2)
    ws2 = dev.alloc(max(ops.pool_wgrad_sparse_workspace(d), 16))
    dw, db = dev.zeros((1, 25 * K, 1, 1)), dev.zeros((1, K, 1, 1))
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, pooled, gpd, dw, db, ws2, 'lrelu', 0.2)
    assert rel(D.unpack_conv_w(dw.numpy().ravel(), K, 1, 5, 5), dW64) < TOL
    assert rel(dw.numpy(), dw_ref.numpy()) < TOL
    assert rel(db.numpy().ravel(), db64) < TOL and rel(db.numpy(), db_ref.numpy()) < TOL
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, pooled, gpd, dw, db, ws2, 'lrelu', 0.2, accumulate=True)
    assert rel(dw.numpy(), 2 * dw_ref.numpy()) < TOL and rel(db.numpy().ravel(), 2 * db64) < TOL
    # without the pooled activation: the slope comes from the mask's sign bit -- the same bits
    dw3, db3 = dev.zeros((1, 25 * K, 1, 1)), dev.zeros((1, K, 1, 1))
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, pooled, gpd, dw, db, ws2, 'lrelu', 0.2)
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, None, gpd, dw3, db3, ws2, 'lrelu', 0.2)
    assert np.array_equal(dw.numpy(), dw3.numpy()) and np.array_equal(db.numpy(), db3.numpy())
    Gf2 = dev.empty((N, K, H, W))
    ops.maxpool2_mask_bwd(mask, None, gpd, Gf2, 'lrelu', 0.2)
    assert np.array_equal(Gf2.numpy(), Gf.numpy())
    dw2 = dev.zeros((1, 25 * K, 1, 1))
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, pooled, gpd, dw2, None, ws2, 'lrelu', 0.2)       # no bias gradient asked
    ops.conv2d_pool_wgrad_sparse(d, xd, mask, pooled, gpd, dw, None, ws2, 'lrelu', 0.2)
    assert np.array_equal(dw.numpy(), dw2.numpy())                                           # bit-repeatable
    if served & 2:
        dx_ref = dev.empty(x.shape)
        ops.conv2d_dgrad(d, Gf, wp, dx_ref)
        dx = dev.zeros(x.shape)
        ops.conv2d_pool_dgrad_sparse(d, mask, pooled, gpd, wp, dx, 'lrelu', 0.2)
        assert rel(dx.numpy(), dx64) < TOL and rel(dx.numpy(), dx_ref.numpy()) < TOL
        dxn = dev.zeros(x.shape)
        ops.conv2d_pool_dgrad_sparse(d, mask, None, gpd, wp, dxn, 'lrelu', 0.2)
        assert np.array_equal(dxn.numpy(), dx.numpy())
        ops.conv2d_pool_dgrad_sparse(d, mask, pooled, gpd, wp, dx, 'lrelu', 0.2, accumulate=True)
        assert rel(dx.numpy(), 2 * dx64) < TOL
        # a sample slice (the generator's gradient only needs the fake half of the batch)
        if N >= 2:
            ds = D.conv_desc(1, 1, H, W, K, 5, 5, 1, 2)
            per = K * (H // 2) * (W // 2)
            dxs = dev.zeros((1, 1, H, W))
            ops.conv2d_pool_dgrad_sparse(ds, mask + per, pooled.samples(1, 2), gpd.samples(1, 2), wp, dxs, 'lrelu', 0.2)
            assert rel(dxs.numpy(), dx64[1:2]) < TOL


def test_conv_pool_not_served_is_refused(gpu):
    dev, ops, D = gpu
    from gan_heightmaps_amd._lib import GhmError
    d = D.conv_desc(2, 16, 16, 16, 64, 5, 5, 1, 2)          # small map: too few tiles for a single-pass plan
    assert ops.conv_pool_supported(d, 'lrelu') == 0
    assert ops.conv_pool_supported(D.conv_desc(8, 16, 128, 128, 64, 5, 5, 1, 2), 'tanh') == 0
    with pytest.raises(GhmError):
        ops.conv2d_fwd_pool(d, dev.zeros((2, 16, 16, 16)), dev.zeros((1, 64, 1, 1)), None, dev.zeros((2, 64, 8, 8)),
                            dev.alloc(2 * 64 * 64), 'lrelu', 0.2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(8, 64, 256, 256, 128, 3, 2, 1), (8, 128, 128, 128, 256, 3, 2, 1), (4, 64, 32, 32, 1, 3, 2, 1),
                                  (2, 32, 16, 16, 2, 5, 1, 2)])
def test_dgrad_with_producer_activation_backward_fused(gpu, case, dtype):
    """ghm_conv2d_dgrad_dact: dx = conv^T(dy) * act'(y) in one kernel (PatchGAN conv -> LeakyRectify -> conv chains,
    p2p.py:285-286) == the plain data gradient followed by ghm_act_bwd, bit for bit, in every served form (<= 4
    filters; 3x3 stride 2 on the fp32 and on the bf16 matrix cores)."""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case))
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    form = ops.dgrad_dact_supported(d, dtype)
    assert form == (1 if K <= 4 else (3 if dtype == 'bf16' else 2)), (case, dtype, form)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    dy = dev.tensor(rng.randn(N, K, d.Ho, d.Wo).astype(np.float32))
    y = dev.tensor(rng.randn(N, C, H, W).astype(np.float32))           # the producer's (post-LeakyReLU) output
    fused, plain = dev.empty((N, C, H, W)), dev.empty((N, C, H, W))
    if form == 1:
        w = wp
        ops.conv2d_dgrad(d, dy, wp, plain)
    elif form == 2:
        w = dev.empty((1, C * k * k * K, 1, 1))
        ops.transpose_weights(d, wp, w)
        ops.conv2d_dgrad_t(d, dy, w, plain)
    else:
        w = dev.alloc(ops.lp_weight_bytes(d, True))
        ops.lp_pack_weights(d, wp, w, dtype, True)
        ops.conv2d_dgrad_lp(d, dy, w, plain, dtype)
    ops.act_bwd(plain, y, plain, 'lrelu', 0.01)
    ops.conv2d_dgrad_dact(d, dy, w, fused, y, 'lrelu', 0.01, dtype)
    a, b = fused.numpy(), plain.numpy()
    assert np.array_equal(a, b) and np.abs(a).max() > 0
    for t in (wp, dy, y, fused, plain):
        dev.free(t.ptr)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tile", ["0", "1", "2"])
@pytest.mark.parametrize("splits", ["1", "3"])
@pytest.mark.parametrize("case", [(2, 128, 64, 64, 48, 3, 2, 1), (1, 96, 128, 64, 32, 3, 2, 1), (3, 64, 24, 64, 16, 3, 2, 1)])
def test_stride2_data_gradient_every_tile_and_split(gpu, case, tile, splits, dtype):
    """dgrad_s2_patch_kernel / lp_dgrad_s2_kernel: the plan picks one of three tiles (128 channels x 2 class rows,
    64 x 4, 64 x 2) and splits the contraction only on small grids; here every tile and a forced 3-way split run on
    the same layers (ragged channel tiles, class-row counts that are not a multiple of the tile, rectangular maps)."""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(7)
    x_shape = (N, C, H, W)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    dy = rng.randn(N, K, d.Ho, d.Wo).astype(np.float32)
    lp = dtype != "f32"
    Wr, dyr = (LP.round_bf16(Wt), LP.round_bf16(dy)) if lp else (Wt, dy)
    dx_ref, _, _ = O.conv2d_vjp(np.zeros(x_shape), Wr.astype(np.float64), dyr.astype(np.float64), s, pad)
    wd = dev.tensor(D.pack_conv_w(Wt).reshape(1, -1, 1, 1))
    dyd, dxd = dev.tensor(dy), dev.empty(x_shape)
    env = {"GHM_LP_DGRAD_S2_TILE" if lp else "GHM_DGRAD_S2_TILE": tile,
           "GHM_LP_DGRAD_S2_SPLITS" if lp else "GHM_DGRAD_S2_SPLITS": splits}
    if lp and (d.Ho % (2 if tile == "0" else (4 if tile == "1" else 2))):
        pytest.skip("class rows not a multiple of this tile: the plan never picks it")
    with tuning_env(**env):
        if lp:
            if not ops.lp_supported(d, 1, dtype):
                pytest.skip("geometry not served at reduced precision")
            wqT = dev.alloc(ops.lp_weight_bytes(d, True))
            ops.lp_pack_weights(d, wd, wqT, dtype, True)
            ops.conv2d_dgrad_lp(d, dyd, wqT, dxd, dtype)
            assert rel(dxd.numpy(), dx_ref) < 2e-5
            ops.conv2d_dgrad_lp(d, dyd, wqT, dxd, dtype, accumulate=True)
            assert rel(dxd.numpy(), 2 * dx_ref) < 2e-5
        else:
            assert ops.dgrad_t_supported(d)
            wtd = dev.zeros((1, C * k * k * K, 1, 1))
            ops.transpose_weights(d, wd, wtd)
            name = ops.conv_variant(d, 3)
            assert name.startswith("dgrad_s2_patch_kernel<%s" % {"0": "128, 2", "1": "64, 1", "2": "64, 2"}[tile]), name
            ops.conv2d_dgrad_t(d, dyd, wtd, dxd)
            assert rel(dxd.numpy(), dx_ref) < TOL
            ops.conv2d_dgrad_t(d, dyd, wtd, dxd, accumulate=True)
            assert rel(dxd.numpy(), 2 * dx_ref) < TOL


@pytest.mark.parametrize("splits", ["1", "3", "7", "50"])
@pytest.mark.parametrize("case", [(3, 16, 24, 64, 32, 3, 1, 1), (2, 8, 20, 96, 48, 5, 1, 2), (3, 16, 48, 128, 24, 3, 2, 1),
                                  (2, 12, 10, 48, 16, 3, 1, 1)])
def test_weight_gradient_split_ranges_cross_strips_and_images(gpu, case, splits):
    """wgrad_patch_kernel walks the slabs in column strips (image, strip, row); a split's range may start and end
    anywhere -- inside a strip, across strips, across images -- and the first / last rows of every strip take the
    bounds-checked path.  Forced split counts that do not divide the slab count, 2-3 strips per row, 16- and 32-pixel slabs."""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(11)
    x = rng.randn(N, C, H, W).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    dy = rng.randn(N, K, d.Ho, d.Wo).astype(np.float32)
    _, dW_ref, _ = O.conv2d_vjp(x.astype(np.float64), np.zeros((K, C, k, k)), dy.astype(np.float64), s, pad)
    with tuning_env(GHM_WGRAD_SPLITS=splits):
        assert ops.conv_variant(d, 2).startswith("wgrad_patch_kernel"), ops.conv_variant(d, 2)
        xd, dyd = dev.tensor(x), dev.tensor(dy)
        dwd = dev.zeros((1, C * k * k * K, 1, 1))
        ws = dev.alloc(max(ops.wgrad_workspace(d), 16))
        ops.conv2d_wgrad(d, xd, dyd, dwd, ws)
        assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), dW_ref) < TOL
        ops.conv2d_wgrad(d, xd, dyd, dwd, ws, accumulate=True)
        assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), 2 * dW_ref) < TOL


@pytest.mark.parametrize("case", [(4, 64, 16, 16, 64, 3, 1, 1), (4, 32, 32, 32, 96, 3, 2, 1), (3, 48, 8, 8, 64, 3, 2, 1),
                                  (4, 32, 2, 2, 64, 2, 1, 0), (2, 16, 4, 4, 32, 5, 1, 2), (4, 24, 4, 4, 40, 3, 1, 1)])
def test_fp32_convolution_with_the_batchnorm_in_its_finishing_kernel(gpu, case):
    """ghm_conv2d_bn_fwd: fp32 Conv2DLayer -> BatchNormLayer -> nonlinearity on small maps as ONE product (the generic
    gather kernel in split-K form + the finishing kernel that holds the whole map of its channels): conv output, batch
    statistics, running update and y against the float64 oracle; equal to conv + ghm_bn_forward."""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(3)
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    gamma, beta = (1 + 0.2 * rng.randn(K)).astype(np.float32), (0.3 * rng.randn(K)).astype(np.float32)
    rm0, ri0 = rng.randn(K).astype(np.float32), (1 + 0.1 * rng.rand(K)).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.conv_bn_fused_supported(d, 'f32')
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    shp = (N, K, d.Ho, d.Wo)
    co, y = dev.empty(shp), dev.empty(shp)
    mean, inv = dev.empty((1, K, 1, 1)), dev.empty((1, K, 1, 1))
    rm, ri = dev.tensor(rm0), dev.tensor(ri0)
    ops.conv2d_bn_fwd(d, dev.tensor(x), wp, dev.tensor(b), co, y, dev.tensor(gamma), dev.tensor(beta), mean, inv, rm, ri,
                      1e-4, 0.1, 'lrelu', 0.01)
    c64 = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    assert rel(co.numpy(), c64) < 1e-5
    yb, mu, iv = O.bn_train_fwd(co.numpy().astype(np.float64), beta.astype(np.float64), gamma.astype(np.float64))
    assert rel(mean.numpy().ravel(), mu.ravel()) < 1e-6 and rel(inv.numpy().ravel(), iv.ravel()) < 1e-6
    assert rel(y.numpy(), O.lrelu_fwd(yb, 0.01)) < 1e-5
    nm, ni = O.bn_running_update(rm0.astype(np.float64), ri0.astype(np.float64), mu.ravel(), iv.ravel())
    assert rel(rm.numpy().ravel(), nm) < 1e-6 and rel(ri.numpy().ravel(), ni) < 1e-6
    co2, y2 = dev.empty(shp), dev.empty(shp)
    ops.conv2d_fwd(d, dev.tensor(x), wp, dev.tensor(b), co2, 'linear', 0.0)
    m2, i2 = dev.empty((1, K, 1, 1)), dev.empty((1, K, 1, 1))
    ops.bn_forward(co2, y2, m2, i2, dev.tensor(gamma), dev.tensor(beta), dev.alloc(ops.bn_workspace(K)), None, None, 1e-4, 0.1,
                   'lrelu', 0.01)
    assert rel(co.numpy(), co2.numpy()) < 1e-6 and rel(y.numpy(), y2.numpy()) < 2e-6
