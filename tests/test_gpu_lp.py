"""bf16 / fp16 matrix-core convolutions (csrc/conv_lp.hip, BASELINE configs 4 / 5) through the C ABI, op by op.

Two bounds per case (rel-L2):
  * EXACT-GIVEN-ROUNDING <= 2e-5: against oracle/lp.py -- the float64 convolution of the operands rounded to
    bf16 / fp16 (nearest-even).  Products of two 8-bit (11-bit) mantissas are exact in fp32, so what is left is the
    fp32 accumulation order: this is the check that the kernels compute the right thing.
  * PRECISION <= 8e-3 (bf16) / 1.5e-3 (fp16): against the unrounded float64 convolution -- what rounding the operands
    costs (2^-9 / 2^-12 relative per operand, random signs).  This is the tolerance the reduced-precision step is
    specified to (north_star's 1e-3 is the bound of the fp32 path, which stays the default).
"""
import os

import numpy as np
import pytest

from oracle import lp as LP
from oracle import ops as O

pytestmark = pytest.mark.gpu

EXACT = 2e-5
PREC = {'bf16': 8e-3, 'f16': 1.5e-3}


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


CASES = [
    # N, C, H, W, K, k, s, pad
    (2, 16, 32, 32, 128, 5, 1, 2),     # one slab, 5x5, 128-row tile
    (1, 32, 32, 64, 64, 5, 1, 2),      # 64-row tile, rectangular
    (2, 48, 32, 32, 160, 3, 1, 1),     # ragged filter tile (160 = 128 + 32), 3 slabs
    (1, 64, 64, 32, 40, 3, 1, 1),      # 40 filters on the 64-row tile (masked rows)
    (2, 32, 64, 64, 128, 3, 2, 1),     # 3x3 stride 2 -> 32x32, 128-row tile
    (1, 16, 64, 128, 64, 3, 2, 1),     # 3x3 stride 2, 64-row tile, rectangular
    (1, 128, 32, 32, 96, 3, 1, 1),     # 8 slabs (split-K when forced)
    (3, 80, 32, 32, 32, 5, 1, 2),      # 5 slabs, 32 filters
    (2, 128, 64, 64, 48, 3, 2, 1),     # stride-2 data gradient, 128-row tile (2 class rows per block)
    (1, 48, 128, 64, 32, 3, 2, 1),     # stride-2 data gradient, 64-row tile (4 class rows), rectangular
]


def _tensors(case, seed):
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(seed)
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = rng.randn(N, K, Ho, Wo).astype(np.float32)
    return x, Wt, b, dy


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", CASES)
def test_lp_conv_forward_dgrad_wgrad(gpu, case, dtype):
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    x, Wt, b, dy = _tensors(case, sum(case))
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    xd, bd, dyd = dev.tensor(x), dev.tensor(b), dev.tensor(dy)
    # ---- forward ----
    assert ops.lp_supported(d, 0, dtype), "forward geometry should be served"
    wq = dev.alloc(ops.lp_weight_bytes(d, False))
    ops.lp_pack_weights(d, wp, wq, dtype, False)
    y_lp = LP.conv2d_fwd(x, Wt, b, s, pad, dtype)
    y_64 = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    yd = dev.empty(y_lp.shape)
    ops.conv2d_fwd_lp(d, xd, wq, bd, yd, dtype)
    got = yd.numpy()
    assert rel(got, y_lp) < EXACT, ("fwd exact", rel(got, y_lp))
    assert rel(got, y_64) < PREC[dtype], ("fwd precision", rel(got, y_64))
    ops.conv2d_fwd_lp(d, xd, wq, bd, yd, dtype, act='lrelu', alpha=0.2)
    assert rel(yd.numpy(), O.lrelu_fwd(y_lp, 0.2)) < EXACT
    ops.conv2d_fwd_lp(d, xd, wq, None, yd, dtype, act='tanh')
    assert rel(yd.numpy(), np.tanh(y_lp - b[None, :, None, None])) < EXACT
    # ---- gradients ----
    dx_lp, dW_lp, _ = LP.conv2d_vjp(x, Wt, dy, s, pad, dtype)
    dx_64, dW_64, _ = O.conv2d_vjp(x.astype(np.float64), Wt.astype(np.float64), dy.astype(np.float64), s, pad)
    # data gradient: at least 32 input channels (its output rows) and filters % 16 == 0 (its reduction slabs);
    # anything else stays on the fp32 kernels
    assert ops.lp_supported(d, 1, dtype) == (C >= 32 and K % 16 == 0)
    if ops.lp_supported(d, 1, dtype):
        wqT = dev.alloc(ops.lp_weight_bytes(d, True))
        ops.lp_pack_weights(d, wp, wqT, dtype, True)
        dxd = dev.zeros(x.shape)
        ops.conv2d_dgrad_lp(d, dyd, wqT, dxd, dtype)
        assert rel(dxd.numpy(), dx_lp) < EXACT, ("dgrad exact", rel(dxd.numpy(), dx_lp))
        assert rel(dxd.numpy(), dx_64) < PREC[dtype]
        ops.conv2d_dgrad_lp(d, dyd, wqT, dxd, dtype, accumulate=True)      # skip-connection accumulation
        assert rel(dxd.numpy(), 2 * dx_lp) < EXACT
        dev.free(wqT)
    assert ops.lp_supported(d, 2, dtype), "weight-gradient geometry should be served"
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    dwd = dev.zeros((1, C * k * k * K, 1, 1))
    ops.conv2d_wgrad_lp(d, xd, dyd, dwd, ws, dtype)
    gw = D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k)
    assert rel(gw, dW_lp) < EXACT, ("wgrad exact", rel(gw, dW_lp))
    assert rel(gw, dW_64) < PREC[dtype]
    ops.conv2d_wgrad_lp(d, xd, dyd, dwd, ws, dtype, accumulate=True)
    assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), 2 * dW_lp) < EXACT
    for p in (wq, ws):
        dev.free(p)


def test_lp_strided_views_and_forced_splits(gpu):
    """channel slices of wider buffers (ConcatLayer in place: explicit sample strides) and split-K on both kernels"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = 2, 64, 32, 32, 96, 3, 1, 1
    x, Wt, b, dy = _tensors((N, C, H, W, K, k, s, pad), 5)
    wide_x = dev.zeros((N, C + 16, H, W))
    wide_y = dev.zeros((N, K + 32, H, W))
    xv, yv = wide_x.channels(16, 16 + C), wide_y.channels(32, 32 + K)
    xv.set(x)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad, xv.nstride, yv.nstride)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wq = dev.alloc(ops.lp_weight_bytes(d, False))
    ops.lp_pack_weights(d, wp, wq, 'bf16', False)
    y_lp = LP.conv2d_fwd(x, Wt, b, s, pad, 'bf16')
    for env in ({}, {"GHM_LP_SPLITS": "4"}):
        os.environ.update(env)
        try:
            dev.memset_zero(wide_y.ptr, 4 * wide_y.size)
            ops.conv2d_fwd_lp(d, xv, wq, dev.tensor(b), yv, 'bf16')
        finally:
            for key in env:
                os.environ.pop(key)
        full = wide_y.numpy()
        assert rel(full[:, 32:], y_lp) < EXACT and not full[:, :32].any()
    yv.set(dy)
    dW_lp = LP.conv2d_vjp(x, Wt, dy, s, pad, 'bf16')[1]
    for env in ({"GHM_LP_WGRAD_SPLITS": "1"}, {"GHM_LP_WGRAD_SPLITS": "7"}, {"GHM_LP_WGRAD_SEG1": "1"}):
        os.environ.update(env)
        try:
            ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
            dwd = dev.zeros((1, C * k * k * K, 1, 1))
            ops.conv2d_wgrad_lp(d, xv, yv, dwd, ws, 'bf16')
        finally:
            for key in env:
                os.environ.pop(key)
        assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), dW_lp) < EXACT, env


def test_lp_unsupported_geometries_are_refused(gpu):
    """the low-precision entry points never fall back silently: geometries they do not serve are refused and the
    caller keeps them on the fp32 kernels"""
    dev, ops, D = gpu
    from gan_heightmaps_amd._lib import GhmError
    for case in [(2, 1, 32, 32, 64, 5, 1, 2),      # 1 input channel (thin layer)
                 (2, 24, 32, 32, 64, 3, 1, 1),     # channels % 16 != 0
                 (2, 32, 16, 16, 64, 3, 1, 1),     # 16-wide map
                 (2, 32, 32, 32, 64, 1, 1, 0)]:    # 1x1
        N, C, H, W, K, k, s, pad = case
        d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
        assert not ops.lp_supported(d, 0, 'bf16')
        with pytest.raises(GhmError):
            ops.conv2d_fwd_lp(d, dev.zeros((N, C, H, W)), dev.zeros((1, 64, 1, 1)), None, dev.zeros((N, K, d.Ho, d.Wo)), 'bf16')
    d = D.conv_desc(2, 32, 32, 32, 64, 3, 3, 1, 1)
    assert ops.lp_supported(d, 0, 'bf16') and not ops.lp_supported(d, 0, 'f32')
