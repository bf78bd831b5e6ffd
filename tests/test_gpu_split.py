"""fp32 convolutions on the bf16 matrix cores by operand splitting (csrc/conv_split.hip) through the C ABI, op by op.

The claim under test: splitting each fp32 operand into three bf16 pieces and running the six leading piece products with
fp32 accumulation is fp32 arithmetic -- not a reduced-precision mode.  So every case is held to the bound of the fp32
path (rel-L2 <= 2e-6 against the float64 oracle of the SAME unrounded operands, tests/test_gpu_ops.py) and, next to it, to
the error the v_mfma_f32_32x32x2_f32 kernels commit on the same inputs (not more than twice that, with a floor of 3e-7).
For scale: rounding the operands to bf16 once costs 4e-3 on these cases (tests/test_gpu_lp.py, PREC).
"""
import copy

import numpy as np
import pytest

from gan_heightmaps_amd._lib import tuning_env

from oracle import ops as O

pytestmark = pytest.mark.gpu

FP32_BOUND = 2e-6


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


def test_three_bf16_pieces_are_the_fp32_value(gpu):
    """ghm_split_pack: piece0 + piece1 + piece2 == x bit for bit (values across the fp32 exponent range, both signs,
    zeros), piece0 is the nearest bf16, and every piece is a bf16 value"""
    dev, ops, D = gpu
    rng = np.random.RandomState(0)
    N, C, H, W = 2, 16, 8, 32
    x = (rng.randn(N, C, H, W) * np.exp2(rng.randint(-60, 60, size=(N, C, H, W)))).astype(np.float32)
    x[0, 0, 0, :8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0000001, 0.33333334]
    xd = dev.tensor(x)
    buf = dev.alloc(3 * x.size * 2)
    ns, ps = ops.split_pack(xd, buf)
    raw = np.empty(3 * x.size, np.uint16)
    dev.d2h(raw, buf, raw.nbytes)
    pieces = (raw.astype(np.uint32) << 16).view(np.float32).reshape(3, N, C // 8, H * W, 8)
    pieces = pieces.transpose(0, 1, 2, 4, 3).reshape(3, N, C, H, W)
    total = (pieces[0].astype(np.float64) + pieces[1].astype(np.float64) + pieces[2].astype(np.float64)).astype(np.float32)
    assert np.array_equal(total.view(np.uint32) & 0x7fffffff, x.view(np.uint32) & 0x7fffffff) or np.array_equal(total, x)
    assert np.array_equal(total, x)
    from oracle import lp as LP
    for got, want in zip(pieces, LP.split_bf16x3(x)):          # each piece is the oracle's piece, bit for bit
        assert np.array_equal(got, want)


CASES = [
    # N, C, H, W, K, k, s, pad
    (2, 16, 32, 32, 64, 5, 1, 2),      # one slab, 5x5
    (1, 32, 32, 64, 96, 5, 1, 2),      # ragged filter tile (64 + 32), rectangular
    (2, 48, 32, 32, 160, 3, 1, 1),     # 128-row tile + ragged, 3 slabs, eight waves
    (1, 64, 64, 32, 40, 3, 1, 1),      # 40 filters on the 64-row tile (masked rows)
    (2, 32, 64, 64, 128, 3, 2, 1),     # 3x3 stride 2 -> 32x32
    (1, 16, 64, 128, 64, 3, 2, 1),     # 3x3 stride 2, rectangular
    (1, 128, 32, 32, 96, 3, 1, 1),     # 8 slabs
    (3, 80, 32, 32, 32, 5, 1, 2),      # 5 slabs, 32 filters
    (2, 64, 16, 16, 128, 3, 1, 1),     # narrow maps: 16 columns (fragments of 2 x 16 pixels) ...
    (3, 32, 16, 16, 64, 5, 1, 2),
    (2, 64, 8, 8, 160, 3, 1, 1),       # ... and 8 columns (4 x 8)
    (4, 48, 8, 8, 64, 5, 1, 2),
    (2, 32, 32, 32, 64, 3, 2, 1),      # stride 2 -> 16 x 16
    (4, 512, 16, 16, 512, 3, 1, 1),    # split-K (few blocks, 32 slabs)
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("presplit", [False, True])
def test_split_forward_is_fp32_arithmetic(gpu, case, presplit):
    """ghm_conv2d_fwd_split (forward of Conv2DLayer, architectures/dcgan.py:22,42 / p2p.py:20-21) against the float64
    oracle, beside ghm_conv2d_fwd (fp32 MFMA) on the same inputs; with the input given as fp32 and as a split q tensor"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case))
    x = (rng.randn(N, C, H, W) * np.exp(rng.randn(N, C, 1, 1))).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.split_supported(d, 0)
    ref = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    xd, bd = dev.tensor(x), dev.tensor(b)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wq = dev.alloc(ops.split_weight_bytes(d, False))
    ops.split_pack_weights(d, wp, wq, False)
    y = dev.empty((N, K, d.Ho, d.Wo))
    xq = None
    if presplit:
        buf = dev.alloc(3 * x.size * 2)
        xq = (buf,) + ops.split_pack(xd, buf)
    ops.conv2d_fwd_split(d, xd, wq, bd, y, 'lrelu', 0.2, xq=xq)
    got = y.numpy()
    y32 = dev.empty((N, K, d.Ho, d.Wo))
    ops.conv2d_fwd(d, xd, wp, bd, y32, 'lrelu', 0.2)
    e_split, e_f32 = rel(got, ref), rel(y32.numpy(), ref)
    print("split %.2e   fp32 MFMA %.2e   %s" % (e_split, e_f32, case))
    assert e_split < FP32_BOUND and e_split < max(2 * e_f32, 3e-7), (e_split, e_f32)
    # accumulate form (a data gradient summed into an existing one)
    ops.conv2d_fwd_split(d, xd, wq, None, y, 'linear', 0.0, accumulate=True, xq=xq)
    lin = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), np.zeros(K), s, pad)
    assert rel(y.numpy(), ref + lin) < FP32_BOUND
    for ptr in ([wq] + ([xq[0]] if xq else [])):
        dev.free(ptr)


@pytest.mark.parametrize("case", [c for c in CASES if c[6] == 1])
def test_split_data_gradient_is_fp32_arithmetic(gpu, case):
    """ghm_conv2d_dgrad_split (stride-1 data gradient on the transposed split pack) against the float64 oracle, beside the
    fp32 MFMA data gradient"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 1)
    dy = (rng.randn(N, K, H, W) * np.exp(rng.randn(N, K, 1, 1))).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    if not ops.split_supported(d, 1):
        pytest.skip("data gradient geometry not served (fewer than 32 input channels)")
    ref = O.conv2d_vjp(np.zeros((N, C, H, W)), Wt.astype(np.float64), dy.astype(np.float64), s, pad)[0]
    dyd = dev.tensor(dy)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wqT = dev.alloc(ops.split_weight_bytes(d, True))
    ops.split_pack_weights(d, wp, wqT, True)
    dx = dev.empty((N, C, H, W))
    ops.conv2d_dgrad_split(d, dyd, wqT, dx)
    dx32 = dev.empty((N, C, H, W))
    if ops.dgrad_t_supported(d):
        wT = dev.empty((1, C * k * k * K, 1, 1))
        ops.transpose_weights(d, wp, wT)
        ops.conv2d_dgrad_t(d, dyd, wT, dx32)
    else:
        ops.conv2d_dgrad(d, dyd, wp, dx32)
    e_split, e_f32 = rel(dx.numpy(), ref), rel(dx32.numpy(), ref)
    print("split %.2e   fp32 MFMA %.2e   %s" % (e_split, e_f32, case))
    assert e_split < FP32_BOUND and e_split < max(2 * e_f32, 3e-7), (e_split, e_f32)
    dev.free(wqT)


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 48, 3, 2, 1), (1, 48, 128, 64, 32, 3, 2, 1), (2, 128, 64, 128, 256, 3, 2, 1),
                                  (1, 32, 24, 64, 16, 3, 2, 1)])
def test_split_stride2_data_gradient_is_fp32_arithmetic(gpu, case):
    """ghm_conv2d_dgrad_split on a 3x3 stride-2 convolution (four parity classes over one dy patch) against the float64
    oracle, beside the fp32 MFMA kernel; the accumulate form; and ghm_conv2d_dgrad_dact_split (the producer's LeakyRectify
    backward in the epilogue, p2p.py:285-286)"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 3)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.split_supported(d, 1)
    with tuning_env(GHM_NO_SPLIT_DGRAD_S2="1"):
        assert not ops.split_supported(d, 1)
    _stride2_dgrad_checks(dev, ops, D, case, rng, d)


def _stride2_dgrad_checks(dev, ops, D, case, rng, d):
    N, C, H, W, K, k, s, pad = case
    dy = (rng.randn(N, K, d.Ho, d.Wo) * np.exp(rng.randn(N, K, 1, 1))).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    ref = O.conv2d_vjp(np.zeros((N, C, H, W)), Wt.astype(np.float64), dy.astype(np.float64), s, pad)[0]
    dyd = dev.tensor(dy)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wqT = dev.alloc(ops.split_weight_bytes(d, True))
    ops.split_pack_weights(d, wp, wqT, True)
    dx, dx32 = dev.empty((N, C, H, W)), dev.empty((N, C, H, W))
    ops.conv2d_dgrad_split(d, dyd, wqT, dx)
    wT = dev.empty((1, C * k * k * K, 1, 1))
    ops.transpose_weights(d, wp, wT)
    ops.conv2d_dgrad_t(d, dyd, wT, dx32)
    e_split, e_f32 = rel(dx.numpy(), ref), rel(dx32.numpy(), ref)
    print("split %.2e   fp32 MFMA %.2e   %s" % (e_split, e_f32, case))
    assert e_split < FP32_BOUND and e_split < max(2 * e_f32, 3e-7), (e_split, e_f32)
    ops.conv2d_dgrad_split(d, dyd, wqT, dx, accumulate=True)
    assert rel(dx.numpy(), 2 * ref) < FP32_BOUND
    if ops.dgrad_dact_supported(d, 'bf16x3') == 3:
        y = rng.randn(N, C, H, W).astype(np.float32)
        dyq = D.QTensor.empty(dev, dy.shape, 'bf16x3')
        ops.q_pack(dyd, dyq)
        out = dev.empty((N, C, H, W))
        ops.conv2d_dgrad_dact_lp_q(d, dyq, wqT, out, None, dev.tensor(y), 'lrelu', 0.2, 'bf16x3')
        assert rel(out.numpy(), ref * np.where(y > 0, 1.0, 0.2)) < FP32_BOUND
    dev.free(wqT)


@pytest.mark.parametrize("case", [(4, 1, 128, 128, 64, 5, 1, 2, True), (2, 4, 256, 256, 64, 3, 2, 1, False),
                                  (2, 1, 256, 256, 64, 3, 2, 1, False), (2, 3, 128, 256, 64, 3, 1, 1, False)])
def test_thin_first_layers_write_the_split_q_copy(gpu, case):
    """ghm_conv2d_fwd_thin_q / ghm_conv2d_fwd_pool_thin_q with dtype 3: the first layers (<= 4 input channels: fp32 operands
    on the fp32 kernels) write their result as three exact bf16 pieces from their own epilogue -- the fp32 result is
    bit-identical to the plain entry point's and the pieces sum to it bit for bit (also inside a channel slice)"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad, pooled = case
    rng = np.random.RandomState(sum(case[:8]))
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.thin_fwd_q_supported(d, 'lrelu', pooled, 'bf16x3')
    wp, xd, bd = dev.tensor(D.pack_conv_w(Wt).ravel()), dev.tensor(x), dev.tensor(b)
    Ho, Wo = (d.Ho // 2, d.Wo // 2) if pooled else (d.Ho, d.Wo)
    y1, y2 = dev.empty((N, K, Ho, Wo)), dev.empty((N, K, Ho, Wo))
    wide = D.QTensor.empty(dev, (N, K + 16, Ho, Wo), 'bf16x3')
    dev.memset_zero(wide.ptr, 3 * wide.nbytes)
    yq = wide.channels(8, 8 + K)
    if pooled:
        m1, m2 = dev.alloc(N * K * Ho * Wo), dev.alloc(N * K * Ho * Wo)
        ops.conv2d_fwd_pool(d, xd, wp, bd, y1, m1, 'lrelu', 0.2, 'f32')
        ops.conv2d_fwd_pool_thin_q(d, xd, wp, bd, y2, m2, yq, 'lrelu', 0.2)
    else:
        ops.conv2d_fwd(d, xd, wp, bd, y1, 'lrelu', 0.2)
        ops.conv2d_fwd_thin_q(d, xd, wp, bd, y2, yq, 'lrelu', 0.2)
    assert np.array_equal(y1.numpy(), y2.numpy())
    assert np.array_equal(yq.numpy(), y1.numpy())
    full = wide.numpy()
    assert not full[:, :8].any() and not full[:, 8 + K:].any()


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 64, 5, 1, 2), (2, 48, 32, 32, 160, 3, 1, 1), (2, 32, 64, 64, 128, 3, 2, 1),
                                  (2, 64, 16, 16, 128, 3, 1, 1), (16, 64, 16, 16, 512, 3, 1, 1)])
def test_split_products_write_their_split_q_copy(gpu, case):
    """yq / dxq of ghm_conv2d_fwd_split / ghm_conv2d_dgrad_split (and the stride-2 data gradient with the producer's
    activation backward): the three pieces the epilogue writes sum to the fp32 result bit for bit, inside a channel slice of a
    wider split q tensor; with the fp32 pointer NULL the q result is unchanged"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 5)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    x = rng.randn(N, C, H, W).astype(np.float32)
    dy = rng.randn(N, K, d.Ho, d.Wo).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wq, wqT = dev.alloc(ops.split_weight_bytes(d, False)), dev.alloc(ops.split_weight_bytes(d, True))
    ops.split_pack_weights(d, wp, wq, False)
    ops.split_pack_weights(d, wp, wqT, True)
    xd, dyd, bd = dev.tensor(x), dev.tensor(dy), dev.tensor(b)

    def wide_q(shape):
        wide = D.QTensor.empty(dev, (shape[0], shape[1] + 16, shape[2], shape[3]), 'bf16x3')
        dev.memset_zero(wide.ptr, 3 * wide.nbytes)
        return wide, wide.channels(8, 8 + shape[1])

    if not ops.lp_q_direct(d, 0, 'bf16x3'):
        pytest.skip("split-K plan on this device: the q copy is then made by ghm_split_pack")
    y = dev.empty((N, K, d.Ho, d.Wo))
    wide, yq = wide_q(y.shape)
    ops.conv2d_fwd_split(d, xd, wq, bd, y, 'lrelu', 0.2, yq=yq)
    assert np.array_equal(yq.numpy(), y.numpy())
    full = wide.numpy()
    assert not full[:, :8].any() and not full[:, 8 + K:].any()
    only = D.QTensor.empty(dev, y.shape, 'bf16x3')
    ops.conv2d_fwd_split(d, xd, wq, bd, None, 'lrelu', 0.2, yq=only)
    assert np.array_equal(only.numpy(), y.numpy())
    if ops.lp_q_direct(d, 1, 'bf16x3'):
        dx = dev.empty((N, C, H, W))
        widex, dxq = wide_q(dx.shape)
        ops.conv2d_dgrad_split(d, dyd, wqT, dx, dxq=dxq)
        assert np.array_equal(dxq.numpy(), dx.numpy())
        fullx = widex.numpy()
        assert not fullx[:, :8].any() and not fullx[:, 8 + C:].any()
        if ops.dgrad_dact_supported(d, 'bf16x3') == 3:
            dyq = D.QTensor.empty(dev, dy.shape, 'bf16x3')
            ops.q_pack(dyd, dyq)
            yact = dev.tensor(rng.randn(N, C, H, W).astype(np.float32))
            out, outq = dev.empty((N, C, H, W)), D.QTensor.empty(dev, (N, C, H, W), 'bf16x3')
            ops.conv2d_dgrad_dact_lp_q(d, dyq, wqT, out, outq, yact, 'lrelu', 0.2, 'bf16x3')
            assert np.array_equal(outq.numpy(), out.numpy())


def test_split_pooled_forward_writes_its_split_q_copy(gpu):
    """ghm_conv2d_fwd_pool_split with pooledq: the pooled result as three exact pieces (also with the fp32 pointer NULL), the
    mask unchanged"""
    dev, ops, D = gpu
    N, C, H, W, K, k = 2, 32, 64, 64, 64, 5
    rng = np.random.RandomState(11)
    d = D.conv_desc(N, C, H, W, K, k, k, 1, 2)
    assert ops.conv_pool_supported(d, 'lrelu', 'bf16x3') == 2
    x = rng.randn(N, C, H, W).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    wp, xd, bd = dev.tensor(D.pack_conv_w(Wt).ravel()), dev.tensor(x), dev.tensor(rng.randn(K).astype(np.float32))
    wq = dev.alloc(ops.split_weight_bytes(d, False))
    ops.split_pack_weights(d, wp, wq, False)
    xq = D.QTensor.empty(dev, x.shape, 'bf16x3')
    ops.q_pack(xd, xq)
    n = N * K * (H // 2) * (W // 2)
    y, yq, m1, m2 = dev.empty((N, K, H // 2, W // 2)), D.QTensor.empty(dev, (N, K, H // 2, W // 2), 'bf16x3'), dev.alloc(n), dev.alloc(n)
    ops.conv2d_fwd_pool_lp_q(d, xq, wq, bd, y, yq, m1, 'lrelu', 0.2, 'bf16x3')
    assert np.array_equal(yq.numpy(), y.numpy())
    only = D.QTensor.empty(dev, y.shape, 'bf16x3')
    ops.conv2d_fwd_pool_lp_q(d, xq, wq, bd, None, only, m2, 'lrelu', 0.2, 'bf16x3')
    a, b = np.empty(n, np.uint8), np.empty(n, np.uint8)
    dev.d2h(a, m1, n)
    dev.d2h(b, m2, n)
    assert np.array_equal(only.numpy(), y.numpy()) and np.array_equal(a, b)


WGRAD_CASES = [
    # N, C, H, W, K, k, s, pad
    (2, 64, 32, 32, 64, 3, 1, 1),      # 3x3 stride 1: 12 waves, one strip
    (1, 128, 16, 64, 192, 3, 1, 1),    # two strips, several channel / filter tiles
    (2, 32, 64, 64, 128, 3, 2, 1),     # 3x3 stride 2 (two parity planes per x row)
    (1, 64, 32, 128, 256, 3, 2, 1),    # stride 2, rectangular
    (2, 32, 32, 64, 64, 5, 1, 2),      # 5x5: ten waves, 64-pixel strips
    (1, 64, 16, 32, 128, 5, 1, 2),     # 5x5, 32-pixel strips
    (4, 128, 16, 16, 128, 3, 1, 1),    # 16-wide maps: 16-pixel strips, one k-step per output row
    (2, 64, 32, 32, 128, 3, 2, 1),     # stride 2 -> 16 x 16
    (4, 64, 16, 16, 64, 5, 1, 2),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_split_weight_gradient_is_fp32_arithmetic(gpu, case):
    """ghm_conv2d_wgrad_split (from the split q tensors of x and dy) against the float64 oracle, beside ghm_conv2d_wgrad
    (fp32 MFMA) on the same inputs; also the accumulate form"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 2)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    assert ops.split_supported(d, 2)
    x = (rng.randn(N, C, H, W) * np.exp(rng.randn(N, C, 1, 1))).astype(np.float32)
    dy = (rng.randn(N, K, d.Ho, d.Wo) * np.exp(rng.randn(N, K, 1, 1))).astype(np.float32)
    ref = O.conv2d_vjp(x.astype(np.float64), np.zeros((K, C, k, k)), dy.astype(np.float64), s, pad)[1]
    xd, dyd = dev.tensor(x), dev.tensor(dy)
    xq, dyq = D.QTensor.empty(dev, x.shape, 'bf16x3'), D.QTensor.empty(dev, dy.shape, 'bf16x3')
    ops.q_pack(xd, xq)
    ops.q_pack(dyd, dyq)
    assert np.array_equal(xq.numpy(), x) and np.array_equal(dyq.numpy(), dy)
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), ops.wgrad_workspace(d), 16))
    dw = dev.zeros((1, C * k * k * K, 1, 1))
    ops.conv2d_wgrad_lp_q(d, xq, dyq, dw, ws, 'bf16x3')
    got = D.unpack_conv_w(dw.numpy().ravel(), K, C, k, k)
    dw32 = dev.zeros((1, C * k * k * K, 1, 1))
    ops.conv2d_wgrad(d, xd, dyd, dw32, ws)
    e_split, e_f32 = rel(got, ref), rel(D.unpack_conv_w(dw32.numpy().ravel(), K, C, k, k), ref)
    print("split %.2e   fp32 MFMA %.2e   %s" % (e_split, e_f32, case))
    assert e_split < FP32_BOUND and e_split < max(2 * e_f32, 3e-7), (e_split, e_f32)
    ops.conv2d_wgrad_lp_q(d, xq, dyq, dw, ws, 'bf16x3', accumulate=True)
    assert rel(D.unpack_conv_w(dw.numpy().ravel(), K, C, k, k), 2 * ref) < FP32_BOUND


def test_split_train_step_meets_the_fp32_bounds(gpu):
    """Pix2Pix(dtype='bf16x3'): the joint train step with its served convolutions on the bf16 matrix cores by operand
    splitting, against the float64 oracle of the same step, at the size where the 5x5 / 3x3 stride-1 and 3x3 stride-2
    forward and the stride-1 data gradients take the split kernels; beside it the fp32 model (dtype='f32':
    v_mfma_f32_32x32x2_f32) on the same inputs, held to the SAME bounds.  Losses 1e-5 (the fp32 step test's).  Gradients: this
    configuration's batch-4 BatchNorm chains amplify ANY fp32 rounding -- the float32 numpy oracle itself sits 1e-4 .. 2e-3
    from the float64 one depending on the draw, and re-compiling an element-wise kernel (a different FMA contraction, no other
    change) moved BOTH device modes from 5e-4 to 2e-3 -- so every net is bounded by 2 x the float32 oracle's own distance
    from float64 on the same step + 1e-4 (the full-size tests' formula; a discriminator by its generator's spread as well:
    its gradients inherit the generator's rounding through the fake batch)."""
    from oracle import step as ostep
    from tests.test_gpu_step import build_model, model_grads, model_params
    from tests.test_gpu_lp import LP_STEP
    from gan_heightmaps_amd import layers as L
    dev, ops, D = gpu
    cfg = ostep.default_cfg(**LP_STEP)
    model = build_model(cfg, 7, dev, dtype='bf16x3', use_graph=False)
    f32 = build_model(cfg, 7, dev, dtype='f32', use_graph=False)
    assert model.engine.loss_scale == 1.0 and model.engine.dtype == 'bf16x3' and f32.engine.dtype == 'f32'
    b = model.engine.built(4)
    kinds = {}
    for lane in b.train_compute:
        for e in lane:
            if len(e) > 2 and e[2] is not None and e[2].get("dtype") == 'bf16x3':
                kinds[(e[0], e[2]["kernel"])] = kinds.get((e[0], e[2]["kernel"]), 0) + 1
    labels = {k[0] for k in kinds}
    assert {"conv_fwd", "conv_dgrad", "conv_wgrad", "upconv_fwd", "upconv_dgrad"} <= labels, kinds
    state = ostep.init_state(cfg, 7, np.float32)
    cat = lambda gs: np.concatenate([g.ravel() for g in gs])
    report = []
    for it in range(3):
        Z, X, Y = ostep.synthetic_batch(4, cfg, seed=200 + it)
        state32 = copy.deepcopy(state)                  # (train_step updates the state it is given)
        ref = ostep.train_step(state, Z, X, Y, dtype=np.float64)
        r32 = ostep.train_step(state32, Z, X, Y, dtype=np.float32)
        got = model.train_fn(Z, X, Y)
        exact = f32.train_fn(Z, X, Y)
        assert rel(got, ref['losses']) < 1e-5 and rel(exact, ref['losses']) < 1e-5, (rel(got, ref['losses']), rel(exact, ref['losses']))
        mg, mg32 = model_grads(model), model_grads(f32)
        spread = {key: rel(cat(r32['grads'][key]), cat(ref['grads'][key])) for key in ref['grads']}
        for key in ref['grads']:
            bound = 2 * max(spread[key], spread[(key[0], 'gen')]) + 1e-4
            e_split, e_f32 = rel(cat(mg[key]), cat(ref['grads'][key])), rel(cat(mg32[key]), cat(ref['grads'][key]))
            report.append((it, key, e_split, e_f32, spread[key], bound))
            assert e_split < bound and e_f32 < bound, (it, key, e_split, e_f32, spread)
        mp = model_params(f32)
        for key in ostep.NET_ORDER:
            state['params'][key[0]][key[1]] = [a.copy() for a in mp[key]]
        for (a_, b_), vals in mp.items():
            L.set_all_param_values(getattr(model, a_)[b_], vals)
    for r in report:
        print("step %d %-16s split %.2e   fp32 MFMA %.2e   float32 oracle %.2e   bound %.2e" % ((r[0], "%s/%s" % r[1]) + r[2:]))
    print("kernels %s" % sorted(kinds))


# ---- two pieces, three products ('bf16x2': BASELINE config 4's arithmetic; include/ghm.h ``pieces`` = 2) -----------------------
X2_CASES = [(2, 32, 32, 32, 64, 5, 1, 2), (2, 48, 32, 32, 160, 3, 1, 1), (2, 64, 64, 64, 128, 3, 2, 1), (2, 64, 16, 16, 128, 3, 1, 1),
            (4, 48, 8, 8, 64, 5, 1, 2), (4, 512, 16, 16, 512, 3, 1, 1)]


@pytest.mark.parametrize("case", X2_CASES)
def test_two_piece_products_are_exact_given_the_pieces(gpu, case):
    """the 'bf16x2' kernels (forward, data gradient incl. stride 2, weight gradient) compute the three products x0 w0 + x1 w0 +
    x0 w1 of the two-piece operands exactly (fp32 accumulation: <= 4e-7 against oracle/lp.py's float64 statement of the same
    products) and sit <= 2e-5 from the unrounded float64 convolution -- plain bf16 operands: 4e-3"""
    from oracle import lp as LP
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 11)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    x = (rng.randn(N, C, H, W) * np.exp(rng.randn(N, C, 1, 1))).astype(np.float32)
    dy = (rng.randn(N, K, d.Ho, d.Wo) * np.exp(rng.randn(N, K, 1, 1))).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    xd, dyd, bd = dev.tensor(x), dev.tensor(dy), dev.tensor(b)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    xq, dyq = D.QTensor.empty(dev, x.shape, 'bf16x2'), D.QTensor.empty(dev, dy.shape, 'bf16x2')
    ops.q_pack(xd, xq)
    ops.q_pack(dyd, dyq)
    for got, want in zip((xq.numpy(0), xq.numpy(1)), LP.split_bf16x2(x)):       # the pieces are the oracle's, bit for bit
        assert np.array_equal(got, want)
    # forward (fp32 operand split by the entry point, and the pre-split q tensor), with the q copy of the result
    assert ops.split_supported(d, 0)
    wq = dev.alloc(ops.split_weight_bytes(d, False, 2))
    ops.split_pack_weights(d, wp, wq, False, 2)
    ref_x2 = LP.conv2d_fwd_x2(x, Wt, b, s, pad)
    exact = O.conv2d_fwd(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), s, pad)
    y = dev.empty((N, K, d.Ho, d.Wo))
    ops.conv2d_fwd_split(d, xd, wq, bd, y, pieces=2)
    assert rel(y.numpy(), ref_x2) < 4e-7 and rel(y.numpy(), exact) < 2e-5, (rel(y.numpy(), ref_x2), rel(y.numpy(), exact))
    y2 = dev.empty((N, K, d.Ho, d.Wo))
    yq = D.QTensor.empty(dev, y2.shape, 'bf16x2') if ops.lp_q_direct(d, 0, 'bf16x2') else None
    ops.conv2d_fwd_lp_q(d, xq, wq, bd, y2, yq, 'bf16x2')
    assert np.array_equal(y2.numpy(), y.numpy())
    if yq is not None:
        for got, want in zip((yq.numpy(0), yq.numpy(1)), LP.split_bf16x2(y2.numpy())):
            assert np.array_equal(got, want)
    # data gradient
    if ops.split_supported(d, 1):
        wqT = dev.alloc(ops.split_weight_bytes(d, True, 2))
        ops.split_pack_weights(d, wp, wqT, True, 2)
        dx = dev.empty((N, C, H, W))
        ops.conv2d_dgrad_lp_q(d, dyq, wqT, dx, None, 'bf16x2')
        dx_x2, dW_x2 = LP.conv2d_vjp_x2(x, Wt, dy, s, pad)
        dx_exact = O.conv2d_vjp(np.zeros((N, C, H, W)), Wt.astype(np.float64), dy.astype(np.float64), s, pad)[0]
        assert rel(dx.numpy(), dx_x2) < 4e-7 and rel(dx.numpy(), dx_exact) < 2e-5, (rel(dx.numpy(), dx_x2), rel(dx.numpy(), dx_exact))
        dev.free(wqT)
    # weight gradient
    if ops.split_supported(d, 2):
        ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
        dw = dev.zeros((1, C * k * k * K, 1, 1))
        ops.conv2d_wgrad_lp_q(d, xq, dyq, dw, ws, 'bf16x2')
        got = D.unpack_conv_w(dw.numpy().ravel(), K, C, k, k)
        dW_x2 = LP.conv2d_vjp_x2(x, Wt, dy, s, pad)[1]
        dW_exact = O.conv2d_vjp(x.astype(np.float64), np.zeros((K, C, k, k)), dy.astype(np.float64), s, pad)[1]
        assert rel(got, dW_x2) < 6e-7 and rel(got, dW_exact) < 2e-5, (rel(got, dW_x2), rel(got, dW_exact))
        dev.free(ws)
    dev.free(wq)


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 64, 5, 1, 2), (3, 48, 32, 64, 40, 3, 1, 1), (2, 32, 64, 64, 128, 3, 2, 1), (3, 64, 16, 32, 96, 5, 1, 2)])
@pytest.mark.parametrize("pieces", [3, 2])
def test_persistent_tiles_are_bit_identical_to_one_tile_per_block(gpu, case, pieces):
    """sp_conv2_kernel walks several tiles per block when the launch has more tiles than blocks (GHM_SPLIT_PERSIST: blocks as a
    fraction of the CUs), its pipeline flowing across the tile boundary: the results must be bit for bit those of one tile per
    block -- forward with the q copy, data gradient, pooled forward; 8 blocks for 12-64 tiles here, uneven tile counts included"""
    dev, ops, D = gpu
    N, C, H, W, K, k, s, pad = case
    rng = np.random.RandomState(sum(case) + 17 * pieces)
    dt = 'bf16x3' if pieces == 3 else 'bf16x2'
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    x = rng.randn(N, C, H, W).astype(np.float32)
    dy = rng.randn(N, K, d.Ho, d.Wo).astype(np.float32)
    Wt = (rng.randn(K, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    xd, dyd, bd = dev.tensor(x), dev.tensor(dy), dev.tensor(b)
    wp = dev.tensor(D.pack_conv_w(Wt).ravel())
    wq = dev.alloc(ops.split_weight_bytes(d, False, pieces))
    wqT = dev.alloc(ops.split_weight_bytes(d, True, pieces))
    ops.split_pack_weights(d, wp, wq, False, pieces)
    ops.split_pack_weights(d, wp, wqT, True, pieces)
    xq, dyq = D.QTensor.empty(dev, x.shape, dt), D.QTensor.empty(dev, dy.shape, dt)
    ops.q_pack(xd, xq)
    ops.q_pack(dyd, dyq)
    res = {}
    for frac in ("0", "0.04"):
        with tuning_env(GHM_SPLIT_PERSIST=frac):
            y = dev.empty((N, K, d.Ho, d.Wo))
            yq = D.QTensor.empty(dev, y.shape, dt) if ops.lp_q_direct(d, 0, dt) else None
            ops.conv2d_fwd_lp_q(d, xq, wq, bd, y, yq, dt, 'lrelu', 0.2)
            out = [y.numpy()] + ([yq.numpy(p) for p in range(pieces)] if yq is not None else [])
            if s == 1 and ops.split_supported(d, 1):
                dx = dev.empty((N, C, H, W))
                ops.conv2d_dgrad_lp_q(d, dyq, wqT, dx, None, dt)
                out.append(dx.numpy())
            if s == 1 and d.Ho % 2 == 0 and ops.conv_pool_supported(d, 'lrelu', dt) == 2:
                pooled = dev.empty((N, K, d.Ho // 2, d.Wo // 2))
                mask = dev.alloc(N * K * (d.Ho // 2) * (d.Wo // 2))
                ops.conv2d_fwd_pool_lp_q(d, xq, wq, bd, pooled, None, mask, 'lrelu', 0.2, dt)
                m = np.empty(N * K * (d.Ho // 2) * (d.Wo // 2), np.uint8)
                dev.d2h(m, mask, m.nbytes)
                out += [pooled.numpy(), m]
                dev.free(mask)
            res[frac] = out
    assert len(res["0"]) == len(res["0.04"]) >= 2
    for a_, b_ in zip(res["0"], res["0.04"]):
        assert np.array_equal(a_, b_)
    dev.free(wq)
    dev.free(wqT)


@pytest.mark.parametrize("case", [(2, 64, 64, 32, 32), (1, 128, 128, 32, 64), (4, 256, 64, 128, 128), (2, 64, 64, 8, 32),
                                  (4, 256, 256, 32, 32)])
@pytest.mark.parametrize("dtype", ["bf16x3", "bf16x2"])
def test_collapsed_bilinear_convolution_skips_its_structural_zeros(gpu, case, dtype):
    """The coarse-grid form of BilinearUpsample2DLayer(2) -> 3x3 conv (p2p.py:204-267; csrc/conv_bilinear.hip) is a 3x3
    convolution with 4K filters in which 11 of the 36 collapsed taps are zero by construction.  ghm_blconv_{fwd,dgrad,wgrad}_split
    run the same kernels without the k-steps of those taps: the results must equal ghm_conv2d_*_split on the same zero-padded
    collapsed weights BIT FOR BIT (same products, same order), forward, data gradient (also accumulating, into a channel slice)
    and weight gradient, in both split arithmetic modes."""
    dev, ops, D = gpu
    N, C, K, n1, n2 = case
    rng = np.random.RandomState(sum(case))
    x = (rng.randn(N, C, n1, n2) * np.exp(rng.randn(N, C, 1, 1))).astype(np.float32)
    Wt = (rng.randn(K, C, 3, 3) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    g = (rng.randn(N, 4 * K, n1, n2) * np.exp(rng.randn(N, 4 * K, 1, 1))).astype(np.float32)
    d = D.conv_desc(N, C, n1, n2, 4 * K, 3, 3, 1, 1)
    for kind in (0, 1, 2):
        assert ops.blconv_split_supported(d, kind, dtype), kind
    wp3, bd = dev.tensor(D.pack_conv_w(Wt).ravel()), dev.tensor(b)
    wpc, b4 = dev.empty((1, C * 9 * 4 * K, 1, 1)), dev.empty((1, 4 * K, 1, 1))
    ops.upconv_collapse_batched(ops.collapse_table([(wp3, bd, wpc, b4, C, K, 1)]))
    w = wpc.numpy().reshape(C, 9, 4, K)
    zero = [(rs, pq) for rs in range(9) for pq in range(4) if not w[:, rs, pq].any()]
    assert len(zero) == 11                                      # 36 - (9 + 6 + 6 + 4)
    pieces = D.SPLITS[dtype]
    wq, wqT = dev.alloc(ops.split_weight_bytes(d, False, pieces)), dev.alloc(ops.split_weight_bytes(d, True, pieces))
    ops.split_pack_weights(d, wpc, wq, False, pieces)
    ops.split_pack_weights(d, wpc, wqT, True, pieces)
    xq, gq = D.QTensor.empty(dev, x.shape, dtype), D.QTensor.empty(dev, g.shape, dtype)
    ops.q_pack(dev.tensor(x), xq)
    ops.q_pack(dev.tensor(g), gq)
    # forward
    y0, y1 = dev.empty(g.shape), dev.empty(g.shape)
    ops.conv2d_fwd_lp_q(d, xq, wq, b4, y0, None, dtype)
    ops.blconv_fwd_split(d, xq, wq, b4, y1, dtype)
    # (4, 256, 256, 32, 32): few tiles -- the class forms (64-filter tiles; a forward launch of one block per CU halves every tile's
    # contraction so that each CU gets a heavy and a light block) cut the contraction into other split-K ranges than the plain
    # plans: the same products, fp32 sums in another order -- equal to rounding there, bit for bit everywhere else
    exact = case != (4, 256, 256, 32, 32)
    same = (lambda u, v: np.array_equal(u, v)) if exact else (lambda u, v: rel(u, v) < 1e-6)
    assert same(y0.numpy(), y1.numpy())
    # data gradient, plain and accumulating into a channel slice of a wider tensor
    dx0, dx1 = dev.empty(x.shape), dev.empty(x.shape)
    ops.conv2d_dgrad_lp_q(d, gq, wqT, dx0, None, dtype)
    ops.blconv_dgrad_split(d, gq, wqT, dx1, dtype)
    assert same(dx0.numpy(), dx1.numpy())
    wide0, wide1 = dev.tensor(rng.randn(N, C + 16, n1, n2).astype(np.float32)), None
    wide1 = dev.tensor(wide0.numpy())
    dv = D.conv_desc(N, C, n1, n2, 4 * K, 3, 3, 1, 1, wide0.nstride, gq.shape[1] * n1 * n2)
    ops.conv2d_dgrad_lp_q(dv, gq, wqT, wide0.channels(16, 16 + C), None, dtype, accumulate=True)
    ops.blconv_dgrad_split(dv, gq, wqT, wide1.channels(16, 16 + C), dtype, accumulate=True)
    assert same(wide0.numpy(), wide1.numpy())
    # weight gradient
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    dw0, dw1 = dev.zeros((1, C * 9 * 4 * K, 1, 1)), dev.zeros((1, C * 9 * 4 * K, 1, 1))
    ops.conv2d_wgrad_lp_q(d, xq, gq, dw0, ws, dtype)
    ops.blconv_wgrad_split(d, xq, gq, dw1, ws, dtype)
    a0, a1 = dw0.numpy().reshape(C, 9, 4, K), dw1.numpy().reshape(C, 9, 4, K)
    for rs in range(9):
        for pq in range(4):
            if (rs, pq) in zero:
                assert not a1[:, rs, pq].any()                  # skipped taps leave zeros (the expansion ignores them)
            else:
                assert np.array_equal(a0[:, rs, pq], a1[:, rs, pq]), (rs, pq)


@pytest.mark.parametrize("pieces", [3, 2])
def test_weight_gradient_with_two_rows_of_lookahead_is_bit_identical(gpu, pieces):
    """GHM_SPLIT_WGRAD_LA2=1 (3x3 stride 1, 32-pixel strips: the operands of output row i + 2 requested while row i is multiplied;
    built and measured in round 6, not the default): the same products in the same order as the one-row form -- plain and class
    form."""
    dev, ops, D = gpu
    dtype = {3: 'bf16x3', 2: 'bf16x2'}[pieces]
    N, C, K, n1, n2 = 2, 64, 64, 16, 64
    rng = np.random.RandomState(5)
    x = rng.randn(N, C, n1, n2).astype(np.float32)
    g = rng.randn(N, 4 * K, n1, n2).astype(np.float32)
    d = D.conv_desc(N, C, n1, n2, 4 * K, 3, 3, 1, 1)
    xq, gq = D.QTensor.empty(dev, x.shape, dtype), D.QTensor.empty(dev, g.shape, dtype)
    ops.q_pack(dev.tensor(x), xq)
    ops.q_pack(dev.tensor(g), gq)
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    out = {}
    for la2 in (False, True):
        with tuning_env(**({"GHM_SPLIT_WGRAD_LA2": "1"} if la2 else {})):
            a, b = dev.zeros((1, C * 9 * 4 * K, 1, 1)), dev.zeros((1, C * 9 * 4 * K, 1, 1))
            ops.conv2d_wgrad_lp_q(d, xq, gq, a, ws, dtype)
            ops.blconv_wgrad_split(d, xq, gq, b, ws, dtype)
            out[la2] = (a.numpy(), b.numpy())
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    assert np.abs(out[True][0]).max() > 0


@pytest.mark.parametrize("case", [(2, 64, 128, 32, 64, 3, 1), (2, 64, 128, 32, 64, 3, 2), (2, 32, 64, 16, 128, 5, 1), (3, 128, 64, 16, 32, 3, 1)])
def test_weight_gradient_blocks_renumbered_onto_one_xcd_are_the_same_blocks(gpu, case):
    """The (channel tile, filter tile) blocks of one strip are re-numbered so that one XCD's L2 serves all of them (round 6;
    GHM_SPLIT_WGRAD_NO_XCD=1 launches them as numbered by the grid): a permutation of WHICH block computes which tile --
    every tile computed once, bit-identical results, split-K partial slices included."""
    dev, ops, D = gpu
    N, C, K, H, W, ks, st = case
    rng = np.random.RandomState(11)
    Ho, Wo = (H + st - 1) // st, (W + st - 1) // st
    x = rng.randn(N, C, H, W).astype(np.float32)
    g = rng.randn(N, K, Ho, Wo).astype(np.float32)
    d = D.conv_desc(N, C, H, W, K, ks, ks, st, ks // 2)
    xq, gq = D.QTensor.empty(dev, x.shape, 'bf16x3'), D.QTensor.empty(dev, g.shape, 'bf16x3')
    ops.q_pack(dev.tensor(x), xq)
    ops.q_pack(dev.tensor(g), gq)
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    out = {}
    for plain in (False, True):
        with tuning_env(**({"GHM_SPLIT_WGRAD_NO_XCD": "1"} if plain else {})):
            a = dev.zeros((1, C * ks * ks * K, 1, 1))
            ops.conv2d_wgrad_lp_q(d, xq, gq, a, ws, 'bf16x3')
            out[plain] = a.numpy()
    assert np.array_equal(out[False], out[True])
    assert np.abs(out[True]).max() > 0


@pytest.mark.parametrize("case", [(2, 32, 64, 16, 64, 'bf16x3', False), (2, 32, 64, 16, 64, 'bf16x3', True), (2, 64, 128, 32, 32, 'bf16x3', True),
                                  (1, 32, 64, 8, 128, 'bf16x2', True), (3, 64, 64, 20, 96, 'bf16x3', True), (2, 32, 128, 64, 64, 'bf16x2', False),
                                  (2, 96, 192, 16, 64, 'bf16x3', True), (1, 32, 64, 4, 32, 'bf16x3', 'all')])
def test_pooled_weight_gradient_on_the_sparse_matrix_instruction(gpu, case):
    """5x5 conv -> activation -> MaxPool2D(2) (architectures/dcgan.py:42-60): the weight gradient contracts x with the max-pool
    backward of the pooled gradient -- one non-zero per window row unless two columns tie.  ghm_maxpool2_mask_bwd_compress_q writes
    it as half-width rows + column bits + tie flags, ghm_conv2d_wgrad_pooled_split contracts tie-free rows on
    v_smfmac_f32_32x32x32_bf16 and flagged rows densely.  Against the dense split kernel (same products; the instruction's own
    summation order differs) and a float64 contraction of the dense gradient; the flags against the mask."""
    dev, ops, D = gpu
    N, C, K, H, W, dtype, ties = case
    rng = np.random.RandomState(sum(case[:5]))
    Ho, Wo = H // 2, W // 2
    x = rng.randn(N, C, H, W).astype(np.float32)
    gp = rng.randn(N, K, Ho, Wo).astype(np.float32)
    mask = (1 << rng.randint(0, 4, size=(N, K, Ho, Wo))).astype(np.uint8)          # bit 2 r + c: one arg-max per window
    if ties == 'all':       # flat terrain everywhere: every window ties, every row runs densely
        mask[:] = 0b1111
    elif ties:      # tied window rows and whole tied windows in three pooled rows
        for (n, i) in [(0, 1), (N - 1, Ho - 1), (0, Ho // 2)]:
            mask[n, ::3, i, ::5] = 0b0011
            mask[n, 1::3, i, 1::7] = 0b1111
    d = D.conv_desc(N, C, H, W, K, 5, 5, 1, 2)
    assert ops.wgrad_pooled_split_supported(d, dtype)
    xq = D.QTensor.empty(dev, x.shape, dtype)
    ops.q_pack(dev.tensor(x), xq)
    mptr = dev.alloc(mask.size)
    dev.h2d(mptr, mask)
    gpt = dev.tensor(gp)
    dyq, dx = D.QTensor.empty(dev, (N, K, H, W), dtype), dev.empty((N, K, H, W))
    ops.maxpool2_mask_bwd_q(mptr, None, gpt, dx, dyq, 'linear', 0.0)
    cq = D.QTensor.empty(dev, (N, K, H, W // 2), dtype)
    idx, flags = dev.alloc(N * (K // 8) * H * (W // 32) * 16 + 256), dev.alloc(N * H * 4 + 256)
    ops.maxpool2_mask_bwd_compress_q(mptr, None, gpt, cq, idx, flags, 'linear', 0.0)
    fl = np.zeros(N * H, np.int32)
    dev.sync()
    dev.d2h(fl, flags, fl.nbytes)
    want = np.zeros((N, H), bool)
    want[:, 0::2] = ((mask & 3) == 3).any(axis=(1, 3))
    want[:, 1::2] = (((mask >> 2) & 3) == 3).any(axis=(1, 3))
    assert np.array_equal(fl.reshape(N, H) != 0, want)
    assert want.any() == bool(ties) and (ties != 'all' or want.all())
    dyd = dx.numpy()
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    a, b = dev.zeros((1, C * 25 * K, 1, 1)), dev.zeros((1, C * 25 * K, 1, 1))
    ops.conv2d_wgrad_lp_q(d, xq, dyq, a, ws, dtype)
    ops.conv2d_wgrad_pooled_split(d, xq, dyq, cq, idx, flags, b, ws, dtype)
    A, B = a.numpy().ravel(), b.numpy().ravel()
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (2, 2), (2, 2)))
    ref = np.zeros((C, 25, K))
    for ta in range(5):
        for tb in range(5):
            ref[:, ta * 5 + tb, :] = np.einsum('nchw,nkhw->ck', xp[:, :, ta:ta + H, tb:tb + W], dyd.astype(np.float64))
    ref = ref.ravel()
    sc = np.abs(ref).max()
    bound = 2e-6 if dtype == 'bf16x3' else 3e-5          # the dense kernel's own distance from float64: 4e-7 / 5e-6 here
    assert np.abs(A - ref).max() / sc < bound
    assert np.abs(B - ref).max() / sc < bound
    assert np.abs(A - B).max() / sc < bound
    # accumulate into an existing gradient
    ops.conv2d_wgrad_pooled_split(d, xq, dyq, cq, idx, flags, b, ws, dtype, accumulate=True)
    assert np.abs(b.numpy().ravel() - 2 * B).max() / sc < 1e-6
