"""The oracle against the committed golden vectors (CPU) -- and the HIP path against the same vectors (GPU)."""
import os

import numpy as np
import pytest

from oracle import ops as O
from oracle import step as S
from tests.golden.make_golden import SMALL, OP_CASES, DECONV_CASES, make_step

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.fixture(scope="module")
def ops_vec():
    return dict(np.load(os.path.join(G, "ops.npz")))


def test_oracle_reproduces_op_vectors(ops_vec):
    v = ops_vec
    for name, (N, C, H, W, K, k, s, pad) in OP_CASES.items():
        x, Wt, b, dy = (v["%s.%s" % (name, n)].astype(np.float64) for n in ("x", "W", "b", "dy"))
        assert rel(O.conv2d_fwd(x, Wt, b, s, pad), v[name + ".y"]) < 1e-6
        dx, dW, db = O.conv2d_vjp(x, Wt, dy, s, pad)
        assert rel(dx, v[name + ".dx"]) < 1e-6 and rel(dW, v[name + ".dW"]) < 1e-6 and rel(db, v[name + ".db"]) < 1e-6
    for name, (N, Ci, h, w, Co, k, s) in DECONV_CASES.items():
        x, Wt, b, dy = (v["%s.%s" % (name, n)].astype(np.float64) for n in ("x", "W", "b", "dy"))
        assert rel(O.deconv2d_fwd(x, Wt, b, s), v[name + ".y"]) < 1e-6
        dx, dW, db = O.deconv2d_vjp(x, Wt, dy, s)
        assert rel(dx, v[name + ".dx"]) < 1e-6 and rel(dW, v[name + ".dW"]) < 1e-6
    for name in ("bn_4d", "bn_bottleneck", "bn_dense"):
        x, beta, gamma, dy = (v["%s.%s" % (name, n)].astype(np.float64) for n in ("x", "beta", "gamma", "dy"))
        y, mu, inv = O.bn_train_fwd(x, beta, gamma)
        assert rel(y, v[name + ".y"]) < 1e-5 and rel(inv, v[name + ".inv"]) < 1e-6
        dx, dbeta, dgamma = O.bn_train_vjp(x, gamma, mu, inv, dy)
        assert rel(dx, v[name + ".dx"]) < 1e-5 and rel(dgamma, v[name + ".dgamma"]) < 1e-5
    for name in ("bilinear_5x3", "bilinear_2x2", "bilinear_1x1"):
        assert rel(O.bilinear_up2_fwd(v[name + ".x"].astype(np.float64)), v[name + ".y"]) < 1e-6
        assert rel(O.bilinear_up2_vjp(v[name + ".g"].astype(np.float64)), v[name + ".dx"]) < 1e-6


@pytest.mark.parametrize("opt,lr", [("rmsprop", 1e-4), ("adam", 1e-3)])
def test_oracle_reproduces_step_trajectory(opt, lr):
    want = np.load(os.path.join(G, "step_%s.npz" % opt))
    got = make_step(opt, lr)
    for k in ("losses", "grad_norms", "param_sums", "param_abs_sums"):
        assert np.allclose(got[k], want[k], rtol=1e-9, atol=1e-12), k


@pytest.mark.gpu
def test_hip_ops_match_golden_vectors(ops_vec):
    from gan_heightmaps_amd import device as D
    dev = D.Device(0)
    ops = D.Ops(dev)
    v = ops_vec
    try:
        for name, (N, C, H, W, K, k, s, pad) in OP_CASES.items():
            d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
            xd, wd, bd = dev.tensor(v[name + ".x"]), dev.tensor(D.pack_conv_w(v[name + ".W"]).ravel()), dev.tensor(v[name + ".b"])
            yd = dev.empty(v[name + ".y"].shape)
            ops.conv2d_fwd(d, xd, wd, bd, yd)
            assert rel(yd.numpy(), v[name + ".y"]) < 1e-5
            dyd, dxd = dev.tensor(v[name + ".dy"]), dev.empty(v[name + ".x"].shape)
            ops.conv2d_dgrad(d, dyd, wd, dxd)
            assert rel(dxd.numpy(), v[name + ".dx"]) < 1e-5
            dwd = dev.zeros((1, C * k * k * K, 1, 1))
            ops.conv2d_wgrad(d, xd, dyd, dwd, dev.alloc(ops.wgrad_workspace(d)))
            assert rel(D.unpack_conv_w(dwd.numpy().ravel(), K, C, k, k), v[name + ".dW"]) < 1e-5
        for name, (N, Ci, h, w, Co, k, s) in DECONV_CASES.items():
            y = v[name + ".y"]
            d = D.conv_desc(N, Co, y.shape[2], y.shape[3], Ci, k, k, s, 0)
            xd, wd, bd = dev.tensor(v[name + ".x"]), dev.tensor(D.pack_conv_w(v[name + ".W"]).ravel()), dev.tensor(v[name + ".b"])
            yd = dev.empty(y.shape)
            ops.conv2d_dgrad(d, xd, wd, yd, bias=bd)
            assert rel(yd.numpy(), y) < 1e-5
        for name in ("bn_4d", "bn_bottleneck", "bn_dense"):
            x = v[name + ".x"]
            C = x.shape[1]
            xd = dev.tensor(x)
            md, ivd = dev.empty((1, C, 1, 1)), dev.empty((1, C, 1, 1))
            rm, ri = dev.tensor(np.zeros(C, np.float32)), dev.tensor(np.ones(C, np.float32))
            ws = dev.alloc(ops.bn_workspace(C))
            ops.bn_stats(xd, md, ivd, ws, rm, ri)
            assert rel(ivd.numpy().ravel(), v[name + ".inv"]) < 1e-5 and rel(ri.numpy().ravel(), v[name + ".run_inv"]) < 1e-5
            yd = dev.empty(xd.shape)
            ops.bn_apply(xd, yd, md, ivd, dev.tensor(v[name + ".gamma"]), dev.tensor(v[name + ".beta"]))
            assert rel(yd.numpy().reshape(x.shape), v[name + ".y"]) < 1e-5
        for name in ("bilinear_5x3", "bilinear_2x2", "bilinear_1x1"):
            x = v[name + ".x"]
            yd = dev.empty(v[name + ".y"].shape)
            ops.upsample_bilinear2_fwd(dev.tensor(x), yd)
            assert rel(yd.numpy(), v[name + ".y"]) < 1e-6
            dxd = dev.empty(x.shape)
            ops.upsample_bilinear2_bwd(dev.tensor(v[name + ".g"]), dxd)
            assert rel(dxd.numpy(), v[name + ".dx"]) < 1e-5
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("opt,lr", [("rmsprop", 1e-4), ("adam", 1e-3)])
def test_hip_step_matches_golden_trajectory(opt, lr):
    """3 consecutive train_fn calls from the seeded init, with NO resync to the oracle: losses within 1e-3 rel
    (north_star tolerance) of the float64 trajectory stored in tests/golden/step_*.npz"""
    from gan_heightmaps_amd import device as D
    from tests.test_gpu_step import build_model
    want = np.load(os.path.join(G, "step_%s.npz" % opt))
    cfg = S.default_cfg(opt=opt, lr=lr, **SMALL)
    dev = D.Device(0)
    try:
        model = build_model(cfg, 7, dev)
        for it in range(3):
            Z, X, Y = S.synthetic_batch(4, cfg, seed=100 + it)
            got = model.train_fn(Z, X, Y)
            assert rel(got, want["losses"][it]) < 1e-3, (it, got, want["losses"][it])
    finally:
        dev.close()
