"""SURVEY 8 f4 on the device: the architecture variants no registered experiment uses -- ``g_unet_256``
(architectures/p2p.py:29-122), ``discriminator2`` (:294-308), ``num_repeats > 0`` (dcgan.py:21, p2p.py:148-149,
:284-288) and ``pool_mode='average_inc_pad'`` (dcgan.py:48-49) -- lowered by the engine and run through libghm.so:
forward, every parameter gradient and the input gradient against the layer-graph interpreter on the oracle's ops
(tests/golden/symtheano.py, float64).  Tolerances: output <= 1e-5, gradients <= 5e-4 (fp32 kernels; north_star 1e-3).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.fixture(scope="module")
def gpu():
    from gan_heightmaps_amd import device
    if device.device_count() == 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    dev = device.Device(0)
    yield dev, device.Ops(dev)
    dev.close()


def _variant(name):
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.nonlinearities import linear, sigmoid, tanh
    rng = np.random.RandomState(3)
    if name == "g_unet_256":
        net = p2p.g_unet_256(256, True, False, nf=4, act=tanh)
        return net, None, {None: rng.rand(2, 1, 256, 256)}
    if name == "discriminator2":
        d = p2p.discriminator2(64, True, False, nf=8, act=linear, mul_factor=[1, 2, 4])
        return d["out"], d["inputs"], {0: rng.rand(3, 1, 64, 64), 1: rng.randn(3, 3, 64, 64)}
    if name == "patchgan_num_repeats":
        d = p2p.discriminator(32, True, False, nf=8, act=sigmoid, mul_factor=[1, 2], num_repeats=1, bn=True)
        return d["out"], d["inputs"], {0: rng.rand(4, 1, 32, 32), 1: rng.randn(4, 3, 32, 32)}
    if name == "unet_num_repeats":
        return p2p.g_unet(32, True, False, nf=4, act=tanh, num_repeats=1, bilinear_upsample=True), None, \
            {None: rng.rand(3, 1, 32, 32)}
    if name == "dcgan_gen_num_repeats":
        return dcgan.default_generator(20, True, nch=16, div=[2, 4], num_repeats=1), None, {None: rng.rand(4, 20)}
    if name == "dcgan_disc_num_repeats_avgpool":
        return dcgan.default_discriminator(32, True, nch=16, div=[4, 2, 2], num_repeats=1, bn=True,
                                           pool_mode='average_inc_pad', nonlinearity=linear), None, \
            {None: rng.rand(4, 1, 32, 32)}
    if name == "dcgan_disc_avgpool":
        return dcgan.default_discriminator(64, False, nch=32, div=[4, 2, 2, 1], pool_mode='average_inc_pad',
                                           nonlinearity=sigmoid), None, {None: rng.rand(2, 3, 64, 64)}
    raise KeyError(name)


@pytest.mark.parametrize("name", ["g_unet_256", "discriminator2", "patchgan_num_repeats", "unet_num_repeats",
                                  "dcgan_gen_num_repeats", "dcgan_disc_num_repeats_avgpool", "dcgan_disc_avgpool"])
def test_architecture_variant_on_the_device(gpu, name):
    import symtheano as ST
    from oracle import tape as TP
    from gan_heightmaps_amd import init as INIT, layers as L
    from gan_heightmaps_amd.engine import NetPlan, ParamStore
    dev, ops = gpu
    INIT.set_rng(np.random.RandomState(11))
    net, in_layers, feeds = _variant(name)
    if in_layers is None:
        in_layers = [l for l in L.get_all_layers(net) if isinstance(l, L.InputLayer)]
        feeds = {0: feeds[None]}
    B = feeds[0].shape[0]
    store = ParamStore(dev, L.get_all_params(net))
    plan = NetPlan(dev, ops, net, B, store, name=name)
    fwd, bwd = [], []
    plan.emit_forward(fwd)
    seed = np.random.RandomState(2).randn(*plan.out.shape).astype(np.float32)
    seed_d = dev.tensor(seed)
    want_in = [l for l in in_layers if len(l.shape) == 4]          # image inputs: their gradient is checked too
    gin = plan.emit_backward(bwd, seed_d, input_grads=want_in)
    for i, l in enumerate(in_layers):
        t = plan.input_tensor(l)
        t.set(np.asarray(feeds[i], np.float32).reshape(t.shape))
    for e in fwd + bwd:
        e[1]()
    dev.sync()
    # ---- the same graph on the oracle's ops, float64 ----
    env = {"in%d" % i: np.asarray(feeds[i], np.float32) for i in range(len(in_layers))}
    c = ST.Ctx(env, np.float64)
    sym_in = {l: ST.placeholder("in%d" % i) for i, l in enumerate(in_layers)}
    ref = ST.get_output(net, sym_in).ev(c)
    assert rel(plan.out.numpy().reshape(ref.v.shape), ref.v) < 1e-5, name
    TP.backward(ref, seed.astype(np.float64).reshape(ref.v.shape))
    checked = 0
    for p in L.get_all_params(net, trainable=True):
        g_ref = c.param(p).g
        if g_ref is None or np.linalg.norm(g_ref) < 1e-9:
            continue                       # e.g. a conv bias that feeds a BatchNorm: exactly zero
        assert rel(store.download_grad(p), g_ref) < 5e-4, (name, p.name, p.shape)
        checked += 1
    assert checked >= 4, name
    for i, l in enumerate(in_layers):
        if l in want_in:
            g_ref = sym_in[l].ev(c).g
            assert g_ref is not None and np.linalg.norm(g_ref) > 0
            assert rel(gin[l].numpy(), g_ref) < 5e-4, (name, "input", i)
    # BatchNorm running statistics took the lasagne update (alpha 0.1) from the batch statistics
    for l, mu, inv in c.bn:
        assert rel(l.mean.get_value(), 0.9 * 0.0 + 0.1 * mu.ravel()) < 1e-4 or np.abs(mu).max() < 1e-6, name
        assert rel(l.inv_std.get_value(), 0.9 * 1.0 + 0.1 * inv.ravel()) < 1e-4, name
    # deterministic pass (gen_fn_det / z_fn_det path): running statistics instead of batch statistics
    det = []
    plan.emit_forward(det, deterministic=True)
    for e in det:
        e[1]()
    c2 = ST.Ctx(env, np.float64)
    ref_det = ST.get_output(net, sym_in, deterministic=True).ev(c2)
    assert rel(plan.out.numpy().reshape(ref_det.v.shape), ref_det.v) < 1e-5, name
