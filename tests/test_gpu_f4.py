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


def _instance_norm_net(kind):
    """small nets with InstanceNormLayer in the places the reference's nets have BatchNormLayer"""
    from gan_heightmaps_amd import layers as L, nonlinearities as NL
    rng = np.random.RandomState(5)
    if kind == "encoder":            # conv (stride 2) -> IN -> leaky -> conv -> IN -> tanh   (p2p.py's encoder block shape)
        i = L.InputLayer((None, 3, 32, 32))
        c = L.Conv2DLayer(i, 16, 3, stride=2, pad=1, nonlinearity=NL.linear)
        c = L.NonlinearityLayer(L.InstanceNormLayer(c), NL.leaky_rectify)
        c = L.Conv2DLayer(c, 24, 3, stride=1, pad=1, nonlinearity=NL.linear)
        net = L.NonlinearityLayer(L.InstanceNormLayer(c), NL.tanh)
        return net, {None: rng.randn(3, 3, 32, 32)}
    if kind == "upsample":           # Upscale2D -> 5x5 conv (collapsed to 3x3 by the lowering) -> IN -> rectify   (dcgan.py:22-24)
        i = L.InputLayer((None, 8, 8, 8))
        c = L.Conv2DLayer(L.Upscale2DLayer(i, 2), 16, 5, pad='same', nonlinearity=NL.linear)
        c = L.NonlinearityLayer(L.InstanceNormLayer(c), NL.rectify)
        net = L.Conv2DLayer(c, 8, 3, pad=1, nonlinearity=NL.linear)
        return net, {None: rng.randn(2, 8, 8, 8)}
    raise KeyError(kind)


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
@pytest.mark.parametrize("kind", ["encoder", "upsample"])
def test_instance_norm_layer_on_the_device(gpu, kind, dtype):
    """InstanceNormLayer (north_star names InstanceNorm beside BatchNorm; the reference has no call site) lowered by the
    engine -> ghm_instance_norm_fwd / _bwd, against the layer-graph interpreter on the oracle's op (float64; the op itself is
    pinned on torch in tests/test_oracle_vs_torch.py::test_instancenorm): output <= 1e-5, parameter and input gradients <= 5e-4,
    and the deterministic pass equals the training pass (no running statistics)"""
    import symtheano as ST
    from oracle import tape as TP
    from gan_heightmaps_amd import init as INIT, layers as L
    from gan_heightmaps_amd.engine import NetPlan, ParamStore
    dev, ops = gpu
    INIT.set_rng(np.random.RandomState(13))
    net, feeds = _instance_norm_net(kind)
    in_layers = [l for l in L.get_all_layers(net) if isinstance(l, L.InputLayer)]
    x0 = np.asarray(feeds[None], np.float32)
    B = x0.shape[0]
    params = L.get_all_params(net)
    for p in params:                     # non-trivial gamma / beta
        if p.name in ("gamma", "beta"):
            p.set_value((p.get_value() + 0.3 * np.random.RandomState(len(p.name)).randn(*p.get_value().shape)).astype(np.float32))
    store = ParamStore(dev, params)
    plan = NetPlan(dev, ops, net, B, store, name="in_" + kind, dtype=dtype)
    fwd, bwd, det = [], [], []
    plan.emit_forward(fwd)
    seed = np.random.RandomState(2).randn(*plan.out.shape).astype(np.float32)
    gin = plan.emit_backward(bwd, dev.tensor(seed), input_grads=in_layers)
    plan.input_tensor(in_layers[0]).set(x0)
    for e in fwd + bwd:
        e[1]()
    dev.sync()
    out = plan.out.numpy().copy()
    env = {"in0": x0}
    c = ST.Ctx(env, np.float64)
    sym_in = {in_layers[0]: ST.placeholder("in0")}
    ref = ST.get_output(net, sym_in).ev(c)
    assert rel(out.reshape(ref.v.shape), ref.v) < 1e-5
    TP.backward(ref, seed.astype(np.float64).reshape(ref.v.shape))
    checked = 0
    for p in L.get_all_params(net, trainable=True):
        g_ref = c.param(p).g
        if g_ref is None or np.linalg.norm(g_ref) < 1e-9:
            continue                     # a conv bias that feeds an InstanceNorm: exactly zero
        assert rel(store.download_grad(p), g_ref) < 5e-4, (p.name, rel(store.download_grad(p), g_ref))
        checked += 1
    assert checked >= 5
    g_in = sym_in[in_layers[0]].ev(c).g
    assert g_in is not None and np.linalg.norm(g_in) > 0
    assert rel(gin[in_layers[0]].numpy(), g_in) < 5e-4
    plan.emit_forward(det, deterministic=True)
    for e in det:
        e[1]()
    dev.sync()
    assert np.array_equal(plan.out.numpy(), out)
