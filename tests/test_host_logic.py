"""Host-side logic without a GPU: layer vocabulary, weight layout, lowering / rewrites / placement and program
emission (against a recording fake device), the iterator, and the rendezvous / sharding helpers."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from gan_heightmaps_amd import layers as L
from gan_heightmaps_amd import device as D
from gan_heightmaps_amd.architectures import dcgan, p2p
from gan_heightmaps_amd.engine import NetPlan, ParamStore
from gan_heightmaps_amd.nonlinearities import linear, tanh
from gan_heightmaps_amd.step import GanStep
from gan_heightmaps_amd import updates, dist, experiments
from tests.fake_device import FakeDevice, RecordingOps


def test_pack_unpack_roundtrip_and_meaning():
    rng = np.random.RandomState(0)
    W = rng.randn(6, 4, 3, 5).astype(np.float32)
    wp = D.pack_conv_w(W)
    assert wp.shape == (4, 15, 6)
    assert np.array_equal(D.unpack_conv_w(wp, 6, 4, 3, 5), W)
    # wp[c][a*kw+b][k] = W[k][c][kh-1-a][kw-1-b]
    assert wp[2, 1 * 5 + 3, 4] == W[4, 2, 3 - 1 - 1, 5 - 1 - 3]


def test_layer_vocabulary_shapes_and_param_io():
    x = L.InputLayer((None, 3, 16, 16))
    c = L.Conv2DLayer(x, num_filters=8.0, filter_size=3, stride=2, pad='same', nonlinearity=linear)
    assert c.output_shape == (None, 8, 8, 8)
    with pytest.raises(ValueError):
        L.Conv2DLayer(x, num_filters=2.5, filter_size=3)
    t = L.Deconv2DLayer(c, num_filters=5, filter_size=(2, 2), stride=(2, 2), nonlinearity=linear)
    assert t.output_shape == (None, 5, 16, 16) and t.W.shape == (8, 5, 2, 2)
    cat = L.ConcatLayer([t, x])
    assert cat.output_shape == (None, 8, 16, 16)
    vals = L.get_all_param_values(cat)
    vals[0] = vals[0] + 1
    L.set_all_param_values(cat, vals)
    assert np.array_equal(L.get_all_param_values(cat)[0], vals[0])
    with pytest.raises(ValueError):
        L.set_all_param_values(cat, vals[:-1])
    assert L.count_params(cat) == 8 * 3 * 9 + 8 + 8 * 5 * 4 + 5
    r = L.ReshapeLayer(L.DenseLayer(L.InputLayer((None, 10)), 32, nonlinearity=linear), (-1, 2, 4, 4))
    assert r.output_shape == (None, 2, 4, 4)


def _plan(net, batch, **kw):
    dev = FakeDevice()
    ops = RecordingOps(dev)
    store = ParamStore(dev, L.get_all_params(net))
    return NetPlan(dev, ops, net, batch, store, **kw), ops, dev


def test_unet_lowering_rewrites_and_concat_in_place():
    net = p2p.g_unet(64, True, False, nf=4, act=tanh, bilinear_upsample=True)
    plan, ops, dev = _plan(net, 2)
    kinds = [n.op for n in plan.order]
    # every leaky_rectify was folded into a BN epilogue; no standalone activation kernels remain
    assert 'act' not in kinds
    assert plan.out_node.op == 'deconv' and plan.out_node.act == tanh
    bns = [n for n in plan.order if n.op == 'bn']
    assert all(n.act.kind == 'lrelu' and abs(n.act.alpha - 0.01) < 1e-9 for n in bns)
    cats = [n for n in plan.order if n.op == 'concat']
    assert len(cats) == 5
    for c in cats:
        # both inputs live inside the concat buffer (no copies), at the right channel offsets
        assert all(i.alias is not None and i.alias[0] is c for i in c.inputs)
        a, b = c.inputs
        assert a.out.ptr == c.out.ptr and b.out.ptr == c.out.ptr + 4 * a.shape[1] * a.shape[2] * a.shape[3]
        assert a.out.nstride == c.shape[1] * c.shape[2] * c.shape[3]
    prog = []
    plan.emit_forward(prog)
    labels = [e[0] for e in prog]
    assert 'concat_copy' not in labels and labels.count('up_bilinear_fwd') == 4
    assert labels.count('bn_stats') == labels.count('bn_apply') == len(bns)
    seed = dev.empty(plan.out.shape)
    bwd = []
    plan.emit_backward(bwd, seed)
    bl = [e[0] for e in bwd]
    assert bl.count('bn_bwd') == len(bns) and 'concat_bwd_copy' not in bl
    # encoder convs accumulate their data gradient into the skip slice the decoder already wrote
    run = []
    for e in bwd:
        e[1]()
    acc_calls = [c for c in ops.calls if c[0] in ('conv2d_dgrad', 'conv2d_dgrad_t') and c[1][-1] is True]
    assert len(acc_calls) == 5      # conv2..conv5 and the 2x2 bottleneck conv accumulate into the 5 skip slices


def test_dcgan_disc_lowering_fuses_lrelu_into_conv_and_pool_backward():
    net = dcgan.default_discriminator(32, True, nch=16, div=[4, 2, 2], nonlinearity=linear)
    plan, ops, dev = _plan(net, 4)
    convs = [n for n in plan.order if n.op == 'conv']
    assert [n.act.kind for n in convs] == ['lrelu', 'lrelu', 'lrelu', 'relu']
    prog = []
    gin = plan.emit_backward(prog, dev.empty((2, 1, 1, 1)), nslice=(2, 4), wgrad=False,
                             input_grads=[plan.input_nodes[0].layer], tag="gloss")
    labels = [e[0] for e in prog]
    assert 'conv_wgrad' not in labels and 'bias_grad' not in labels
    assert labels.count('maxpool_bwd') == 3 and labels.count('act_bwd') == 1      # only d_out's ReLU is separate
    g = gin[plan.input_nodes[0].layer]
    assert g.shape == (2, 1, 32, 32)


def test_gan_step_program_structure_on_fake_device():
    dev = FakeDevice()
    G = dcgan.default_generator(24, True, nch=16, div=[2, 2, 4])
    Dn = dcgan.default_discriminator(32, True, nch=16, div=[4, 2, 2], nonlinearity=linear)
    U = p2p.g_unet(32, True, False, nf=4, act=tanh, bilinear_upsample=True)
    P = p2p.discriminator(32, True, False, nf=4, act=linear, mul_factor=[1, 2])
    spec = updates.rmsprop(learning_rate=updates.shared(1e-4))
    import gan_heightmaps_amd.step as step_mod
    orig = step_mod.Ops
    step_mod.Ops = RecordingOps
    try:
        eng = GanStep(dev, G, Dn, U, P, 100, True, 'l1', spec, 'both', use_graph=False, two_streams=False)
        b = eng.built(4)
    finally:
        step_mod.Ops = orig
    # G writes straight into the fake half of D's input batch; U into channels 1..3 of the PatchGAN pair buffer
    assert b.G.out.ptr == b.d_in.ptr + 4 * 4 * 32 * 32 and b.D.batch == 8
    assert b.U.out.nstride == 4 * 32 * 32 and b.U.out.shape == (4, 3, 32, 32)
    # one launch list per stream: [DCGAN stage, pix2pix stage]
    la, lb = ([e[0] for e in lane] for lane in b.train_compute)
    assert la.count('loss') == 3 and lb.count('loss') == 3 and lb.count('recon') == 1 and 'recon' not in la
    assert [[e[0] for e in lane] for lane in b.update] == [['rmsprop_dcgan_gen', 'rmsprop_dcgan_disc'],
                                                           ['rmsprop_p2p_gen', 'rmsprop_p2p_disc']]
    assert b.exchange == []
    # D is differentiated twice: once with weight gradients (2B batch), once data-gradient only (fake half)
    # (the generator's Upscale2D -> 5x5 convs run as collapsed 3x3 convs: 'upconv_wgrad')
    d_wgrads = sum(1 for lane in b.train_compute for e in lane if e[0] in ('conv_wgrad', 'upconv_wgrad'))
    assert sum(1 for lane in b.train_compute for e in lane if e[0] == 'upconv_wgrad') == 3
    n_convs = sum(1 for net in (G, Dn, U, P["out"]) for l in L.get_all_layers(net)
                  if isinstance(l, L.Conv2DLayer))
    assert d_wgrads == n_convs


def test_array_iterator_normalisation_and_layout():
    X, Y = experiments.synthetic_arrays(6, 8, True, False, seed=1)
    it = experiments.ArrayIterator(X, Y, 4, True, False)
    assert it.N == 6
    x, y = next(it)
    x2, y2 = it.next()          # py2-style spelling used by the reference loop
    # slices are shuffled per pass (util.py:24-26); one of the two is the ragged tail util._get_slices produces
    assert sorted([x.shape[0], x2.shape[0]]) == [2, 4]
    assert x.shape[1:] == (1, 8, 8) and y.shape[1:] == (3, 8, 8) and x.dtype == np.float32
    assert 0 <= x.min() and x.max() <= 1 and -1 <= y.min() and y.max() <= 1
    assert next(it)[0].shape[0] in (2, 4)      # infinite generator: a new shuffled pass starts


def test_shard_batch():
    a = np.arange(24).reshape(8, 3)
    parts = [dist.shard_batch([a], r, 4)[0] for r in range(4)]
    assert np.array_equal(np.concatenate(parts), a) and parts[2][0, 0] == 12
    with pytest.raises(ValueError):
        dist.shard_batch([a], 0, 3)


def _rdzv_worker(rank, path, q):
    uid = dist.exchange_unique_id(rank, 2, lambda: bytes(range(128)), path=path, timeout=30)
    q.put((rank, uid))


def test_unique_id_rendezvous_two_processes(tmp_path):
    path = str(tmp_path / "uid")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_worker, args=(r, path, q)) for r in (1, 0)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    assert got[0] == got[1] == bytes(range(128))
