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
    assert labels.count('bn_fwd') == len(bns)                   # statistics + normalise + activation: one entry point
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


def test_input_pipeline_protocol_on_fake_device():
    """GanStep.train_pipelined without a GPU: consecutive steps alternate between two plans, the upload of batch i+1 is
    issued after step i was enqueued and before its losses are read, a plan's inputs are overwritten only after a host-side
    wait for the step that last used it, the stage stream waits for the event behind ITS batch's upload, and a page-locked set
    is reused only after its upload has passed.  (Bit-identity with the sequential loop is the GPU test
    test_pipelined_upload_is_bit_identical_to_the_sequential_loop.)"""
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
        FakeDevice.pipe_log.clear()
        drawn = []

        def batches():
            for i in range(5):
                drawn.append(i)
                yield (np.full((4, 24), i, np.float32), np.full((4, 1, 32, 32), i, np.float32), np.full((4, 3, 32, 32), i, np.float32))
        seen = []
        for i, losses in enumerate(eng.train_pipelined(batches())):
            assert len(losses) == 5
            seen.append(list(drawn))          # which batches had been drawn when step i's losses came back
    finally:
        step_mod.Ops = orig
    assert seen == [[0, 1], [0, 1, 2], [0, 1, 2, 3], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4]]     # one batch ahead, never more
    b0, b1 = eng.built(4, 0), eng.built(4, 1)
    assert b0 is not b1 and b0.x.ptr != b1.x.ptr and b0.G.store is b1.G.store            # own inputs, shared parameters
    log = FakeDevice.pipe_log
    ups = [e for e in log if e[0] == "h2d_async"]
    assert [e[2] for e in ups[0::3]] == [0.0, 1.0, 2.0, 3.0, 4.0]                        # z of batch i ...
    assert [e[1] for e in ups[0::3]] == [b0.z.ptr, b1.z.ptr, b0.z.ptr, b1.z.ptr, b0.z.ptr]   # ... into alternating plans
    # before the THIRD upload (plan 0 again) the host waited for the events recorded behind step 0 on that plan, and for the
    # event behind the first upload from the same page-locked set
    i3 = log.index(ups[6])
    syncs = [e for e in log[:i3] if e[0] == "host_sync"]
    recs0 = [e[1] for e in log[:i3] if e[0] == "record"]
    assert syncs and all(e[1] in recs0 for e in syncs)
    # every step's stage stream waits for an event recorded (on the copy stream) after that batch's three uploads
    waits = [i for i, e in enumerate(log) if e[0] == "wait"]
    assert len(waits) == 5
    for k, wi in enumerate(waits):
        ev = log[wi][1]
        ri = max(i for i, e in enumerate(log[:wi]) if e[0] == "record" and e[1] == ev)
        assert sum(1 for e in log[:ri] if e[0] == "h2d_async") == 3 * (k + 1)
    eng.close_pipeline()


def test_resident_batches_rotate_through_two_plans_on_fake_device():
    """bench.py's timed loop (no reference counterpart; the contract is "inputs resident in HBM, every step a different
    batch"): GanStep.upload_resident_async copies batch k+1 device-to-device into the OTHER plan's inputs on the copy stream
    while step k runs; the stage stream of step k waits for the event behind ITS batch's three copies, and a plan's inputs
    are overwritten only after a host-side wait for the step that last used them."""
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
        plans = [eng.built(4, 0), eng.built(4, 1)]
        pool = [(dev.empty((4, 24)), dev.empty((4, 1, 32, 32)), dev.empty((4, 3, 32, 32))) for _ in range(3)]
        FakeDevice.pipe_log.clear()
        eng.upload_resident_async(plans[0], *pool[0])
        for k in range(5):
            eng.enqueue_train_uploaded(plans[k & 1])
            eng.upload_resident_async(plans[(k + 1) & 1], *pool[(k + 1) % 3])
    finally:
        step_mod.Ops = orig
    log = FakeDevice.pipe_log
    cps = [e for e in log if e[0] == "d2d"]
    assert len(cps) == 3 * 6
    # z of batch k: from pool[k % 3] into plans[k & 1], whole tensor
    for k in range(6):
        z = cps[3 * k]
        assert z[1] == plans[k & 1].z.ptr and z[2] == pool[k % 3][0].ptr and z[3] == 4 * 4 * 24
        assert cps[3 * k + 1][1] == plans[k & 1].x.ptr and cps[3 * k + 2][1] == plans[k & 1].y.ptr
    # every step's stage stream waits for an event recorded after that batch's three copies
    waits = [i for i, e in enumerate(log) if e[0] == "wait"]
    assert len(waits) == 5
    for k, wi in enumerate(waits):
        ri = max(i for i, e in enumerate(log[:wi]) if e[0] == "record" and e[1] == log[wi][1])
        assert sum(1 for e in log[:ri] if e[0] == "d2d") == 3 * (k + 1)
    # before the third copy set (plan 0 again) the host waited for the events recorded behind step 0 on that plan
    i3 = log.index(cps[6])
    syncs = [e for e in log[:i3] if e[0] == "host_sync"]
    recs0 = [e[1] for e in log[:i3] if e[0] == "record"]
    assert syncs and all(e[1] in recs0 for e in syncs)
    eng.close_pipeline()


def test_bench_stdout_line_is_short_strict_json():
    """the driver keeps only the tail of bench.py's stdout: the ONE JSON line must stay below 4096 bytes whatever the side
    record holds (round 5's 20.8 KB line was not parsed), be strict JSON, and keep the contract's keys + roofline + cpu_baseline"""
    import json
    import bench
    thin = [{"entry": "conv_fwd", "kernel": "fanout_kernel<18, 2, 4>", "geom": "N8 C4 512x512 K64 5x5", "ms": 0.1,
             "algorithmic_MB": 537.0, "GB/s": 3400.0, "frac_of_8TB/s": 0.42, "moved_MB": 700.0, "moved_GB/s": 4400.0}] * 21
    roof = {"bound": "mfma", "achieved": 109.8, "peak": 416.67, "kernel_dtype": "bf16x3", "unit": "TFLOP/s", "frac": 0.2635,
            "traffic": 211000000.0, "concurrent_streams": 3, "achieved_isolated": 206.4, "frac_isolated": 0.495,
            "kernel": "sp_conv2_kernel<3, 1", "launches_per_step": 14, "avg_launch_ms": 0.3867,
            "algorithmic_gflop_per_launch": 42.451, "share_of_step_time": 0.379}
    sec = {"name": "x" * 48, "metric": "m", "value": 1.0, "config": {"workload": "w" * 400}, "roofline": roof, "losses": [0.1] * 5}
    out = {"metric": "512px heightmap+texture train images/sec", "value": 280.3, "unit": "images/s", "n_gpus": 1, "steps": 20,
           "warmup": 5, "ms_per_step": 14.27, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3",
           "data": "synthetic", "config": {"workload": "w" * 600, "global_batch": 4, "in_shp": 512, "parallelism": "dp1",
                                           "hip_graph": False, "issue": "recorded", "host_calls_per_step": 1, "streams": 3},
           "step_frac_of_peak": 0.4596, "step_executed_frac_of_peak": 0.4004, "losses": [0.123456789] * 5,
           "steady_state": {"steps": 60, "after_steps": 25, "value": 285.4}, "value_lr0": 275.0, "value_single_batch": 281.0,
           "resident_batches": 8, "hbm_bound_layers": {"note": "n" * 200, "total_ms": 1.0, "launches": thin}, "roofline": roof,
           "secondary": [sec] * 9, "value_fp32_mfma": 180.2, "arithmetic_note": "a" * 500,
           "cpu_baseline": {"value": 0.0719, "unit": "images/s", "cores": 64, "kind": "port", "sample": "s" * 400,
                            "config1": {"value": 24.0, "unit": "images/s", "sample": "t" * 300}},
           "exchange": {"rccl_nranks": 8, "buckets": [{"label": "allreduce_p2p_gen_%d" % i, "MB": 32.0, "avg_ms": 0.5}
                                                      for i in range(40)], "exposed_wait_ms_per_step": {"A": 0.1, "B": 0.2}},
           "side_file": "gpurun_out/bench_secondary.json"}
    assert len(json.dumps(out)) > 12000
    txt = bench.headline_line(out)
    assert len(txt.encode()) < 4096 and "\n" not in txt
    line = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))     # no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "value_fp32_mfma", "value_lr0", "step_executed_frac_of_peak"):
        assert k in line, k
    assert "secondary" not in line and "hbm_bound_layers" not in line and "arithmetic_note" not in line
    assert line["roofline"]["frac"] == 0.2635 and line["cpu_baseline"]["cores"] == 64 and line["config"]["workload"].endswith("...")


def test_discriminator_conv_lrelu_maxpool_is_fused_when_the_library_serves_it():
    """Conv2DLayer -> LeakyRectify -> MaxPool2DLayer (dcgan.py:42-47) becomes one 'convpool' node: pooled output, a
    byte mask, and in the backward the mask pass (which also sums the bias gradient) in front of an ordinary conv
    backward -- for the D-loss pass (weight gradients) and the G-loss pass on the fake half (data gradients only)"""
    from tests.fake_device import PolicyDevice, PolicyOps
    net = dcgan.default_discriminator(64, True, nch=32, div=[2, 1, 1], nonlinearity=linear)
    dev = PolicyDevice()
    ops = PolicyOps(dev)
    store = ParamStore(dev, L.get_all_params(net))
    plan = NetPlan(dev, ops, net, 4, store)
    kinds = [n.op for n in plan.order]
    assert kinds.count('convpool') == 3 and 'maxpool' not in kinds and kinds.count('conv') == 1     # d_out stays a conv
    cp = [n for n in plan.order if n.op == 'convpool']
    assert [n.shape for n in cp] == [(4, 16, 32, 32), (4, 32, 16, 16), (4, 32, 8, 8)]
    assert [n.aux['full_shape'] for n in cp] == [(4, 16, 64, 64), (4, 32, 32, 32), (4, 32, 16, 16)]
    fwd = []
    plan.emit_forward(fwd)
    assert [e[0] for e in fwd].count('convpool_fwd') == 3 and 'maxpool_fwd' not in [e[0] for e in fwd]
    seed = dev.empty(plan.out.shape)
    bwd = []
    plan.emit_backward(bwd, seed, wgrad=True, tag="dloss")
    labels = [e[0] for e in bwd]
    # (the first block, one input channel, takes both gradients straight from the pooled operands: no mask pass at all)
    assert labels.count('maxpool_mask_bwd') == 2 and 'maxpool_bwd' not in labels
    assert labels.count('conv_wgrad') == 4 and labels.count('bias_grad') == 1       # only d_out's bias is a separate sum
    for e in bwd:
        e[1]()
    unpool = [c for c in ops.calls if c[0] == 'maxpool2_mask_bwd']
    assert all(c[1][6] is not None for c in unpool)                # the bias-gradient slice rides along
    sparse = [c for c in ops.calls if c[0] == 'conv2d_pool_wgrad_sparse']
    assert len(sparse) == 1 and sparse[0][1][2] == cp[0].aux['mask'] and sparse[0][1][6] is not None   # weights + bias
    assert not [c for c in ops.calls if c[0] == 'conv2d_pool_dgrad_sparse']       # the D-loss pass needs no input gradient
    ops.calls.clear()
    g2 = []
    gin = plan.emit_backward(g2, dev.empty((2, 1, 1, 1)), nslice=(2, 4), wgrad=False,
                             input_grads=[plan.input_nodes[0].layer], tag="gloss")
    for e in g2:
        e[1]()
    unpool = [c for c in ops.calls if c[0] == 'maxpool2_mask_bwd']
    assert len(unpool) == 2 and all(c[1][6] is None for c in unpool)
    # the fake half's masks: byte offset = 2 samples into each mask buffer
    for c, n in zip(unpool, reversed(cp)):
        assert c[1][0] == n.aux['mask'] + 2 * int(np.prod(n.shape[1:]))
    sparse = [c for c in ops.calls if c[0] == 'conv2d_pool_dgrad_sparse']
    assert len(sparse) == 1 and sparse[0][1][1] == cp[0].aux['mask'] + 2 * int(np.prod(cp[0].shape[1:]))
    assert sparse[0][1][0].N == 2 and not [c for c in ops.calls if c[0] == 'conv2d_pool_wgrad_sparse']
    assert gin[plan.input_nodes[0].layer].shape == (2, 1, 64, 64)
    # with GHM_NO_POOL_FUSE the graph keeps its max-pool nodes
    os.environ["GHM_NO_POOL_FUSE"] = "1"
    try:
        plain = NetPlan(dev, ops, net, 4, ParamStore(dev, L.get_all_params(net)))
    finally:
        os.environ.pop("GHM_NO_POOL_FUSE")
    assert [n.op for n in plain.order].count('maxpool') == 3


def test_reduced_precision_lowering_on_fake_device():
    """dtype='bf16': served convolutions call the *_lp entry points, every pack of a net is refreshed by ONE batched
    launch at the start of its forward (after the collapse of the generator's 5x5 weights), layers the low-precision
    kernels do not serve keep the fp32 entry points, and fp16 scales the loss seeds / unscales in the optimiser"""
    from tests.fake_device import PolicyDevice, PolicyOps
    dev = PolicyDevice()
    G = dcgan.default_generator(24, True, nch=64, div=[1, 2, 2, 2], initial_size=8)     # 8 -> 128
    Dn = dcgan.default_discriminator(128, True, nch=64, div=[2, 1, 1], nonlinearity=linear)
    U = p2p.g_unet(128, True, False, nf=32, act=tanh, bilinear_upsample=True)
    P = p2p.discriminator(128, True, False, nf=32, act=linear, mul_factor=[1, 2])
    spec = updates.rmsprop(learning_rate=updates.shared(1e-4))
    eng = GanStep(dev, G, Dn, U, P, 100, True, 'l1', spec, 'both', use_graph=False, two_streams=False, dtype='f16')
    assert eng.loss_scale == 32768.0
    b = eng.built(4)
    prog = b.train_compute[0] + b.train_compute[1]
    labels = [e[0] for e in prog]
    assert labels.count('lp_pack') == 4                                 # one per net
    lp_entries = [e for e in prog if len(e) > 2 and e[2] and e[2].get('dtype') == 'f16']
    fp32_convs = [e for e in prog if len(e) > 2 and e[2] and e[2].get('dtype') == 'f32']
    assert len(lp_entries) >= 20 and len(fp32_convs) >= 10              # both kinds of layers exist in these nets
    assert {'conv_fwd', 'conv_dgrad', 'conv_wgrad', 'upconv_fwd', 'upconv_wgrad', 'convpool_fwd'} <= {e[0] for e in lp_entries}
    # in the generator's forward every collapse precedes the batched pack, which precedes the first convolution
    ga = [e[0] for e in b.train_compute[0]]
    first_pack = ga.index('lp_pack')
    assert all(i < first_pack for i, l in enumerate(ga[:ga.index('upconv_fwd')]) if l == 'collapse_w')
    assert first_pack < ga.index('upconv_fwd')
    for e in prog + b.update[0] + b.update[1]:
        e[1]()
    calls = [o.calls for o in eng.ops]
    flat = [c for cs in calls for c in cs]
    # the loss scale is DEVICE state attached to the stage's context (dynamic: ghm_set_loss_scale_state); the host-side
    # factors stay 1, every gradient bucket is checked before the first update and the scale is updated after the last
    assert dev.ls_state is not None and eng.loss_scale_state()[0]['scale'] == 32768.0
    seeds = [c for c in flat if c[0] in ('lsgan_loss',) and c[1][3] is not None]
    assert seeds and all(c[1][4] == 1.0 for c in seeds)
    rms = [c for c in flat if c[0] == 'rmsprop']
    assert len(rms) == 4 and all(c[1][-1] == 1.0 for c in rms)
    names = [c[0] for c in flat if c[0] in ('grad_check', 'rmsprop', 'loss_scale_update')]
    assert names == ['grad_check'] * 4 + ['rmsprop'] * 4 + ['loss_scale_update'], names
    # the same nets in fp32: no low-precision call at all
    eng32 = GanStep(PolicyDevice(), G, Dn, U, P, 100, True, 'l1', spec, 'both', use_graph=False, two_streams=False, dtype='f32')
    p32 = eng32.built(4)
    assert 'lp_pack' not in [e[0] for lane in p32.train_compute for e in lane]


def test_recorded_issue_replays_the_whole_sequence_in_one_call():
    """use_graph='recorded': call 0 runs eagerly, call 1 records every entry of both stage programs, the exchange and
    the updates between step_record_begin / _end on ALL contexts (two stage streams, the gradient stream, communication) and replays,
    later calls are one step_run"""
    events = []

    class RecDevice(FakeDevice):
        ops_class = RecordingOps

        @staticmethod
        def step_record_begin(devs):
            events.append(('begin', len(devs)))
            return 'step'

        @staticmethod
        def step_record_end(st):
            events.append(('end', st))

        @staticmethod
        def step_run(st):
            events.append(('run', st))

    dev = RecDevice()
    G = dcgan.default_generator(24, True, nch=16, div=[2, 2, 4])
    Dn = dcgan.default_discriminator(32, True, nch=16, div=[4, 2, 2], nonlinearity=linear)
    U = p2p.g_unet(32, True, False, nf=4, act=tanh, bilinear_upsample=True)
    P = p2p.discriminator(32, True, False, nf=4, act=linear, mul_factor=[1, 2])
    spec = updates.rmsprop(learning_rate=updates.shared(1e-4))
    eng = GanStep(dev, G, Dn, U, P, 100, True, 'l1', spec, 'both', use_graph='recorded')
    assert eng.side[0] is not None and eng.side[0][0] is eng.side[1][0]      # ONE gradient stream for both stages
    assert len(eng._all_devs()) == 3
    b = eng.built(4)
    n_entries = len(eng._sequence(b, 'train'))
    assert n_entries == sum(len(l) for l in b.train_compute) + sum(len(l) for l in b.update)
    count = lambda: sum(len(o.calls) for o in eng.ops) + sum(len(sd[1].calls) for sd in eng.side)
    base = count()                               # capability queries made while the plans were built
    eng.enqueue_train(b)
    per_step = count() - base
    assert events == [] and per_step >= n_entries - 50       # fork / join entries are stream waits, not Ops calls
    eng.enqueue_train(b)
    assert events == [('begin', 3), ('end', 'step'), ('run', 'step')]
    assert count() == base + 2 * per_step        # the recording pass issued the same calls once more
    eng.enqueue_train(b)
    eng.enqueue_train(b)
    assert events[3:] == [('run', 'step'), ('run', 'step')] and count() == base + 2 * per_step


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


@pytest.mark.parametrize("leftovers", [False, True])
def test_unique_id_rendezvous_two_processes(tmp_path, leftovers):
    """leftovers: a crashed earlier job with the same key left a FRESH publication, announcement and acknowledgement
    behind (the case a modification-time test cannot tell from this launch's files): they must not be accepted"""
    path = str(tmp_path / "uid")
    if leftovers:
        stale = bytes(16)
        for name, data in ((path, b"\xee" * 128 + stale), (path + ".h1", stale), (path + ".a1", stale)):
            with open(name, "wb") as f:
                f.write(data)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_worker, args=(r, path, q)) for r in (1, 0)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    assert got[0] == got[1] == bytes(range(128))


def test_bilinear_decoder_stages_are_lowered_onto_the_coarse_grid(monkeypatch):
    """engine.py R3b without a GPU: BilinearUpsample2DLayer(2) -> Conv2DLayer(3x3, 'same') pairs of p2p.g_unet (p2p.py:204-267)
    become mode-1 ``upconv`` nodes in the split arithmetic modes where the geometry is served (channels / filters multiples of 32,
    coarse maps >= GHM_BLCONV_MIN): collapsed-weight table in mode 1, the frame launch behind the collapsed convolution, the
    BatchNorm written straight into the next ConcatLayer buffer (no concat copy), gather / frame launches in the backward pass,
    the frame weight gradients behind the batched expansion; the literal form in 'f32' and under GHM_NO_BLCONV."""
    from tests.fake_device import PolicyDevice, PolicyOps
    U = p2p.g_unet(128, True, False, nf=32, act=tanh, bilinear_upsample=True)

    def plan_of(dtype):
        dev = PolicyDevice()
        ops = PolicyOps(dev)
        return NetPlan(dev, ops, U, 4, ParamStore(dev, L.get_all_params(U)), dtype=dtype), ops

    monkeypatch.setenv("GHM_BLCONV_MIN", "16")
    plan, ops = plan_of('bf16x3')
    ups = [n for n in plan.order if n.op == 'upconv']
    assert [n.attrs.get('mode') for n in ups] == [1, 1]                     # coarse 16x16 (C 256 -> K 64) and 32x32 (C 128 -> K 32)
    assert [tuple(n.shape) for n in ups] == [(16, 64, 16, 16), (16, 32, 32, 32)]
    assert sum(n.op == 'up_bilinear' for n in plan.order) == 3              # the inner stages (8x8 .. 2x2 coarse) stay literal
    for n in ups:
        sh = n.consumers[0].consumers[0]
        assert n.consumers[0].op == 'bn' and sh.op == 'pp_to_hi' and sh.alias is not None       # lives inside the concat buffer
        assert 'fl' in n.aux and 'dyl' in n.aux
    fwd = []
    plan.emit_forward(fwd)
    labels = [e[0] for e in fwd]
    assert labels.count('blconv_fwd') == 2 and labels.count('blconv_frame_fwd') == 2 and 'concat_copy' not in labels
    for i, lab in enumerate(labels):
        if lab == 'blconv_fwd':
            assert labels[i + 1] == 'blconv_frame_fwd'                      # the frame before anything reads the convolution's output
    for e in fwd:
        e[1]()
    tab = [c for c in ops.calls if c[0] == 'collapse_table'][0][1][0]
    assert [it[6] for it in tab] == [1, 1]                                  # mode 1: the bilinear tap map
    seed = plan.dev.empty(plan.out.shape)
    bwd = []
    plan.emit_backward(bwd, seed, wgrad=True)
    lb = [e[0] for e in bwd]
    assert lb.count('blconv_frame_gather') == 2 and lb.count('blconv_frame_dgrad') == 2 and lb.count('blconv_frame_wgrad') == 2
    assert lb.index('expand_wgrad') < min(i for i, x in enumerate(lb) if x == 'blconv_frame_wgrad')
    for i, x in enumerate(lb):
        if x == 'blconv_dgrad':
            assert lb[i + 1] == 'blconv_frame_dgrad'
    monkeypatch.delenv("GHM_BLCONV_MIN")
    assert sum(n.op == 'upconv' for n in plan_of('bf16x3')[0].order) == 1   # default: coarse maps of at least 32 x 32
    assert not any(n.op == 'upconv' for n in plan_of('f32')[0].order)       # the fp32-MFMA mode keeps the literal form
    monkeypatch.setenv("GHM_NO_BLCONV", "1")
    assert not any(n.op == 'upconv' for n in plan_of('bf16x3')[0].order)
