"""Whole-step parity: Pix2Pix.train_fn / loss_fn / gen_fn / z_fn on the MI355X against the numpy oracle
(oracle.step.train_step in float64) on identical seeded parameters and inputs.

Tolerance: rel-L2 <= 1e-3 is the north_star bound; fp32 kernels are expected (and asserted) at <= 2e-4 on
gradients and <= 1e-5 on losses / outputs / post-step parameters."""
import numpy as np
import pytest

from oracle import step as ostep

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


SMALL = dict(in_shp=32, latent_dim=24,
             gen_dcgan=dict(nch=16, div=[2, 2, 4]),
             disc_dcgan=dict(nch=16, div=[4, 2, 2]),
             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))


def build_model(cfg, seed, dev=None, **kw):
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.pix2pix import Pix2Pix
    from gan_heightmaps_amd import nonlinearities as NL, updates as UP
    g, d, u, p = cfg['gen_dcgan'], cfg['disc_dcgan'], cfg['gen_p2p'], cfg['disc_p2p']
    nl = {'linear': NL.linear, 'tanh': NL.tanh, 'sigmoid': NL.sigmoid}
    opt = UP.rmsprop if cfg['opt'] == 'rmsprop' else UP.adam
    return Pix2Pix(
        gen_fn_dcgan=dcgan.default_generator, disc_fn_dcgan=dcgan.default_discriminator,
        gen_params_dcgan=dict(nch=g['nch'], h=g['h'], initial_size=g['initial_size'], div=g['div'],
                              bilinear_upsample=g['bilinear_upsample']),
        disc_params_dcgan=dict(nch=d['nch'], h=d['h'], div=d['div'], bn=d['bn'],
                               nonlinearity=nl[d['nonlinearity']], pool_mode=d['pool_mode']),
        gen_fn_p2p=p2p.g_unet, disc_fn_p2p=p2p.discriminator,
        gen_params_p2p=dict(nf=u['nf'], act=nl[u['act']], bilinear_upsample=u['bilinear_upsample']),
        disc_params_p2p=dict(nf=p['nf'], bn=p['bn'], act=nl[p['act']], mul_factor=p['mul_factor']),
        in_shp=cfg['in_shp'], latent_dim=cfg['latent_dim'],
        is_a_grayscale=cfg['is_a_grayscale'], is_b_grayscale=cfg['is_b_grayscale'],
        alpha=cfg['alpha'], lsgan=cfg['lsgan'], reconstruction=cfg['reconstruction'],
        opt=opt, opt_args={'learning_rate': UP.shared(np.float32(cfg['lr']))},
        train_mode=cfg['train_mode'], verbose=False, seed=seed, device=dev, **kw)


# rel-L2 bound of config 1's generator gradients: 2 x the float32 ORACLE's worst draw over data seeds 100..115 at parameter
# seed 7 (5.7e-3, seed 103: tests below recompute it) + 1e-4 -- the same number for the fp32 MFMA and the split-fp32 mode
CONFIG1_GEN_BOUND = 1.15e-2

NETS = [('dcgan', 'gen', 'dcgan_gen'), ('dcgan', 'disc', 'dcgan_disc'), ('p2p', 'gen', 'p2p_gen'),
        ('p2p', 'disc', 'p2p_disc')]


def model_params(model):
    from gan_heightmaps_amd import layers as L
    return {(a, b): L.get_all_param_values(getattr(model, a)[b]) for a, b, _ in NETS}


def model_grads(model):
    from gan_heightmaps_amd import layers as L
    out = {}
    for a, b, k in NETS:
        st = model.engine.stores[k]
        out[(a, b)] = [st.download_grad(p) for p in L.get_all_params(getattr(model, a)[b], trainable=True)]
    return out


@pytest.fixture(scope="module")
def dev():
    from gan_heightmaps_amd import device
    if device.device_count() == 0:
        pytest.fail("no HIP device visible")
    d = device.Device(0)
    yield d
    d.close()


@pytest.mark.parametrize("variant", ["rmsprop_bilinear", "adam_deconv", "p2p_only_l2", "dcgan_only_bce", "bn_discriminators",
                                     "config1_dcgan64_b16"])
@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_train_step_parity(dev, variant, dtype):
    """(dtype 'bf16x3': the split-fp32 mode, csrc/conv_split.hip -- the same variants, the same bounds)"""
    over = dict(SMALL)
    if variant == "adam_deconv":
        over.update(opt='adam', lr=1e-3, gen_p2p=dict(nf=4, bilinear_upsample=False),
                    gen_dcgan=dict(nch=16, div=[2, 2, 4], bilinear_upsample=True))
    elif variant == "p2p_only_l2":
        over.update(train_mode='p2p', reconstruction='l2')
    elif variant == "dcgan_only_bce":
        over.update(train_mode='dcgan', lsgan=False,
                    disc_dcgan=dict(nch=16, div=[4, 2, 2], nonlinearity='sigmoid'),
                    disc_p2p=dict(nf=4, mul_factor=[1, 2], act='sigmoid'))
    elif variant == "bn_discriminators":
        # dcgan.default_discriminator(bn=True) / p2p.discriminator(bn=True): the batched [real | fake] pass keeps
        # per-half BatchNorm statistics (two get_output calls in the reference)
        over.update(disc_dcgan=dict(nch=16, div=[4, 2, 2], bn=True), disc_p2p=dict(nf=4, mul_factor=[1, 2], bn=True))
    elif variant == "config1_dcgan64_b16":
        # BASELINE config 1: DCGAN 64x64 generator + discriminator, batch 16
        over = dict(in_shp=64, latent_dim=100, train_mode='dcgan',
                    gen_dcgan=dict(nch=64, div=[2, 2, 4, 4]), disc_dcgan=dict(nch=64, div=[8, 4, 2, 1]),
                    gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))
    cfg = ostep.default_cfg(**over)
    B, seed = (16, 7) if variant == "config1_dcgan64_b16" else (4, 7)      # seed chosen so that D's final ReLU (dcgan.py:50) is alive: seed 11 gives d == 0
    model = build_model(cfg, seed, dev, dtype=dtype)
    state = ostep.init_state(cfg, seed, np.float32)
    # identical initial parameters on both sides (independent construction paths)
    mp = model_params(model)
    for key in ostep.NET_ORDER:
        for a, b in zip(mp[key], state['params'][key[0]][key[1]]):
            assert np.array_equal(a, b)
    for it in range(3):          # step 0 eager, step 1 captured + launched, step 2 graph replay
        Z, X, Y = ostep.synthetic_batch(B, cfg, seed=100 + it)
        ref = ostep.train_step(state, Z, X, Y, dtype=np.float64)
        got = model.train_fn(Z, X, Y)
        assert len(got) == 5
        assert rel(got, ref['losses']) < 1e-5, (it, got, ref['losses'])
        mg = model_grads(model)
        for key in ref['grads']:
            flat_g = np.concatenate([g.ravel() for g in mg[key]])
            flat_r = np.concatenate([g.ravel() for g in ref['grads'][key]])
            assert np.linalg.norm(flat_r) > 1e-3, "vacuous test: reference gradient is zero"
            # config 1 at initialisation has generator channels whose batch mean is ~50x their spread: BatchNorm's
            # E[x^2] - mean^2 amplifies ANY fp32 rounding of the convolutions there, and the figure of one (seed, step) is a
            # draw -- of the float32 oracle as much as of either device path (test_config1_gradient_statistic_over_seeds
            # below holds both arithmetic modes to the SAME statistic over 16 data seeds).  Here, on three fixed batches, the
            # generator is held to that test's distribution bound (2 x the float32 oracle's worst draw + 1e-4), the
            # discriminator -- no BatchNorm chain of its own, it sees the generator's output -- to 6e-4; one number per net for both modes
            tol = (CONFIG1_GEN_BOUND if key == ('dcgan', 'gen') else 6e-4) if variant == "config1_dcgan64_b16" else 2e-4
            assert rel(flat_g, flat_r) < tol, (it, key, rel(flat_g, flat_r))
        mp = model_params(model)
        for key in ostep.NET_ORDER:
            ref_p = state['params'][key[0]][key[1]]
            tr = [i for i, t in enumerate(ostep.specs(cfg)[key].trainable) if t]
            zero_grad = {i for j, i in enumerate(tr) if key in ref['grads']
                         and np.linalg.norm(ref['grads'][key][j]) < 1e-10}
            keep = [i for i in range(len(ref_p)) if i not in zero_grad]
            assert rel(np.concatenate([mp[key][i].ravel() for i in keep]),
                       np.concatenate([np.asarray(ref_p[i], np.float64).ravel() for i in keep])) < 1e-5, (it, key)
            for i, (a, b) in enumerate(zip(mp[key], ref_p)):   # every tensor, incl. BN running mean / inv_std
                if i in zero_grad:
                    # conv bias feeding a BatchNorm: the true gradient is exactly 0, fp32 leaves ~1e-9 noise and
                    # Adam's normalisation turns noise into steps of up to lr -- bounded, not comparable
                    assert np.abs(a - b).max() <= 2 * cfg['lr'], (it, key, a.shape)
                else:
                    assert rel(a, b) < 1e-3 or np.abs(a - b).max() < 1e-6, (it, key, a.shape)
        # keep the two sides from drifting apart through fp32 rounding: resync the oracle to the device
        for key in ostep.NET_ORDER:
            state['params'][key[0]][key[1]] = [a.copy() for a in mp[key]]


CONFIG1 = dict(in_shp=64, latent_dim=100, train_mode='dcgan',
               gen_dcgan=dict(nch=64, div=[2, 2, 4, 4]), disc_dcgan=dict(nch=64, div=[8, 4, 2, 1]),
               gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))


@pytest.fixture(scope="module")
def config1_oracle_draws():
    """BASELINE config 1 at initialisation (parameter seed 7), data seeds 100..115: the float64 oracle's gradients and the
    rel-L2 distance of the SAME oracle run in float32 from them -- the spread any float32 implementation of this step has"""
    cfg = ostep.default_cfg(**CONFIG1)
    out = []
    for dseed in range(100, 116):
        Z, X, Y = ostep.synthetic_batch(16, cfg, seed=dseed)
        r64 = ostep.train_step(ostep.init_state(cfg, 7, np.float32), Z, X, Y, dtype=np.float64)
        r32 = ostep.train_step(ostep.init_state(cfg, 7, np.float32), Z, X, Y, dtype=np.float32)
        flat = {key: np.concatenate([g.ravel() for g in r64['grads'][key]]) for key in r64['grads']}
        spread = {key: rel(np.concatenate([g.ravel() for g in r32['grads'][key]]), flat[key]) for key in flat}
        out.append((dseed, (Z, X, Y), flat, spread))
    return cfg, out


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_config1_gradient_statistic_over_seeds(dev, dtype, config1_oracle_draws):
    """Config 1's generator gradients are ill-conditioned at initialisation on some batches (a channel whose batch mean is far
    above its spread: the BatchNorm chain amplifies a 1e-7 rounding 1000x) -- WHICH batches is a draw of the rounding errors
    that differs between the float32 oracle (8 of these 16 seeds), the fp32 MFMA path (5) and the split-fp32 path (6), so no
    per-seed bound is fair to any of them.  Both device modes are held to the same statistic of the 16 draws:
      * the median is the unamplified figure: <= 1e-5 (measured 6e-7 in both modes; the float32 oracle's own median is 1e-5);
      * the worst draw is within 2 x the float32 oracle's worst draw + 1e-4;
      * the discriminator (no BatchNorm chain of its own) is within 6e-4 on every seed.
    (profiles/r04_config1_gradient_lottery.txt is the table this test replaces a one-draw bound with.)"""
    cfg, draws = config1_oracle_draws
    gen, disc = ('dcgan', 'gen'), ('dcgan', 'disc')
    errs, derrs = [], []
    for dseed, (Z, X, Y), flat, spread in draws:
        model = build_model(cfg, 7, dev, dtype=dtype)
        model.train_fn(Z, X, Y)
        mg = model_grads(model)
        errs.append(rel(np.concatenate([g.ravel() for g in mg[gen]]), flat[gen]))
        derrs.append(rel(np.concatenate([g.ravel() for g in mg[disc]]), flat[disc]))
        del model
    errs, derrs = np.asarray(errs), np.asarray(derrs)
    oracle32 = np.asarray([sp[gen] for _, _, _, sp in draws])
    table = ["%d: %s %.2e | float32 oracle %.2e" % (d[0], dtype, e, o) for d, e, o in zip(draws, errs, oracle32)]
    assert np.median(errs) <= 1e-5, table
    bound = 2.0 * oracle32.max() + 1e-4
    assert abs(bound - CONFIG1_GEN_BOUND) <= 0.25 * CONFIG1_GEN_BOUND, (bound, CONFIG1_GEN_BOUND)   # the constant above is this figure
    assert errs.max() <= bound, table
    assert derrs.max() <= 6e-4, derrs



def test_loss_fn_and_generators(dev):
    cfg = ostep.default_cfg(**SMALL)
    B, seed = 3, 5
    model = build_model(cfg, seed, dev)
    state = ostep.init_state(cfg, seed, np.float32)
    Z, X, Y = ostep.synthetic_batch(B, cfg, seed=7)
    before = model_params(model)
    ref = ostep.train_step(state, Z, X, Y, update=False, want=('gz', 'ux'))
    got = model.loss_fn(Z, X, Y)
    assert rel(got, ref['losses']) < 1e-5
    after = model_params(model)
    from oracle import step as S
    sp = S.specs(cfg)
    for key in S.NET_ORDER:
        for a, b, r, kind in zip(before[key], after[key], state['params'][key[0]][key[1]], sp[key].kinds):
            if kind in ('mean', 'inv_std'):
                # running stats DID move (lasagne default_updates); g_conv1's batch mean is exactly 0
                assert rel(b, r) < 1e-5 or np.abs(b - r).max() < 1e-6
            else:
                assert np.array_equal(a, b)      # loss_fn never touches trainable parameters
    # z_fn / gen_fn (non deterministic: batch statistics) and *_det (running statistics)
    st2 = ostep.clone_state(state)
    fw = ostep.forward(st2, Z, X, Y)
    assert rel(model.z_fn(Z), fw['gz'].v) < 1e-5
    assert rel(model.gen_fn(X), fw['ux'].v) < 1e-5
    # the two calls above updated the running stats once more on the device; mirror on the oracle
    ostep._apply_bn_running(st2, fw)
    fwd = ostep.forward(st2, Z, X, Y, deterministic=True)
    assert rel(model.z_fn_det(Z), fwd['gz'].v) < 1e-5
    assert rel(model.gen_fn_det(X), fwd['ux'].v) < 1e-5


def test_checkpoint_roundtrip(dev, tmp_path):
    cfg = ostep.default_cfg(**SMALL)
    m1 = build_model(cfg, 1, dev)
    Z, X, Y = ostep.synthetic_batch(2, cfg, seed=3)
    m1.train_fn(Z, X, Y)
    path = str(tmp_path / "a.model")
    m1.save_model(path)
    m2 = build_model(cfg, 2, dev)
    m2.load_model(path, mode='p2p')
    p1, p2 = model_params(m1), model_params(m2)
    for a, b in zip(p1[('p2p', 'gen')], p2[('p2p', 'gen')]):
        assert np.array_equal(a, b)
    assert not np.array_equal(p1[('dcgan', 'gen')][0], p2[('dcgan', 'gen')][0])
    m2.load_model(path)
    assert rel(m2.loss_fn(Z, X, Y), m1.loss_fn(Z, X, Y)) < 1e-6


@pytest.mark.parametrize("mode", ["graph_comm_on_stream_a", "eager_comm_stream_overlap"])
def test_data_parallel_code_path_with_one_rank(dev, mode):
    """world=1 RCCL communicator + force_exchange: the exact N-rank program must reproduce the single-process step bit
    for bit (a sum over one rank, grad_scale 1).
    graph: the communicator lives on stream A; compute graphs, then one ncclAllReduce per net bucket + the losses,
    then the update graphs.  eager: the communicator has its own context = a communication stream; each bucket's
    all-reduce is enqueued inside the stage programs right after the bucket's last gradient kernel (event waits on
    the stage and gradient streams), the stage streams wait for the communication stream before their updates."""
    from gan_heightmaps_amd import device, dist
    cfg = ostep.default_cfg(**SMALL)
    Zs = [ostep.synthetic_batch(4, cfg, seed=40 + i) for i in range(3)]
    ref_model = build_model(cfg, 7, dev)
    ref = [ref_model.train_fn(*b) for b in Zs]
    ref_params = model_params(ref_model)
    eager = mode.startswith("eager")
    cdev = device.Device(dev.index) if eager else dev
    comm = dist.Comm(cdev, 0, 1)
    try:
        m = build_model(cfg, 7, dev, comm=comm, force_exchange=True, use_graph=not eager)
        b = m.engine.built(4)
        assert m.engine.exchange and m.engine.cdev is cdev
        labels = [e[0] for e in b.exchange]
        inside = [e[0] for lane in b.train_compute for e in lane if e[0].startswith("allreduce_")]
        if eager:
            # ... and one wait for the communication stream per stage stream (each is bracketed by bench.py: the
            # exposed part of the exchange)
            assert m.engine.side[0] is not None and len(inside) == 4 and labels == ["allreduce_losses", "wait_comm", "wait_comm"]
        else:
            assert inside == [] and labels == ["allreduce_dcgan_disc", "allreduce_dcgan_gen", "allreduce_p2p_disc",
                                               "allreduce_p2p_gen", "allreduce_losses", "wait_comm", "wait_comm"]
        got = [m.train_fn(*b_) for b_ in Zs]        # graph mode: eager, captured, replayed
        assert np.array_equal(np.asarray(got), np.asarray(ref))
        p = model_params(m)
        for key in ref_params:
            for a, b_ in zip(p[key], ref_params[key]):
                assert np.array_equal(a, b_)
        assert m.engine.replica_checksums()[0] == m.engine.replica_checksums()[1]
    finally:
        comm.close()
        if eager:
            cdev.close()


@pytest.mark.parametrize("issue", [False, "recorded"])
def test_bf16_exchange_buffer_with_one_rank(dev, issue):
    """exchange_mode='allreduce_bf16' (opt-in, reduced precision: every gradient sub-bucket rounded to bf16 for the trip, summed
    by RCCL in bf16, widened back) on a world-1 communicator: ncclAllReduce on ncclBfloat16 runs -- eager and replayed inside
    ghm_step_run --, the losses stay fp32 and bit-identical, and what reaches the optimiser is the gradient rounded to bf16
    (nearest even): every element within 2^-8 of the single-process step's, most of them changed."""
    from gan_heightmaps_amd import device, dist
    cfg = ostep.default_cfg(**SMALL)
    batches = [ostep.synthetic_batch(4, cfg, seed=60 + i) for i in range(3)]
    ref_model = build_model(cfg, 7, dev, use_graph=issue)
    cdev = device.Device(dev.index)
    comm = dist.Comm(cdev, 0, 1)
    try:
        m = build_model(cfg, 7, dev, comm=comm, force_exchange=True, use_graph=issue, exchange_mode='allreduce_bf16',
                        bucket_mb=2048.0 / 2 ** 20)
        assert m.engine.exchange_mode == 'allreduce_bf16' and not m.engine.sharded
        for it, b_ in enumerate(batches):
            from gan_heightmaps_amd import layers as L
            for a_, h_, _ in NETS:          # same parameters on both sides before every step (the updates differ by the rounding)
                L.set_all_param_values(getattr(m, a_)[h_], L.get_all_param_values(getattr(ref_model, a_)[h_]))
            want, got = ref_model.train_fn(*b_), m.train_fn(*b_)
            assert want == got                                   # the losses are reduced in fp32
            gr, gm = model_grads(ref_model), model_grads(m)
            for key in gr:
                fr, fm = (np.concatenate([g.ravel() for g in x[key]]) for x in (gr, gm))
                nz = fr != 0
                assert np.all(np.abs(fm[nz] - fr[nz]) <= np.abs(fr[nz]) * 2.0 ** -8), (it, key)
                assert np.array_equal(fm, fm.view(np.uint32).__and__(0xffff0000).view(np.float32))     # bf16 values
                assert (fm != fr).mean() > 0.5
    finally:
        comm.close()
        cdev.close()


@pytest.mark.parametrize("issue", [False, "recorded", True])
@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_sharded_update_code_path_with_one_rank(dev, issue, dtype):
    """exchange_mode='rs_ag' on a world-1 RCCL communicator: ncclReduceScatter per sub-bucket inside the stage programs, the
    (whole, at one rank) optimiser update and ncclAllGather on the communication stream -- the N-rank program, which at one
    rank must reproduce the single-process step bit for bit, eager and as a recorded step (the collectives replay inside
    ghm_step_run); tests/test_dp_product.py runs the same program with two ranks on CPU"""
    from gan_heightmaps_amd import device, dist
    cfg = ostep.default_cfg(**SMALL)
    Zs = [ostep.synthetic_batch(4, cfg, seed=40 + i) for i in range(4)]
    ref_model = build_model(cfg, 7, dev, dtype=dtype)       # ('bf16x3': the split weight packs follow the sharded update too)
    ref = [ref_model.train_fn(*b) for b in Zs]
    ref_params = model_params(ref_model)
    cdev = device.Device(dev.index)
    comm = dist.Comm(cdev, 0, 1, channels=(2, 4))
    try:
        m = build_model(cfg, 7, dev, comm=comm, force_exchange=True, use_graph=issue, exchange_mode='rs_ag', bucket_mb=2048.0 / 2 ** 20,
                        dtype=dtype)
        b = m.engine.built(4)
        assert m.engine.sharded
        inside = [e[0] for lane in b.train_compute for e in lane if e[0].startswith("reducescatter_")]
        after = [e[0] for e in b.exchange]
        waits = [e[0] for lane in b.train_compute for e in lane if e[0].startswith(("wait_gather_", "wait_losses_reduced"))]
        if issue is True:
            # captured HIP graphs (the Pix2Pix default): collectives and event waits stay OUTSIDE the graphs -- the reduce-scatters
            # behind both stage programs, one wait for the communication stream per stage stream at the end of the step
            assert not inside and not waits and after.count("wait_comm") == 2
            inside = [l for l in after if l.startswith("reducescatter_")]
        else:
            assert len(waits) == 6 and "wait_comm" not in after and "losses_reduced" in after
        assert len(inside) >= 6 and not any(e[0].startswith("rmsprop") for lane in b.update for e in lane)
        assert sum(l.startswith("allgather_") for l in after) == len(inside) == sum(l.startswith("rmsprop_shard_") for l in after)
        got = [m.train_fn(*b_) for b_ in Zs]
        assert np.array_equal(np.asarray(got), np.asarray(ref))
        p = model_params(m)
        for key in ref_params:
            for a, b_ in zip(p[key], ref_params[key]):
                assert np.array_equal(a, b_)
    finally:
        comm.close()
        cdev.close()


def test_allreduce_without_a_communicator_is_an_error(dev):
    """the C ABI has no identity shortcut: reducing on a context that has no communicator fails loudly
    (a step whose gradients silently stayed local would still scale them by 1/world)"""
    from gan_heightmaps_amd.device import Ops
    t = dev.zeros((1, 8, 1, 1))
    with pytest.raises(RuntimeError, match="communicator"):
        Ops(dev).allreduce_sum(t, 8)


def test_train_loop_with_device_iterator(dev, tmp_path):
    """Pix2Pix.train (pix2pix.py:187-275) end to end at small scale: device-side data iterator with augmentation,
    CSV log with the reference's header, checkpoint every `save_every` epochs, quick_run."""
    from gan_heightmaps_amd import data as D
    cfg = ostep.default_cfg(**SMALL)
    m = build_model(cfg, 7, dev)
    rng = np.random.RandomState(0)
    X = rng.randint(0, 256, (8, 32, 32, 1)).astype(np.uint8)
    Y = rng.randint(0, 256, (8, 32, 32, 3)).astype(np.uint8)
    imgen = D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
    it_t = D.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
    it_v = D.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
    out, models = str(tmp_path / "out"), str(tmp_path / "models")
    m.train(it_t, it_v, batch_size=4, num_epochs=2, out_dir=out, model_dir=models, save_every=2)
    lines = open(out + "/results.txt").read().strip().split("\n")
    assert lines[0].split(",") == ["epoch"] + ["train_" + k for k in m.train_keys] + ["valid_" + k for k in m.train_keys] \
        + ["lr", "time", "mode"]
    assert len(lines) == 3 and lines[2].startswith("2,") and lines[2].endswith(",both")
    vals = np.array([float(v) for v in lines[1].split(",")[1:11]])
    assert np.all(np.isfinite(vals)) and np.all(vals > 0)
    import os
    assert os.path.exists(models + "/2.model")
    m2 = build_model(cfg, 99, dev)
    m2.load_model(models + "/2.model")
    Zb, Xb, Yb = ostep.synthetic_batch(4, cfg, seed=1)
    assert rel(m2.loss_fn(Zb, Xb, Yb), m.loss_fn(Zb, Xb, Yb)) < 1e-6


@pytest.mark.parametrize("issue", ["recorded", False])
def test_pipelined_upload_is_bit_identical_to_the_sequential_loop(dev, tmp_path, issue):
    """The asynchronous input pipeline (GanStep.train_pipelined: page-locked staging, a copy stream, the batch of step i+1
    uploaded while step i runs) against the reference's strictly sequential train_fn(Z, X, Y) calls (pix2pix.py:201-212):
    the same losses for every step bit for bit, the same parameters afterwards -- including a ragged last batch -- and the
    whole Pix2Pix.train loop on host-array iterators writes the same results.txt with prefetch on and off."""
    cfg = ostep.default_cfg(**SMALL)
    batches = [ostep.synthetic_batch(n, cfg, seed=10 + i) for i, n in enumerate([4, 4, 4, 2])]
    a = build_model(cfg, 7, dev, use_graph=issue)
    seq = [a.train_fn(*bt) for bt in batches]
    b = build_model(cfg, 7, dev, use_graph=issue)
    pip = list(b.engine.train_pipelined(iter(batches)))
    assert len(pip) == len(seq)
    for x, y in zip(seq, pip):
        assert np.array_equal(np.asarray(x), np.asarray(y)), (x, y)
    pa, pb = model_params(a), model_params(b)
    assert all(np.array_equal(u, v) for k in pa for u, v in zip(pa[k], pb[k]))

    class It:                       # a host-array iterator with the reference's surface (util.py:45-62: .N, next())
        def __init__(self):
            self.N, self.i = 12, 0

        def __next__(self):
            _, X, Y = ostep.synthetic_batch(4, cfg, seed=100 + self.i)
            self.i += 1
            return X, Y
        next = __next__

    rows = []
    for prefetch in (True, False):
        m = build_model(cfg, 3, dev, use_graph=issue, prefetch=prefetch)
        m.sampler = np.random.RandomState(5).rand
        out = str(tmp_path / ("out%d" % prefetch))
        m.train(It(), It(), batch_size=4, num_epochs=2, out_dir=out, dump_images=False)
        rows.append([l.split(",")[:11] for l in open(out + "/results.txt").read().strip().split("\n")])
    assert rows[0] == rows[1]
    # the device-side iterator (uint8 rows + augmentation kernel) through the same pipeline: identical CSV rows again
    from gan_heightmaps_amd import data as D
    rng = np.random.RandomState(0)
    Xu = rng.randint(0, 256, (10, 32, 32, 1)).astype(np.uint8)          # 10 samples: the last slice of a pass is ragged
    Yu = rng.randint(0, 256, (10, 32, 32, 3)).astype(np.uint8)
    imgen = D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
    rows = []
    for prefetch in (True, False):
        m = build_model(cfg, 3, dev, use_graph=issue, prefetch=prefetch)
        m.sampler = np.random.RandomState(5).rand
        it_t, it_v = (D.Hdf5Iterator(Xu, Yu, 4, imgen, True, False, device=dev) for _ in range(2))
        out = str(tmp_path / ("dout%d" % prefetch))
        m.train(it_t, it_v, batch_size=4, num_epochs=3, out_dir=out, dump_images=False)
        rows.append([l.split(",")[:11] for l in open(out + "/results.txt").read().strip().split("\n")])
    assert rows[0] == rows[1]


def test_sampling_utilities_on_device(dev, tmp_path):
    """f3 (pix2pix.py:276-425): the in-HBM G -> U chain equals z_fn_det followed by gen_fn_det bit for bit, one
    batched interpolation pass equals the reference's batch-1 calls, and the frame files appear."""
    from gan_heightmaps_amd import util
    cfg = ostep.default_cfg(**SMALL)
    model = build_model(cfg, 5, dev)
    model.sampler = np.random.RandomState(9).rand
    Z = np.random.RandomState(1).rand(5, cfg['latent_dim']).astype(np.float32)
    a0 = model.z_fn_det(Z)
    b0 = model.gen_fn_det(a0)
    a1, b1 = model.engine.generate_chain(Z, deterministic=True)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    # oracle: deterministic forward of both generators on the same state
    state = ostep.init_state(cfg, 5, np.float32)
    X = np.zeros((5, 1, 32, 32), np.float32)
    fw = ostep.forward(state, Z, X, np.zeros((5, 3, 32, 32), np.float32), deterministic=True)
    assert rel(a1, fw['gz'].v) < 1e-5
    fw2 = ostep.forward(state, Z, np.asarray(fw['gz'].v, np.float32), np.zeros((5, 3, 32, 32), np.float32),
                        deterministic=True)
    assert rel(b1, fw2['ux'].v) < 1e-5
    # non-deterministic chain: batch statistics in both nets (and running stats move, as z_fn/gen_fn do)
    a2, b2 = model.engine.generate_chain(Z, deterministic=False)
    assert a2.shape == a1.shape and b2.shape == b1.shape and np.isfinite(b2).all()
    # batched interpolation == per-sample deterministic calls
    z1, z2 = Z[0], Z[1]
    grid = model.interpolation_grid(z1, z2, mode='row')
    for n, c in enumerate([0.0, 0.1, 0.3, 0.6, 0.9, 1.0]):
        c = np.float32(c)
        one = model.z_fn_det(((1 - c) * z1 + c * z2)[None])
        assert np.abs(grid[0, n] - util.convert_to_rgb(one[0], True)).max() < 1e-5
    model.generate_interpolation_clip(2, 5, str(tmp_path / "clip"), concat=True)
    assert len(list((tmp_path / "clip").iterdir())) == 25
    assert util.imread(str(tmp_path / "clip" / "concat_0024.png")).shape == (32, 64, 3)
    model.generate_gz(6, 3, str(tmp_path / "gz"))
    assert len(list((tmp_path / "gz").iterdir())) == 6


def test_gradient_side_streams_bitwise(dev):
    """weight / bias gradients forked onto a second stream per stage (eager mode): same bits as the plain path,
    step after step"""
    cfg = ostep.default_cfg(**SMALL)
    B = 4
    Z, X, Y = ostep.synthetic_batch(B, cfg, seed=3)
    plain = build_model(cfg, 11, dev, use_graph=False, side_streams=False)
    forked = build_model(cfg, 11, dev, use_graph=False, side_streams=True)
    assert forked.engine.side[0] is not None and plain.engine.side[0] is None
    for _ in range(3):
        a, b = plain.train_fn(Z, X, Y), forked.train_fn(Z, X, Y)
        assert a == b
    pa, pb = model_params(plain), model_params(forked)
    for k in pa:
        for u, v in zip(pa[k], pb[k]):
            assert np.array_equal(u, v)
    # the default picks the side streams exactly when the step is not captured into a graph
    assert build_model(cfg, 11, dev, use_graph=False).engine.side[0] is not None
    assert build_model(cfg, 11, dev).engine.side[0] is None


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
@pytest.mark.parametrize("lsgan,pool", [(True, 'max'), (False, 'max'), (True, 'average_inc_pad')])
def test_generator_gradient_from_the_discriminator_loss_pass(dev, monkeypatch, dtype, lsgan, pool):
    """The DCGAN discriminator returns one scalar per sample and (bn=False) couples no samples, so its backward pass on a fake
    sample is linear in ONE number: the step takes dgen_loss/dG(z) as the discriminator-loss pass's fake half times
    seed_G[n] / seed_D[n] (step.py, `per_sample_ratio`) instead of walking D a second time.  A/B against the two separate
    passes (GHM_NO_RANK_ONE=1): same losses, generator gradients equal to fp32 rounding, every other net bit-identical; a
    BatchNorm discriminator keeps the separate pass."""
    over = dict(SMALL, lsgan=lsgan)
    if not lsgan:
        over.update(disc_dcgan=dict(nch=16, div=[4, 2, 2], nonlinearity='sigmoid'),
                    disc_p2p=dict(nf=4, mul_factor=[1, 2], act='sigmoid'))
    if pool != 'max':       # (dcgan.py:48-49: the first layer is then a plain conv -> LeakyRectify, its gradient buffer holds the
        over.update(disc_dcgan=dict(nch=16, div=[4, 2, 2], pool_mode=pool))      # gradient in FRONT of the nonlinearity)
    cfg = ostep.default_cfg(**over)
    B = 4
    labels = lambda m: [e[0] for e in m.engine.built(B).train_compute[0]]
    one = build_model(cfg, 7, dev, dtype=dtype)
    monkeypatch.setenv("GHM_NO_RANK_ONE", "1")
    two = build_model(cfg, 7, dev, dtype=dtype)
    two.engine.built(B)
    monkeypatch.delenv("GHM_NO_RANK_ONE")
    assert "per_sample_ratio" in labels(one) and "per_sample_ratio" not in labels(two)
    assert len(labels(one)) < len(labels(two))
    for it in range(3):          # step 0 eager, step 1 recorded, step 2 replayed
        Z, X, Y = ostep.synthetic_batch(B, cfg, seed=300 + it)
        la, lb = one.train_fn(Z, X, Y), two.train_fn(Z, X, Y)
        ga, gb = model_grads(one), model_grads(two)
        if it == 0:
            assert la == lb                  # the losses are taken before either backward pass
        else:                                # (the two generators have taken steps that differ by rounding since)
            assert rel(la, lb) < 1e-5
        for key in ga:
            fa, fb = (np.concatenate([g.ravel() for g in x[key]]) for x in (ga, gb))
            assert np.linalg.norm(fb) > 1e-6
            if it == 0 and key != ('dcgan', 'gen'):
                assert np.array_equal(fa, fb), key
            else:
                assert rel(fa, fb) < (2e-6 if it == 0 else 1e-4), (it, key, rel(fa, fb))
    # a discriminator with BatchNorm couples its samples: the separate pass stays
    bn = build_model(ostep.default_cfg(**dict(SMALL, disc_dcgan=dict(nch=16, div=[4, 2, 2], bn=True))), 7, dev, dtype=dtype)
    assert "per_sample_ratio" not in labels(bn)


def _confident_discriminator_state(cfg, seed, Z, X, Y, target):
    """parameters (float32 lists, oracle order) in which the DCGAN discriminator is WINNING: d_out's bias is lowered until its
    final ReLU (dcgan.py:50) is dead at every position of exactly ONE fake sample (D(G(z))_n == 0: both cotangents are exactly
    zero there), then d_out is scaled so that the live fake samples sit at D(G(z)) ~ ``target`` (the ReLU head is positively
    homogeneous).  Float64 oracle forward passes only."""
    st = ostep.init_state(cfg, seed, np.float32)
    dp = st['params']['dcgan']['disc']
    W0, b0 = dp[-2].copy(), dp[-1].copy()

    def d_fake(beta, scale=1.0):
        dp[-2], dp[-1] = (W0 * scale).astype(np.float32), ((b0 - beta) * scale).astype(np.float32)
        return ostep.forward(st, Z, X, Y, np.float64)['d_fake'].v.ravel()

    def first_beta(ndead):           # smallest bias shift with >= ndead dead fake samples
        lo, hi = 0.0, 1.0
        while (d_fake(hi) == 0).sum() < ndead:
            hi *= 2.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            lo, hi = (lo, mid) if (d_fake(mid) == 0).sum() >= ndead else (mid, hi)
        return hi
    assert (d_fake(0.0) > 0).all(), "seed with a dead final ReLU at initialisation"
    beta = 0.5 * (first_beta(1) + first_beta(2))
    d = d_fake(beta)
    assert (d == 0).sum() == 1
    scale = target / d[d > 0].mean()
    d = d_fake(beta, scale)
    assert (d == 0).sum() == 1 and abs(d[d > 0].mean() / target - 1) < 1e-3
    return st, d


@pytest.mark.parametrize("target", [1e-2, 1e-4])
@pytest.mark.parametrize("dtype", ["f32", "bf16x3", "bf16x2", "bf16", "f16"])
def test_generator_gradient_shortcut_with_a_confident_discriminator(dev, monkeypatch, dtype, target):
    """DESIGN 4e in the TRAINED state (the reference's own run ends at dcgan_disc 0.0119, output/.../results.txt:1001): the
    shared pass carries the fake half at seed_D[n] = 2 D(G(z))_n / B, 1e-2 .. 1e-4 of the generator-loss seed it is scaled back
    up to.  With D(G(z)) ~ 1e-2 and ~ 1e-4 and one fake sample behind a dead final ReLU (seed_D[n] == 0 exactly), in every
    arithmetic mode: generator gradients against the two separate passes (GHM_NO_RANK_ONE=1) AND against the float64 oracle --
    the shortcut may not be further from the oracle than 2 x the separate passes + 1e-6.  'f16' keeps the separate passes by
    construction (fp16 gradient operands underflow at the smaller seed: step.py _per_sample_scalar_head)."""
    from gan_heightmaps_amd import layers as L
    cfg = ostep.default_cfg(**SMALL)
    B, seed = 4, 7
    Z, X, Y = ostep.synthetic_batch(B, cfg, seed=310)
    st, d = _confident_discriminator_state(cfg, seed, Z, X, Y, target)
    ref = ostep.train_step(ostep.clone_state(st), Z, X, Y, dtype=np.float64)
    gref = np.concatenate([g.ravel() for g in ref['grads'][('dcgan', 'gen')]])
    assert np.linalg.norm(gref) > 1e-8, "vacuous: the generator gradient vanished"
    labels = lambda m: [e[0] for e in m.engine.built(B).train_compute[0]]

    def run(no_rank_one):
        if no_rank_one:
            monkeypatch.setenv("GHM_NO_RANK_ONE", "1")
        m = build_model(cfg, seed, dev, dtype=dtype)
        m.engine.built(B)
        if no_rank_one:
            monkeypatch.delenv("GHM_NO_RANK_ONE")
        L.set_all_param_values(m.dcgan['disc'], st['params']['dcgan']['disc'])
        losses = m.train_fn(Z, X, Y)
        g = model_grads(m)
        return m, losses, {k: np.concatenate([x.ravel() for x in v]) for k, v in g.items()}
    one, l1, g1 = run(False)
    two, l2, g2 = run(True)
    assert "per_sample_ratio" not in labels(two)
    assert ("per_sample_ratio" in labels(one)) == (dtype != 'f16')
    assert l1 == l2
    for key in g1:
        if key != ('dcgan', 'gen'):
            assert np.array_equal(g1[key], g2[key]), key
    scale = one.engine.loss_scale if dtype == 'f16' else 1.0
    e1, e2 = rel(g1[('dcgan', 'gen')] / scale, gref), rel(g2[('dcgan', 'gen')] / scale, gref)
    assert np.isfinite(g1[('dcgan', 'gen')]).all()
    # the shortcut against the oracle: never more than twice as far as the separate passes (+ fp32 rounding of the ratio)
    assert e1 <= 2 * e2 + 1e-6, (dtype, target, e1, e2)
    bound = {"f32": 2e-4, "bf16x3": 2e-4, "bf16x2": 2e-3, "bf16": 0.5, "f16": 0.2}[dtype]
    assert e2 < bound and e1 < bound, (dtype, target, e1, e2)
    if dtype in ("f32", "bf16x3"):
        assert rel(g1[('dcgan', 'gen')], g2[('dcgan', 'gen')]) < 5e-6


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16x2"])
def test_bilinear_decoder_on_the_coarse_grid(dev, monkeypatch, dtype):
    """p2p.g_unet(bilinear_upsample=True): every decoder stage is BilinearUpsample2DLayer(2) -> 3x3 'same' conv (p2p.py:204-267).
    In the split modes the engine evaluates the pair on the COARSE grid (engine.py R3b, csrc/conv_bilinear.hip: collapsed 3x3
    with 4K filters + the border frame) wherever the library serves the geometry; the up-sampled tensor is never written.
    A/B against the literal form (GHM_NO_BLCONV=1) and the float64 oracle over three steps (eager, recorded, replayed):
    same losses, gradients no further from the oracle than the literal form's (+ fp32 rounding), parameters after the steps."""
    cfg = ostep.default_cfg(in_shp=128, latent_dim=16, train_mode='p2p', gen_dcgan=dict(nch=16, div=[1, 1, 2, 2, 2]),
                            disc_dcgan=dict(nch=16, div=[2, 2, 2]), gen_p2p=dict(nf=32), disc_p2p=dict(nf=8, mul_factor=[1, 2]))
    B, seed = 4, 7
    monkeypatch.setenv("GHM_BLCONV_MIN", "16")      # (default 32: this 128-pixel net then has one such stage, here two)
    a = build_model(cfg, seed, dev, dtype=dtype)
    a.engine.built(B)
    monkeypatch.delenv("GHM_BLCONV_MIN")
    monkeypatch.setenv("GHM_NO_BLCONV", "1")
    lit = build_model(cfg, seed, dev, dtype=dtype)
    lit.engine.built(B)
    monkeypatch.delenv("GHM_NO_BLCONV")
    la = [e[0] for e in a.engine.built(B).train_compute[1]]
    ll = [e[0] for e in lit.engine.built(B).train_compute[1]]
    # decoder levels with coarse maps of 32 and 16 pixels take the coarse-grid form; the inner ones (8 .. 2) stay literal
    assert la.count("blconv_fwd") == 2 and la.count("blconv_frame_fwd") == 2 and la.count("blconv_frame_dgrad") == 2
    assert la.count("blconv_frame_wgrad") == 2 and "blconv_fwd" not in ll
    assert la.count("up_bilinear_fwd") == ll.count("up_bilinear_fwd") - 2
    state = ostep.init_state(cfg, seed, np.float32)
    tol = 2e-4 if dtype == "bf16x3" else 2e-3
    for it in range(3):
        Z, X, Y = ostep.synthetic_batch(B, cfg, seed=500 + it)
        ref = ostep.train_step(state, Z, X, Y, dtype=np.float64)
        ga_l, gl_l = a.train_fn(Z, X, Y), lit.train_fn(Z, X, Y)
        assert rel(ga_l, ref['losses']) < (1e-5 if dtype == "bf16x3" else 1e-4), (it, ga_l, ref['losses'])
        assert rel(ga_l, gl_l) < (1e-5 if dtype == "bf16x3" else 1e-4)
        ga, gl = model_grads(a), model_grads(lit)
        for key in ref['grads']:
            fr = np.concatenate([g.ravel() for g in ref['grads'][key]])
            fa, fl = (np.concatenate([g.ravel() for g in x[key]]) for x in (ga, gl))
            ea, el = rel(fa, fr), rel(fl, fr)
            assert ea < max(tol, 2 * el + 1e-6), (it, key, ea, el)
        # both device models and the oracle continue from the coarse-grid model's parameters
        mp = model_params(a)
        from gan_heightmaps_amd import layers as L
        for key in ostep.NET_ORDER:
            state['params'][key[0]][key[1]] = [v.copy() for v in mp[key]]
            L.set_all_param_values(getattr(lit, key[0])[key[1]], mp[key])
    # the forward-only entry points (pix2pix.py:144-147) through the same lowering: batch statistics and running statistics
    X = ostep.synthetic_batch(B, cfg, seed=510)[1]
    for fn in ("gen_fn", "gen_fn_det"):
        ya, yl = getattr(a, fn)(X), getattr(lit, fn)(X)
        assert ya.shape == (B, 3, 128, 128) and rel(ya, yl) < (1e-5 if dtype == "bf16x3" else 1e-3), (fn, rel(ya, yl))


def test_patchgan_chain_without_fp32_activations(dev, monkeypatch):
    """Split modes: in the PatchGAN's conv -> LeakyRectify -> conv chain (p2p.py:278-292) every reader of an activation takes its q
    copy, and the consumer's data gradient takes the LeakyRectify slope from the SIGN of that copy's first piece
    (ghm_conv2d_dgrad_dact_split_q) -- the fp32 activations are never written.  Same bits as the step that writes and reads
    them (GHM_DACT_FP32=1): bf16(x) has the sign of x."""
    # (256^2: the first-layer kernel with a q epilogue wants 128 output columns, the stride-2 data gradient 32 columns of dy)
    cfg = ostep.default_cfg(in_shp=256, latent_dim=16, train_mode='p2p', gen_dcgan=dict(nch=16, div=[1, 1, 2, 2, 2, 2]),
                            disc_dcgan=dict(nch=16, div=[2, 2, 2, 2]), gen_p2p=dict(nf=8),
                            disc_p2p=dict(nf=64, mul_factor=[1, 2, 4]))
    B = 4
    q = build_model(cfg, 7, dev, dtype='bf16x3')
    plan = q.engine.built(B).P
    assert any(plan._act_fp32_dropped(n) for n in plan.order), "no layer of this PatchGAN takes the q-only form: vacuous test"
    monkeypatch.setenv("GHM_DACT_FP32", "1")
    f = build_model(cfg, 7, dev, dtype='bf16x3')
    planf = f.engine.built(B).P
    assert not any(planf._act_fp32_dropped(n) for n in planf.order)
    monkeypatch.delenv("GHM_DACT_FP32")
    for it in range(3):          # eager, recorded, replayed
        Z, X, Y = ostep.synthetic_batch(B, cfg, seed=400 + it)
        assert q.train_fn(Z, X, Y) == f.train_fn(Z, X, Y)
        gq, gf = model_grads(q), model_grads(f)
        for key in gq:
            for a, b in zip(gq[key], gf[key]):
                assert np.array_equal(a, b), (it, key)


@pytest.mark.parametrize("variant", ["plain", "adam_bn_disc", "one_rank_exchange", "one_rank_exchange_subbuckets"])
def test_recorded_step_is_the_eager_schedule_in_one_call(dev, variant):
    """use_graph='recorded' (ghm_step_record_begin / ghm_step_run): the eager four-stream launch sequence -- gradient
    side streams, stream waits, (world-1) RCCL all-reduces on the communication stream, updates -- recorded once in the
    library and replayed by ONE C call per step: bit-identical to eager, step after step, and loss_fn too; the learning
    rate stays a device scalar (set_value between replays is seen); dropout counters advance on replay."""
    from gan_heightmaps_amd import device, dist
    over = dict(SMALL)
    kw = {}
    if variant == "adam_bn_disc":
        over.update(opt='adam', lr=1e-3, disc_dcgan=dict(nch=16, div=[4, 2, 2], bn=True),
                    disc_p2p=dict(nf=4, mul_factor=[1, 2], bn=True))
    cfg = ostep.default_cfg(**over)
    cdev = comm = None
    if variant.startswith("one_rank_exchange"):
        cdev = device.Device(dev.index)
        comm = dist.Comm(cdev, 0, 1)
        kw = dict(comm=comm, force_exchange=True)
        if variant.endswith("subbuckets"):          # 2 KB sub-buckets: ~10 RCCL calls per net inside the stage programs
            kw["bucket_mb"] = 2048.0 / 2 ** 20
    try:
        eager = build_model(cfg, 11, dev, use_graph=False, **{k: v for k, v in kw.items() if k != "bucket_mb"})
        rec = build_model(cfg, 11, dev, use_graph='recorded', **kw)
        if "bucket_mb" in kw:
            assert len(rec.engine.built(4).xchg_order) > 8
        assert rec.engine.side[0] is not None                    # the recorded form keeps the side streams
        for it in range(5):
            Z, X, Y = ostep.synthetic_batch(4, cfg, seed=30 + it)
            assert eager.train_fn(Z, X, Y) == rec.train_fn(Z, X, Y), it
            if it == 2:
                assert eager.loss_fn(Z, X, Y) == rec.loss_fn(Z, X, Y)
                eager.lr.set_value(np.float32(3e-4))
                rec.lr.set_value(np.float32(3e-4))
            if it == 3:
                assert eager.loss_fn(Z, X, Y) == rec.loss_fn(Z, X, Y)      # the recorded loss program, replayed
        b = rec.engine.built(4)
        assert 'train' in b.steps and 'loss' in b.steps and b.calls['train'] == 5
        pa, pb = model_params(eager), model_params(rec)
        for k in pa:
            for u, v in zip(pa[k], pb[k]):
                assert np.array_equal(u, v)
    finally:
        if comm is not None:
            comm.close()
            cdev.close()


def test_recorded_step_survives_workspace_growth(dev):
    """A recorded step replays launches that carry the context's library workspace (split-K partials, reduction
    partials) BY VALUE.  A later eager call on the same context that needs more workspace must not free that block:
    it is retired (kept until the context dies, ghm_scratch_info) and replays stay bit-identical to eager issue."""
    from gan_heightmaps_amd import device
    cfg = ostep.default_cfg(**SMALL)
    eager = build_model(cfg, 11, dev, use_graph=False)
    rec = build_model(cfg, 11, dev, use_graph='recorded')
    for it in range(3):                             # call 0 eager, call 1 records + replays, call 2 replays
        Z, X, Y = ostep.synthetic_batch(4, cfg, seed=60 + it)
        assert eager.train_fn(Z, X, Y) == rec.train_fn(Z, X, Y), it
    devs = rec.engine._all_devs()
    before = [d.scratch_info() for d in devs]
    assert all(pinned >= 1 for _, _, pinned in before)               # the recorded step pins every context it spans
    assert any(size > 0 for size, _, _ in before)                    # ... and some of its kernels do use the workspace
    # forward-only calls at a larger batch and a deep split-K convolution issued eagerly on the SAME contexts
    for m in (rec, eager):          # (both: z_fn / gen_fn update the BatchNorm running statistics, pix2pix.py:144-147)
        m.z_fn(np.random.RandomState(0).rand(32, cfg['latent_dim']).astype(np.float32))
        m.gen_fn(np.random.RandomState(1).rand(32, 1, cfg['in_shp'], cfg['in_shp']).astype(np.float32))
    d = device.conv_desc(1, 2048, 16, 16, 512, 3, 3, 1, 1)
    for dv in devs:
        ops = device.Ops(dv)
        x, w = dv.zeros((1, 2048, 16, 16)), dv.zeros((1, 2048 * 9 * 512, 1, 1))
        y, ws = dv.empty((1, 512, 16, 16)), dv.alloc(ops.wgrad_workspace(d))
        ops.conv2d_fwd(d, x, w, None, y)
        ops.conv2d_wgrad(d, x, y, w, ws)
        dv.sync()
        for t in (x, w, y):
            dv.free(t.ptr)
        dv.free(ws)
    after = [d.scratch_info() for d in devs]
    grown = [a[0] > b_[0] for a, b_ in zip(after, before)]
    assert any(grown), (before, after)                               # otherwise this test checks nothing
    for (size, retired, pinned), (size0, retired0, _), g in zip(after, before, grown):
        assert (retired - retired0 >= size0) if g else (retired == retired0)      # every outgrown block is kept, not freed
    for it in range(3, 6):
        Z, X, Y = ostep.synthetic_batch(4, cfg, seed=60 + it)
        assert eager.train_fn(Z, X, Y) == rec.train_fn(Z, X, Y), it
    pa, pb = model_params(eager), model_params(rec)
    for k in pa:
        for u, v in zip(pa[k], pb[k]):
            assert np.array_equal(u, v)


def test_dropout_generators_in_the_full_step(dev):
    """g_unet(dropout=True) + default_generator(dropout_p) inside Pix2Pix: train_fn / loss_fn run, the
    non-deterministic generator functions draw a fresh mask per call, the deterministic ones are repeatable
    (DropoutLayer is the identity there)"""
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.pix2pix import Pix2Pix
    from gan_heightmaps_amd import nonlinearities as NL, updates as UP
    model = Pix2Pix(
        gen_fn_dcgan=dcgan.default_generator, disc_fn_dcgan=dcgan.default_discriminator,
        gen_params_dcgan=dict(nch=16, div=[2, 2, 4], final_size=32, dropout_p=0.25),
        disc_params_dcgan=dict(nch=16, div=[4, 2, 2], nonlinearity=NL.linear),
        gen_fn_p2p=p2p.g_unet, disc_fn_p2p=p2p.discriminator,
        gen_params_p2p=dict(nf=4, act=NL.tanh, dropout=True, bilinear_upsample=True),
        disc_params_p2p=dict(nf=4, act=NL.linear, mul_factor=[1, 2]),
        in_shp=32, latent_dim=24, is_a_grayscale=True, is_b_grayscale=False, lsgan=True,
        opt=UP.rmsprop, opt_args={'learning_rate': UP.shared(np.float32(1e-4))}, train_mode='both', verbose=False,
        seed=3, device=dev)
    cfg = ostep.default_cfg(**SMALL)
    Z, X, Y = ostep.synthetic_batch(4, cfg, seed=1)
    for _ in range(3):                           # eager, captured, replayed: the counter lives in HBM
        losses = model.train_fn(Z, X, Y)
        assert np.isfinite(losses).all()
    assert np.isfinite(model.loss_fn(Z, X, Y)).all()
    a, b = model.gen_fn(X), model.gen_fn(X)
    assert not np.array_equal(a, b)
    assert np.array_equal(model.gen_fn_det(X), model.gen_fn_det(X))
    za, zb = model.z_fn(Z), model.z_fn(Z)
    assert not np.array_equal(za, zb) and np.array_equal(model.z_fn_det(Z), model.z_fn_det(Z))
    # graph replay keeps drawing fresh masks: two replayed steps on identical inputs give different recon losses
    l1, l2 = model.loss_fn(Z, X, Y), model.loss_fn(Z, X, Y)
    assert l1[3] != l2[3]
