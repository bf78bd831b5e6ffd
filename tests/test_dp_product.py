"""The PRODUCT's data-parallel exchange program, driven by two processes on CPU.

``GanStep`` is built on tests/fake_device.host_device_class(): host-memory contexts whose kernels are recorded
no-ops except the ones the exchange is made of -- the all-reduce (gloo here, RCCL on the GPU), the optimiser update
and memset.  Everything else is the product: stream layout, where the bucket all-reduces sit in the two stage
programs, the waits between the compute / gradient / communication streams, the 1/world scale, the parameter
broadcast, rank-0-only file output.  Asserted:
  * both ranks issue the identical collective sequence (names, buffers, sizes, order),
  * a bucket is reduced only after the communication stream has waited for the streams that write it, and no
    kernel writes the bucket between its all-reduce and its update,
  * the update of a net waits for the communication stream,
  * replicas that start from DIFFERENT weights (the reference never seeds its RNG) are identical after
    broadcast_parameters, and stay bit-identical after the step,
  * for a BatchNorm-free quantity (PatchGAN D-loss on real pairs) the averaged bucket equals the full-batch gradient.
"""
import os
import pickle
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as tdist  # noqa: E402
import torch.multiprocessing as tmp_  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_disc']
GRAD_WRITERS = ('conv2d_wgrad', 'channel_sum', 'bn_backward_x', 'upconv_expand_batched')


def _nets(seed):
    from gan_heightmaps_amd import init
    from gan_heightmaps_amd.architectures import dcgan, p2p
    from gan_heightmaps_amd.nonlinearities import linear, tanh
    init.set_rng(np.random.RandomState(seed))
    G = dcgan.default_generator(24, True, nch=16, div=[2, 2, 4])
    D = dcgan.default_discriminator(32, True, nch=16, div=[4, 2, 2], nonlinearity=linear)
    U = p2p.g_unet(32, True, False, nf=4, act=tanh, bilinear_upsample=True)
    P = p2p.discriminator(32, True, False, nf=4, act=linear, mul_factor=[1, 2])
    return G, D, U, P


def _patchgan_real_grads(values, A, B):
    """oracle gradient of mean((P(A, B) - 1)^2) w.r.t. the PatchGAN parameters (lasagne layout), float64"""
    from oracle import nets, ops
    from oracle import tape as T
    P = [T.leaf(np.asarray(v, np.float64)) for v in values]
    out, _ = nets.patchgan_fwd(P, T.leaf(A), T.leaf(B), act='linear', mul_factor=(1, 2))
    T.backward(T.scalar_loss(out, lambda v: ops.squared_error_mean(v, 1.0)))
    return [p.g for p in P]


def _pairs(n):
    rng = np.random.RandomState(0)
    return rng.rand(n, 1, 32, 32), rng.randn(n, 3, 32, 32)


def _worker(rank, world, port, out_dir, bucket_mb, exchange_mode='allreduce'):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from gan_heightmaps_amd import dist, layers as L, updates
    from gan_heightmaps_amd.engine import ParamStore
    from gan_heightmaps_amd.step import GanStep
    from tests.fake_device import host_device_class

    HostDevice = host_device_class()
    dev, cdev = HostDevice(0), HostDevice(0)          # compute stream A, communication stream

    class GlooComm:
        def __init__(self):
            self.dev, self.rank, self.world = cdev, rank, world

        def max_scalar(self, v):
            t = torch.tensor([float(v)], dtype=torch.float64)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            return float(t[0])

    G, D, U, P = _nets(seed=7 if rank == 0 else 1000 + rank)       # unseeded replicas differ
    spec = updates.rmsprop(learning_rate=updates.shared(1e-2))
    eng = GanStep(dev, G, D, U, P, 100, True, 'l1', spec, 'both', comm=GlooComm(), use_graph=False,
                  two_streams=True, side_streams=True, bucket_mb=bucket_mb, exchange_mode=exchange_mode)
    before = eng.replica_checksums()
    eng.broadcast_parameters()
    after = eng.replica_checksums()
    b = eng.built(4)
    # this rank's gradient buckets: random for three nets, the oracle's PatchGAN-on-real-pairs gradient of ITS shard
    # for the fourth (the compute kernels are no-ops here, so the buckets keep what is put in them)
    rng = np.random.RandomState(50 + rank)
    sent = {}
    for k in KEYS:
        st = eng.stores[k]
        v = rng.randn(st.n_train).astype(np.float32)
        st.g.set(np.concatenate([v, np.zeros(st.n_pad - st.n_train, np.float32)]))     # (n_pad > n_train in the sharded form only)
        sent[k] = v
    A, B = dist.shard_batch(list(_pairs(4 * world)), rank, world)
    pparams = L.get_all_params(P["out"], trainable=True)
    pvalues = [p.get_value() for p in pparams]
    pgrads = _patchgan_real_grads(pvalues, A, B)
    stp = eng.stores['p2p_disc']
    for p, g in zip(pparams, pgrads):
        stp.grad(p).set(ParamStore._to_device_layout(p, g))
    sent['p2p_disc'] = stp.g.numpy().ravel()[:stp.n_train].copy()
    w0 = {k: eng.stores[k].w.numpy().ravel()[:eng.stores[k].n_train].copy() for k in KEYS}
    log0 = len(HostDevice.shared.log)
    eng.enqueue_train(b)
    eng.sync()
    out = {
        "rank": rank, "crc_before": before, "crc_after": after, "crc_end": eng.replica_checksums(),
        "log": HostDevice.shared.log[log0:], "sent": sent, "w0": w0,
        "g": {k: eng.stores[k].g.numpy().ravel()[:eng.stores[k].n_train].copy() for k in KEYS},
        "w": {k: eng.stores[k].w.numpy().ravel()[:eng.stores[k].n_train].copy() for k in KEYS},
        "g_range": {k: (eng.stores[k].g.ptr, eng.stores[k].n_train) for k in KEYS},
        "names": {"A": eng.devs[0].name, "B": eng.devs[1].name, "sideA": eng.side[0][0].name,
                  "sideB": eng.side[1][0].name, "comm": cdev.name},
        "pgrad_avg": [stp.download_grad(p).astype(np.float64) / world for p in pparams],
        "pvalues": pvalues, "xchg_order": list(b.xchg_order),
        "w_range": {k: (eng.stores[k].w.ptr, eng.stores[k].n_pad) for k in KEYS},
        "acc": {k: eng.stores[k].opt_state['acc'].numpy().ravel()[:eng.stores[k].n_train].copy() for k in KEYS},
    }
    with open(os.path.join(out_dir, "r%d.pkl" % rank), "wb") as f:
        pickle.dump(out, f)
    tdist.destroy_process_group()


# "whole": every net's bucket is one collective (the nets of this test are far below the 32 MB default);
# "sub": 2 KB sub-buckets -- the shape the full-size U-Net's 140 MB bucket takes at the default size
@pytest.fixture(scope="module", params=["whole", "sub"])
def ranks(request, tmp_path_factory):
    d = tmp_path_factory.mktemp("dp_product_" + request.param)
    world, port = 2, 33500 + (os.getpid() % 2000) + (0 if request.param == "whole" else 1)
    bucket_mb = None if request.param == "whole" else 2048.0 / 2 ** 20
    tmp_.spawn(_worker, args=(world, port, str(d), bucket_mb), nprocs=world, join=True)
    out = [pickle.load(open(os.path.join(str(d), "r%d.pkl" % r), "rb")) for r in range(world)]
    for r in out:
        r["mode"] = request.param
    return out


def _net_of(r, ptr):
    for k in KEYS:
        g0, n = r["g_range"][k]
        if g0 <= ptr < g0 + 4 * n:
            return k
    return "losses"


def test_replicas_agree_after_broadcast_and_after_the_step(ranks):
    r0, r1 = ranks
    assert r0["crc_before"][0] != r0["crc_before"][1]            # (min, max) over ranks: the inits really differed
    for r in ranks:
        assert r["crc_after"][0] == r["crc_after"][1]
        assert r["crc_end"][0] == r["crc_end"][1]
    for k in KEYS:
        assert np.array_equal(r0["w0"][k], r1["w0"][k])           # rank 1 now holds rank 0's weights
        assert np.array_equal(r0["w"][k], r1["w"][k])             # and the replicas stay bit-identical
        assert not np.array_equal(r0["w"][k], r0["w0"][k])


def test_both_ranks_issue_the_same_collective_sequence(ranks):
    seqs = [[e for e in r["log"] if e[1] == "allreduce_sum"] for r in ranks]
    assert seqs[0] == seqs[1]
    r0 = ranks[0]
    order = [_net_of(r0, e[2]) for e in seqs[0]]
    # discriminator buckets go out before their generators' (they are ready first), the losses last
    assert order[-1] == "losses" and "losses" not in order[:-1] and set(order[:-1]) == set(KEYS)
    last = {k: max(i for i, o in enumerate(order) if o == k) for k in KEYS}
    first = {k: min(i for i, o in enumerate(order) if o == k) for k in KEYS}
    assert last['dcgan_disc'] < first['dcgan_gen'] and last['p2p_disc'] < first['p2p_gen']
    assert all(e[0] == r0["names"]["comm"] for e in seqs[0])      # all on the communication stream
    # the program's own record of its sub-buckets (per net: the host interleaves the two stage programs)
    for k in KEYS:
        assert [(e[2], e[3]) for e in seqs[0] if _net_of(r0, e[2]) == k] == \
            [(r0["g_range"][k][0] + 4 * lo, n) for _, kk, lo, n in r0["xchg_order"] if kk == k]
    for k in KEYS:
        g0, n = r0["g_range"][k]
        rng_ = sorted((e[2], e[3]) for e in seqs[0] if _net_of(r0, e[2]) == k)
        if r0["mode"] == "whole":
            assert rng_ == [(g0, n)]                               # whole bucket, one collective
        else:
            # sub-buckets: disjoint, contiguous, covering the bucket exactly; each at least the requested size except
            # the one holding the net's FIRST parameters (whatever is left); sent highest offsets first -- the order
            # the backward pass completes them in a chain-structured net
            assert rng_[0][0] == g0 and all(a[0] + 4 * a[1] == b_[0] for a, b_ in zip(rng_, rng_[1:]))
            assert rng_[-1][0] + 4 * rng_[-1][1] == g0 + 4 * n
            assert all(4 * m >= 2048 for p_, m in rng_[1:])
            if k in ('dcgan_disc', 'p2p_disc'):
                assert len(rng_) >= 2
                sent = [e[2] for e in seqs[0] if _net_of(r0, e[2]) == k]
                assert sent == sorted(sent, reverse=True)


def test_buckets_are_reduced_after_their_writers_and_updated_after_the_reduce(ranks):
    for r in ranks:
        log, nm = r["log"], r["names"]
        reduces = [(i, e) for i, e in enumerate(log) if e[1] == "allreduce_sum" and _net_of(r, e[2]) != "losses"]
        assert len(reduces) >= 4
        some_writers = {k: False for k in KEYS}
        for i_red, e_red in reduces:
            k = _net_of(r, e_red[2])
            g0, n = e_red[2], e_red[3]                  # this (sub-)bucket
            lane, side = (nm["A"], nm["sideA"]) if k.startswith("dcgan") else (nm["B"], nm["sideB"])
            writers = [i for i, e in enumerate(log) if e[1] in GRAD_WRITERS
                       and any(g0 <= p < g0 + 4 * n for p in e[2:] if isinstance(p, int))]
            if not writers:                             # a sub-bucket of BatchNorm-fed biases only: never written (zero)
                assert r["mode"] == "sub"
                continue
            some_writers[k] = True
            assert max(writers) < i_red, k
            # the communication stream waited for the stage stream and its gradient stream after the last writer
            waits = [i for i, e in enumerate(log[:i_red]) if e[0] == nm["comm"] and e[1] == "wait_for"]
            for src in (lane, side):
                assert any(i > max(writers) and log[i][2] == src for i in waits), (k, src)
            i_upd = next(i for i, e in enumerate(log) if e[1] == "rmsprop")
            i_upd_k = next(i for i, e in enumerate(log) if e[1] == "rmsprop" and e[0] == lane)
            assert i_red < i_upd
            # the stage stream waits for the communication stream (after the LAST collective) before its updates
            last_coll = max(i for i, e in enumerate(log) if e[1] == "allreduce_sum")
            assert any(last_coll < i < i_upd_k and e == (lane, "wait_for", nm["comm"]) for i, e in enumerate(log))
        assert all(some_writers.values())


def test_reduced_buckets_are_the_sum_and_the_update_uses_the_mean(ranks):
    r0, r1 = ranks
    for k in KEYS:
        total = r0["sent"][k] + r1["sent"][k]
        assert np.array_equal(r0["g"][k], total) and np.array_equal(r1["g"][k], total)
        gs = total * np.float32(0.5)
        acc = np.float32(0.1) * gs * gs                                     # rho 0.9, zero accumulator
        want = r0["w0"][k] - np.float32(1e-2) * gs / np.sqrt(acc + np.float32(1e-6))
        assert np.allclose(r0["w"][k], want, rtol=1e-6, atol=1e-7)


def test_averaged_bucket_equals_the_full_batch_gradient_without_batchnorm(ranks):
    A, B = _pairs(8)
    full = _patchgan_real_grads(ranks[0]["pvalues"], A, B)
    for got, want in zip(ranks[0]["pgrad_avg"], full):
        assert got.shape == want.shape
        assert np.linalg.norm(got - want) <= 2e-6 * np.linalg.norm(want) + 1e-12      # float32 buckets


# ---- the sharded-update form of the same exchange (exchange_mode='rs_ag'): reduce-scatter per sub-bucket, this rank's
# ---- 1 / world slice of the optimiser update on the communication stream, all-gather of the updated parameters -------
@pytest.fixture(scope="module")
def sharded(tmp_path_factory):
    out = {}
    for i, mode in enumerate(("allreduce", "rs_ag")):
        d = tmp_path_factory.mktemp("dp_sharded_" + mode)
        world, port = 2, 35500 + (os.getpid() % 2000) + i
        tmp_.spawn(_worker, args=(world, port, str(d), 2048.0 / 2 ** 20, mode), nprocs=world, join=True)
        out[mode] = [pickle.load(open(os.path.join(str(d), "r%d.pkl" % r), "rb")) for r in range(world)]
    return out


def test_sharded_update_equals_the_allreduce_form_and_keeps_the_replicas_identical(sharded):
    ar, sh = sharded["allreduce"], sharded["rs_ag"]
    for r in sh:
        assert r["crc_end"][0] == r["crc_end"][1]                      # replicas identical after the step
    for k in KEYS:
        assert np.array_equal(sh[0]["w"][k], sh[1]["w"][k])
        assert np.array_equal(sh[0]["w0"][k], ar[0]["w0"][k])          # (same seeds: the two runs start from the same weights)
        # every element was updated exactly once, by the rank that owns its shard, with the same arithmetic
        assert np.array_equal(sh[0]["w"][k], ar[0]["w"][k]), k
        assert not np.array_equal(sh[0]["w"][k], sh[0]["w0"][k])
        # the optimiser state is SHARDED: a rank's accumulator is the all-reduce form's on its own shards, untouched (zero)
        # on the other rank's; together the two ranks hold it exactly once
        a0, a1, full = sh[0]["acc"][k], sh[1]["acc"][k], ar[0]["acc"][k]
        assert np.array_equal(np.where(a0 != 0, a0, a1), full)
        assert not np.any((a0 != 0) & (a1 != 0))


def test_sharded_collective_sequence_and_stream_discipline(sharded):
    sh = sharded["rs_ag"]
    seqs = [[e for e in r["log"] if e[1] in ("reduce_scatter_sum", "all_gather", "allreduce_sum")] for r in sh]
    assert seqs[0] == seqs[1]                                          # identical on both ranks
    r = sh[0]
    nm = r["names"]
    assert all(e[0] == nm["comm"] for e in seqs[0])
    names = [e[1] for e in seqs[0]]
    # every reduce-scatter precedes the loss all-reduce, every all-gather follows it (the updates wait for all compute)
    i_loss = names.index("allreduce_sum")
    assert set(names[:i_loss]) == {"reduce_scatter_sum"} and set(names[i_loss + 1:]) == {"all_gather"}
    rs = [(e[2], e[3]) for e in seqs[0] if e[1] == "reduce_scatter_sum"]
    ag = [(e[2], e[3]) for e in seqs[0] if e[1] == "all_gather"]
    # one all-gather per reduce-scatter, same element ranges (of w instead of g), world equal 256-byte-aligned shards
    for k in KEYS:
        g0, _ = r["g_range"][k]
        w0, n_pad = r["w_range"][k]
        rk = sorted((p - g0, n) for p, n in rs if g0 <= p < g0 + 4 * n_pad)
        ak = sorted((p - w0, n) for p, n in ag if w0 <= p < w0 + 4 * n_pad)
        assert rk == ak and len(rk) >= 1
        assert rk[0][0] == 0 and all(a[0] + 4 * a[1] == b_[0] for a, b_ in zip(rk, rk[1:])) and rk[-1][0] + 4 * rk[-1][1] == 4 * n_pad
        assert all(n % (2 * 64) == 0 for _, n in rk)
    # a shard update sits between its bucket's reduce-scatter and its all-gather, on the communication stream, and no stage
    # stream runs an optimiser kernel in this form; the stage streams wait for the communication stream after the last gather
    log = r["log"]
    upd = [i for i, e in enumerate(log) if e[1] == "rmsprop"]
    assert upd and all(log[i][0] == nm["comm"] for i in upd)
    # the gathers run in FORWARD order -- the generators of both stages, then the discriminators; inside a net by ascending
    # offset (first layers first) -- and a per-net event is recorded on the communication stream right behind a net's last gather
    gath = [(i, e) for i, e in enumerate(log) if e[1] == "all_gather"]
    net_of = lambda ptr: next(k for k in KEYS if r["w_range"][k][0] <= ptr < r["w_range"][k][0] + 4 * r["w_range"][k][1])
    nets_seq = [net_of(e[2]) for _, e in gath]
    firsts = [k for j, k in enumerate(nets_seq) if k not in nets_seq[:j]]
    assert firsts == ["dcgan_gen", "p2p_gen", "dcgan_disc", "p2p_disc"], firsts
    recs = [(i, e[2]) for i, e in enumerate(log) if e[1] == "event_record"]
    assert len(recs) == 5 and all(log[i][0] == nm["comm"] for i, _ in recs)
    # the first record sits directly behind the loss all-reduce ("losses_dev has been read": the next step's loss kernels of
    # BOTH stage streams wait for it, whichever nets were exchanged); the other four are the nets' gather events
    i_lr = next(i for i, e in enumerate(log) if e[1] == "allreduce_sum")
    assert recs[0][0] == i_lr + 1
    loss_ev, recs = recs[0][1], recs[1:]
    for k in KEYS:
        idx = [i for (i, e), kk in zip(gath, nets_seq) if kk == k]
        offs = [log[i][2] for i in idx]
        assert offs == sorted(offs)                                     # ascending offsets: the net's first layers first
        assert log[idx[-1] + 1][1] == "event_record"                    # the net's event directly behind its last gather
    # ... which the stage streams wait for per net, in front of the net's forward, in THIS program (the next replay's waits
    # see this step's records): generator before discriminator on both stage streams, every wait before the first collective;
    # there is no wait for the whole communication stream at the end of the step any more
    waits = [(i, e) for i, e in enumerate(log) if e[1] == "event_wait"]
    lw = [(i, e) for i, e in waits if e[2] == loss_ev]
    assert sorted(e[0] for _, e in lw) == sorted([nm["A"], nm["B"]])       # one wait per stage stream, ahead of its forward
    assert all(i < min(j for j, e in waits if e[2] != loss_ev and e[0] == lane) for (i, e_), lane in ((w, w[1][0]) for w in lw))
    waits = [(i, e) for i, e in waits if e[2] != loss_ev]
    assert len(waits) == 4
    first_coll = min(i for i, e in enumerate(log) if e[1] in ("reduce_scatter_sum", "allreduce_sum"))
    assert all(i < first_coll for i, _ in waits)
    by_lane = {lane: [e[2] for _, e in waits if e[0] == lane] for lane in (nm["A"], nm["B"])}
    rec_order = [ev for _, ev in recs]                                  # G, U, D, P
    assert by_lane[nm["A"]] == [rec_order[0], rec_order[2]] and by_lane[nm["B"]] == [rec_order[1], rec_order[3]]
    last_gather = max(i for i, _ in gath)
    assert not any(i > last_gather and e[1] == "wait_for" and e[2] == nm["comm"] for i, e in enumerate(log))
    first_gather = min(i for i, e in enumerate(log) if e[1] == "all_gather")
    waits = [i for i, e in enumerate(log[:first_gather]) if e[0] == nm["comm"] and e[1] == "wait_for"]
    for lane in (nm["A"], nm["B"]):                                    # ... and the gathers wait for every reader of the old weights
        assert any(log[i][2] == lane for i in waits)


# ---- world 8 (the node the step is built for): shard padding to world x 64 elements and the sub-bucket cuts have 8-way
# geometry that two ranks never exercise --------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def world8(tmp_path_factory):
    out = {}
    for i, mode in enumerate(("allreduce", "rs_ag")):
        d = tmp_path_factory.mktemp("dp_world8_" + mode)
        world, port = 8, 37500 + (os.getpid() % 2000) + i
        tmp_.spawn(_worker, args=(world, port, str(d), 2048.0 / 2 ** 20, mode), nprocs=world, join=True)
        out[mode] = [pickle.load(open(os.path.join(str(d), "r%d.pkl" % r), "rb")) for r in range(world)]
    return out


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_eight_ranks_same_program_identical_replicas_mean_update(world8, mode):
    rs = world8[mode]
    assert len(rs) == 8
    kinds = ("allreduce_sum", "reduce_scatter_sum", "all_gather")
    seq0 = [e[1:] for e in rs[0]["log"] if e[1] in kinds]
    for r in rs:
        assert r["crc_before"][0] != r["crc_before"][1]              # eight different initialisations ...
        assert r["crc_after"][0] == r["crc_after"][1]                # ... one after the broadcast ...
        assert r["crc_end"][0] == r["crc_end"][1]                    # ... and still one after the step
        # the collective sequence (kind, element offset inside its buffer, count) is the same program on every rank
        seq = [e[1:] for e in r["log"] if e[1] in kinds]
        assert [(e[0], e[2]) for e in seq] == [(e[0], e[2]) for e in seq0]
        for k in KEYS:
            assert np.array_equal(r["w"][k], rs[0]["w"][k])
    # reduced buckets = the sum over the eight ranks of what each put in (fp32 sums in rank order differ from a float64 sum by
    # rounding only); the all-reduce form keeps the reduced bucket in place on every rank
    if mode == "allreduce":
        for k in KEYS:
            tot = np.sum([r["sent"][k].astype(np.float64) for r in rs], axis=0)
            assert np.linalg.norm(rs[3]["g"][k] - tot) <= 1e-6 * np.linalg.norm(tot), k
    # the PatchGAN-on-real-pairs bucket (no BatchNorm, no generator): mean over the eight shards == full-batch gradient
    if mode == "allreduce":
        A, B = _pairs(32)
        full = _patchgan_real_grads(rs[0]["pvalues"], A, B)
        for got, want in zip(rs[5]["pgrad_avg"], full):
            assert got.shape == want.shape
            assert np.linalg.norm(got - want) <= 4e-6 * np.linalg.norm(want) + 1e-12      # float32 buckets, eight-way sums


def test_eight_rank_sharded_update_equals_the_allreduce_form(world8):
    ar, sh = world8["allreduce"], world8["rs_ag"]
    for k in KEYS:
        assert np.array_equal(sh[0]["w0"][k], ar[0]["w0"][k])           # same seeds: same start
        # every element updated once with the same arithmetic.  (Bit-identical at two ranks, where a + b has one order; an
        # eight-way fp32 sum is ordered differently by a ring all-reduce and a reduce-scatter -- gloo here, RCCL on the node --
        # so the two forms agree to rounding of the reduced gradient, amplified by RMSprop's g / sqrt(0.1 g^2) at step one)
        d = np.abs(sh[0]["w"][k].astype(np.float64) - ar[0]["w"][k])
        assert d.max() <= 1e-6 * (1 + np.abs(ar[0]["w"][k]).max()), (k, d.max())
        assert not np.array_equal(sh[0]["w"][k], sh[0]["w0"][k])
        # the optimiser state is held exactly once across the eight ranks
        accs = np.stack([r["acc"][k] for r in sh])
        assert np.all((accs != 0).sum(axis=0) <= 1)
        assert np.abs(accs.sum(axis=0) - ar[0]["acc"][k]).max() <= 1e-5 * np.abs(ar[0]["acc"][k]).max()
    r = sh[0]
    seq = [e for e in r["log"] if e[1] in ("reduce_scatter_sum", "all_gather")]
    rsx = [(e[2], e[3]) for e in seq if e[1] == "reduce_scatter_sum"]
    for k in KEYS:
        g0, _ = r["g_range"][k]
        _, n_pad = r["w_range"][k]
        rk = sorted((p - g0, n) for p, n in rsx if g0 <= p < g0 + 4 * n_pad)
        # contiguous cover of the padded buffer; every sub-bucket is eight equal, 256-byte-aligned shards
        assert rk and rk[0][0] == 0 and rk[-1][0] + 4 * rk[-1][1] == 4 * n_pad
        assert all(a[0] + 4 * a[1] == b_[0] for a, b_ in zip(rk, rk[1:]))
        assert all(n % (8 * 64) == 0 for _, n in rk), (k, rk)
        assert n_pad % (8 * 64) == 0


# ---- bench.py's timed loop at N > 1: resident batches rotating through two plans (step.py upload_resident_async +
# ---- enqueue_train_uploaded) with the exchange program inside every step -------------------------------------------------------
def _worker_rotating(rank, world, port, out_dir, exchange_mode):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from gan_heightmaps_amd import updates
    from gan_heightmaps_amd.step import GanStep
    from tests.fake_device import host_device_class
    HostDevice = host_device_class()
    dev, cdev = HostDevice(0), HostDevice(0)

    class GlooComm:
        def __init__(self):
            self.dev, self.rank, self.world = cdev, rank, world

        def max_scalar(self, v):
            t = torch.tensor([float(v)], dtype=torch.float64)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            return float(t[0])

    G, D, U, P = _nets(seed=7 if rank == 0 else 1000 + rank)
    spec = updates.rmsprop(learning_rate=updates.shared(1e-2))
    eng = GanStep(dev, G, D, U, P, 100, True, 'l1', spec, 'both', comm=GlooComm(), use_graph=False, two_streams=True,
                  side_streams=True, bucket_mb=2048.0 / 2 ** 20, exchange_mode=exchange_mode)
    eng.broadcast_parameters()
    plans = [eng.built(4, 0), eng.built(4, 1)]
    rng = np.random.RandomState(3 + rank)
    pool = [(dev.tensor(rng.rand(4, 24)), dev.tensor(rng.rand(4, 1, 32, 32)), dev.tensor(rng.randn(4, 3, 32, 32))) for _ in range(3)]
    # (the compute kernels are no-ops: give every step's gradient buckets rank-dependent contents so that the updates move)
    log0 = len(HostDevice.shared.log)
    eng.upload_resident_async(plans[0], *pool[0])
    for k in range(4):
        for key in KEYS:
            st = eng.stores[key]
            st.g.set(np.random.RandomState(100 * k + rank).randn(st.n_pad).astype(np.float32))
        eng.enqueue_train_uploaded(plans[k & 1])
        eng.upload_resident_async(plans[(k + 1) & 1], *pool[(k + 1) % 3])
    eng.sync()
    x_seen = [plans[k & 1].x.numpy().copy() for k in (0, 1)]
    out = {"rank": rank, "crc": eng.replica_checksums(), "log": HostDevice.shared.log[log0:], "comm": cdev.name,
           "x": x_seen, "pool_x": [p[1].numpy().copy() for p in pool]}
    eng.close_pipeline()
    with open(os.path.join(out_dir, "r%d.pkl" % rank), "wb") as f:
        pickle.dump(out, f)
    tdist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag", "allreduce_bf16"])
def test_rotating_resident_batches_with_the_exchange_inside_every_step(tmp_path, mode):
    """What ``bench.py --gpus N`` times since round 6: step k runs on plan k & 1 with batch k % len(pool), copied device-to-device
    on the copy stream while step k - 1 runs.  Two gloo ranks, both exchange forms: identical collective sequences on both ranks
    over four steps, replicas bit-identical at the end, every plan holding the batch the rotation says."""
    world, port = 2, 35500 + (os.getpid() % 2000) + ["allreduce", "rs_ag", "allreduce_bf16"].index(mode)
    tmp_.spawn(_worker_rotating, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    out = [pickle.load(open(os.path.join(str(tmp_path), "r%d.pkl" % r), "rb")) for r in range(world)]
    coll = ("allreduce_sum", "allreduce_sum_bf16", "reduce_scatter_sum", "all_gather")
    seqs = [[(e[1],) + tuple(e[3:]) for e in r["log"] if e[1] in coll] for r in out]      # (sizes; pointers differ per process)
    assert seqs[0] == seqs[1] and len(seqs[0]) >= 4 * 5
    assert all(e[0] == r["comm"] for r in out for e in r["log"] if e[1] in coll)
    assert out[0]["crc"] == out[1]["crc"] and out[0]["crc"][0] == out[0]["crc"][1]
    if mode == "allreduce_bf16":            # every gradient sub-bucket travels as bf16, the losses in fp32
        names = [e[0] for e in seqs[0]]
        assert names.count("allreduce_sum") == 4 and names.count("allreduce_sum_bf16") >= 4 * 4
    for r in out:
        # after steps 0..3 and the staging of batch 4: plan 0 holds batch 4 % 3 = 1, plan 1 holds batch 3 % 3 = 0
        assert np.array_equal(r["x"][0], r["pool_x"][1]) and np.array_equal(r["x"][1], r["pool_x"][0])
        copies = [e for e in r["log"] if e[1] == "d2d"]
        assert len(copies) == 3 * 5
