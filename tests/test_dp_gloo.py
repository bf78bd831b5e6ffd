"""Data-parallel semantics on CPU with 2 processes (gloo): each rank differentiates ITS shard, the flat
gradient buckets are summed and scaled by 1/world -- exactly what GanStep does with ghm_allreduce_sum and
grad_scale -- and the result must equal the full-batch gradient for the BatchNorm-free nets (discriminators).
For the BatchNorm nets the per-rank batch statistics make the sharded result differ: replicas keep the
reference's batch-4 semantics, which this test documents."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as tdist  # noqa: E402
import torch.multiprocessing as tmp_  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL = dict(in_shp=32, latent_dim=24,
             gen_dcgan=dict(nch=16, div=[2, 2, 4]), disc_dcgan=dict(nch=16, div=[4, 2, 2]),
             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from oracle import step as S
    from gan_heightmaps_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = S.default_cfg(**SMALL)
    st = S.init_state(cfg, seed=7, dtype=np.float64)            # identical seeded init on every rank
    Z, X, Y = S.synthetic_batch(4 * world, cfg, seed=3, dtype=np.float64)
    z, x, y = dist.shard_batch([Z, X, Y], rank, world)
    fw = S.forward(st, z, x, y)
    grads = S.gradients(fw, st)
    losses = torch.tensor(S.losses_of(fw))
    tdist.all_reduce(losses)
    out = {}
    for key in S.NET_ORDER:
        flat = torch.from_numpy(np.concatenate([g.ravel() for g in grads[key]]))
        tdist.all_reduce(flat)                                   # bucket sum (RCCL on the GPU)
        out["%s_%s" % key] = (flat / world).numpy()              # grad_scale = 1/world in the optimiser
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), losses=(losses / world).numpy(), **out)
    tdist.destroy_process_group()


def test_sharded_gradients_equal_full_batch(tmp_path):
    from oracle import step as S
    world, port = 2, 29500 + (os.getpid() % 2000)
    tmp_.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / "dp.npz"))
    cfg = S.default_cfg(**SMALL)
    st = S.init_state(cfg, seed=7, dtype=np.float64)
    Z, X, Y = S.synthetic_batch(4 * world, cfg, seed=3, dtype=np.float64)
    def rel(a, b):
        return np.linalg.norm(a - b) / np.linalg.norm(b)

    # what the data-parallel step must compute: every replica differentiates ITS shard with ITS
    # OWN BatchNorm statistics, buckets are summed, the optimiser scales by 1 / world -- i.e. the MEAN over the shards of the
    # single-process step on each shard (not the full-batch step: the BatchNorm generators see batch-4 statistics per
    # replica, the reference's semantics).  Recomputed here shard by shard in one process and compared exactly.
    want_l = np.zeros(5)
    want = {k: 0.0 for k in S.NET_ORDER}
    for r in range(world):
        sl = slice(r * 4, (r + 1) * 4)
        fw = S.forward(st, Z[sl], X[sl], Y[sl])
        g = S.gradients(fw, st)
        want_l += np.asarray(S.losses_of(fw)) / world
        for k in S.NET_ORDER:
            want[k] = want[k] + np.concatenate([t.ravel() for t in g[k]]) / world
    assert rel(got['losses'], want_l) < 1e-6          # (the worker reduces the five losses as a float32 tensor)
    for k in S.NET_ORDER:
        assert got["%s_%s" % k].shape == want[k].shape
        assert rel(got["%s_%s" % k], want[k]) < 1e-12, k
    # and it is NOT the full-batch gradient for the BatchNorm generators (local statistics), which is the documented semantics
    full = S.gradients(S.forward(st, Z, X, Y), st)
    k = ('dcgan', 'gen')
    assert rel(got["%s_%s" % k], np.concatenate([t.ravel() for t in full[k]])) > 1e-6


def _worker_nobn(rank, world, port, out_dir):
    """BatchNorm-free path: PatchGAN D-loss on real pairs only -> sharded mean of gradients is exact."""
    sys.path.insert(0, ROOT)
    from oracle import nets, ops
    from oracle import tape as T
    from gan_heightmaps_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    sp = nets.patchgan_spec(32, True, False, 4, (1, 2))
    params = sp.init(np.random.RandomState(5), np.float64)
    A, B = rng.rand(8, 1, 32, 32), rng.randn(8, 3, 32, 32)
    a, b = dist.shard_batch([A, B], rank, world)
    P = [T.leaf(p) for p in params]
    out, _ = nets.patchgan_fwd(P, T.leaf(a), T.leaf(b), act='linear', mul_factor=(1, 2))
    loss = T.scalar_loss(out, lambda v: ops.squared_error_mean(v, 1.0))
    T.backward(loss)
    flat = torch.from_numpy(np.concatenate([p.g.ravel() for p in P]))
    tdist.all_reduce(flat)
    if rank == 0:
        np.save(os.path.join(out_dir, "nobn.npy"), (flat / world).numpy())
    tdist.destroy_process_group()


def test_sharded_gradients_exact_without_batchnorm(tmp_path):
    from oracle import nets, ops
    from oracle import tape as T
    world, port = 2, 31500 + (os.getpid() % 2000)
    tmp_.spawn(_worker_nobn, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / "nobn.npy"))
    rng = np.random.RandomState(0)
    sp = nets.patchgan_spec(32, True, False, 4, (1, 2))
    params = sp.init(np.random.RandomState(5), np.float64)
    A, B = rng.rand(8, 1, 32, 32), rng.randn(8, 3, 32, 32)
    P = [T.leaf(p) for p in params]
    out, _ = nets.patchgan_fwd(P, T.leaf(A), T.leaf(B), act='linear', mul_factor=(1, 2))
    loss = T.scalar_loss(out, lambda v: ops.squared_error_mean(v, 1.0))
    T.backward(loss)
    full = np.concatenate([p.g.ravel() for p in P])
    assert np.linalg.norm(got - full) <= 1e-12 * np.linalg.norm(full)
