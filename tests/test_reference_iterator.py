"""f1 pinned on the reference: tests/golden/reference_iterator.npz holds what the REFERENCE's util.iterate_hdf5
yields (executed unmodified by tests/golden/make_reference_iterator.py with a recording stand-in for Keras'
generator).  The oracle restatement (oracle/keras_aug.py) and the product's host-side planner
(gan_heightmaps_amd.data.Hdf5Iterator.plan_next) must reproduce slice order, normalisation and flow seeds."""
import os

import numpy as np
import pytest

from gan_heightmaps_amd import data as D
from oracle import keras_aug as KA

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "reference_iterator.npz"))
CASES = sorted({k.split("/")[0] for k in FIX.files} - {"rgb"})


def cfg(name):
    N, H, W, bs, ga, gb, gen, steps = (int(v) for v in FIX[name + "/cfg"])
    return N, H, W, bs, bool(ga), bool(gb), bool(gen), steps


@pytest.mark.parametrize("name", [c for c in CASES if cfg(c)[6]])
def test_oracle_restatement_against_the_reference_generator(name, monkeypatch):
    N, H, W, bs, ga, gb, gen, steps = cfg(name)
    seeds = []

    def identity_flow(x, seed, augment):
        seeds.append(int(seed))
        return x, None
    monkeypatch.setattr(KA, "flow_first_batch", identity_flow)
    it = KA.Hdf5IteratorOracle(FIX[name + "/X"], FIX[name + "/Y"], bs, ga, gb, augment=True)
    for s in range(steps):
        a, b = it.next()
        ra, rb = FIX["%s/a%d" % (name, s)], FIX["%s/b%d" % (name, s)]
        assert a.shape == ra.shape and b.shape == rb.shape
        assert np.array_equal(np.asarray(a, np.float32), ra) and np.array_equal(np.asarray(b, np.float32), rb)
    # A and B share one seed per slice (util.py:38-40): the reference logged each twice
    assert [s for s in seeds] == list(FIX[name + "/seeds"])


@pytest.mark.parametrize("name", CASES)
def test_product_planner_slices_and_seeds(name, monkeypatch):
    N, H, W, bs, ga, gb, gen, steps = cfg(name)
    X, Y = FIX[name + "/X"], FIX[name + "/Y"]
    seeds = []

    def identity_plan(imgen, n, seed, h, w):
        seeds.append(int(seed))
        return np.arange(n), np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0], np.float64), (n, 1))
    monkeypatch.setattr(D, "plan_flow", identity_plan)
    it = D.Hdf5Iterator(X, Y, bs, D.ImageDataGenerator() if gen else None, is_a_grayscale=ga, is_b_grayscale=gb)
    assert it.N == N
    for s in range(steps):
        sl, perm, table = it.plan_next()
        a = KA.normalise(X[sl][perm], ga)          # the device kernel's arithmetic is checked in test_gpu_ops / test_data_path
        b = KA.normalise(Y[sl][perm], gb)
        assert np.array_equal(np.asarray(a, np.float32), FIX["%s/a%d" % (name, s)])
        assert np.array_equal(np.asarray(b, np.float32), FIX["%s/b%d" % (name, s)])
    # the reference draws one seed per slice and hands it to both flows
    assert [x for x in seeds for _ in range(2)] == list(FIX[name + "/seeds"])


def test_image_helpers_against_the_reference():
    """util.convert_to_rgb / compose_imgs (util.py:69-99) executed from the reference -> golden arrays"""
    from gan_heightmaps_amd import util as U
    g, c = FIX["rgb/g"], FIX["rgb/c"]
    for key, img, gray in (("rgb/g_gray", g, True), ("rgb/g_tanh", g, False), ("rgb/c_tanh", c, False),
                           ("rgb/c_gray", c, True)):
        got = U.convert_to_rgb(img.copy(), is_grayscale=gray)
        assert got.shape == FIX[key].shape
        assert np.abs(got - FIX[key]).max() < 1e-6, key
    got = U.compose_imgs(g.copy(), c.copy(), is_a_grayscale=True, is_b_grayscale=False)
    assert got.shape == FIX["rgb/compose"].shape and np.abs(got - FIX["rgb/compose"]).max() < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_fixture_is_reproducible_from_the_reference(tmp_path):
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_reference_iterator.py")],
                         capture_output=True, text=True, env=dict(os.environ, GHM_FIXTURE_OUT=str(tmp_path / "g.npz")))
    assert out.returncode == 0, out.stderr[-2000:]
    new = np.load(tmp_path / "g.npz")
    assert sorted(new.files) == sorted(FIX.files)
    for k in FIX.files:
        assert np.array_equal(new[k], FIX[k]), k
