"""the data iterator + augmentation in front of the hot path.
CPU: the oracle restatement of util.iterate_hdf5 + Keras random_transform (oracle/keras_aug.py) and the product's
batch plan (gan_heightmaps_amd/data.py) draw the same numbers in the same order.
GPU: ghm_image_batch against scipy.ndimage.affine_transform(order=0, mode='reflect') + flips."""
import numpy as np
import pytest

from oracle import keras_aug as K
from gan_heightmaps_amd import data as D


def make_arrays(n=6, h=32, w=32, seed=0):
    rng = np.random.RandomState(seed)
    return (rng.randint(0, 256, (n, h, w, 1)).astype(np.uint8), rng.randint(0, 256, (n, h, w, 3)).astype(np.uint8))


def test_oracle_transform_properties():
    x = np.random.RandomState(1).rand(2, 9, 9).astype(np.float32)
    assert np.array_equal(K.apply_transform(x, 0.0, False, False), x)
    assert np.array_equal(K.apply_transform(x, 0.0, True, False), x[:, :, ::-1])
    assert np.array_equal(K.apply_transform(x, 0.0, False, True), x[:, ::-1, :])
    # Keras' centre is (h/2 + 0.5) = 5.0 for h = 9 (true centre 4.0): a half-turn maps out[i] = in[10 - i], i.e. a
    # point reflection shifted by two pixels, the first two rows / columns coming from the reflect border
    r = K.apply_transform(x, np.pi, False, False)
    assert np.array_equal(r[:, 2:, 2:], x[:, ::-1, ::-1][:, :7, :7])
    # normalisation branches (util.py:34-35)
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1)
    assert np.allclose(K.normalise(u8, True).ravel(), np.arange(256) / 255.0)
    assert np.allclose(K.normalise(u8, False).ravel(), (np.arange(256) - 127.5) / 127.5)


def test_product_plan_draws_match_oracle():
    X, Y = make_arrays()
    imgen = D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
    it = D.Hdf5Iterator(X, Y, 4, imgen, True, False)
    ora = K.Hdf5IteratorOracle(X, Y, 4, True, False, augment=True)
    ref_rng = np.random.RandomState(0)
    for _ in range(5):                                   # crosses an epoch boundary (slices reshuffled)
        sl, perm, table = it.plan_next()
        # replay the oracle's bookkeeping by hand
        if not ora._pending:
            ora._pending = K.get_slices(ora.N, ora.bs)
            ora.rnd_state.shuffle(ora._pending)
        osl = ora._pending.pop(0)
        seed = ora.rnd_state.randint(0, 100000)
        assert (sl.start, sl.stop) == (osl.start, osl.stop)
        n = len(range(*sl.indices(6)))
        _, operm, oparams = K.flow_first_batch(np.zeros((n, 1, 4, 4), np.float32), seed)
        assert np.array_equal(perm, operm)
        for row, (theta, hf, vf) in zip(table, oparams):
            m = K.rotation_matrix_centered(theta, 32, 32)
            assert np.allclose(row[:6], [m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2]], rtol=0, atol=1e-12)
            assert (bool(row[6]), bool(row[7])) == (hf, vf)
    with pytest.raises(NotImplementedError):
        D.ImageDataGenerator(rotation_range=10, fill_mode='nearest')
    with pytest.raises(NotImplementedError):
        D.ImageDataGenerator(zoom_range=0.2)


@pytest.mark.gpu
def test_device_iterator_matches_oracle():
    from gan_heightmaps_amd import device
    dev = device.Device(0)
    try:
        X, Y = make_arrays(n=10, h=64, w=48, seed=3)
        for da in (False, True):
            imgen = (D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
                     if da else D.ImageDataGenerator())
            it = D.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
            ora = K.Hdf5IteratorOracle(X, Y, 4, True, False, augment=da)
            mism, total = 0, 0
            for _ in range(6):
                a, b = it.next()
                ra, rb = ora.next()
                assert a.shape == ra.shape and b.shape == rb.shape and a.dtype == np.float32
                for got, ref in ((a, ra), (b, rb)):
                    mism += int((got != ref).sum())
                    total += got.size
                    # every value is one of the normalised uint8 levels: a mismatch can only be a neighbouring pixel
            if not da:
                assert mism == 0                         # identity generator: bit exact, permutation included
            else:
                # nearest-neighbour ties at exact .5 coordinates may resolve differently from scipy's C code
                assert mism / total < 2e-3, (mism, total)
        # flips only: exact
        imgen = D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True)
        it = D.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
        for _ in range(3):
            sl, perm, table = it.plan_next()
            it._pending.insert(0, sl)                    # replay the same slice through the device path
            state = it.rnd_state.get_state()
        # direct kernel check on hand-made transforms incl. 90-degree rotations (exact in fp64)
        x = np.random.RandomState(5).randint(0, 256, (4, 17, 23, 3)).astype(np.uint8)
        rows = [D.transform_row(0.0, False, False, 17, 23), D.transform_row(0.0, True, False, 17, 23),
                D.transform_row(0.0, True, True, 17, 23), D.transform_row(0.7, False, True, 17, 23)]
        table = np.asarray(rows, np.float64)
        src, xf = dev.alloc(x.nbytes), dev.alloc(table.nbytes)
        dev.h2d(src, x)
        dev.h2d(xf, table)
        out = dev.empty((4, 3, 17, 23))
        device.Ops(dev).image_batch(src, 4, 17, 23, 3, xf, True, out)
        got = out.numpy()
        xn = K.normalise(x, False)
        assert np.array_equal(got[0], xn[0])
        assert np.array_equal(got[1], xn[1][:, :, ::-1])
        assert np.array_equal(got[2], xn[2][:, ::-1, ::-1])
        ref3 = K.apply_transform(xn[3], 0.7, False, True)
        assert (got[3] != ref3).mean() < 5e-3
    finally:
        dev.close()
