"""gan_heightmaps_amd/h5lite.py against HDF5 files written by the real library (tests/golden/hdf5/*.h5: h5py 3.3 /
HDF5 1.10.6 in the build container, make_hdf5_fixtures.py next to them).  Expected contents are rebuilt from the
generator's seeds, so nothing here needs libhdf5."""
import importlib.util
import os

import numpy as np
import pytest

from gan_heightmaps_amd import h5lite

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
_spec = importlib.util.spec_from_file_location("make_hdf5_fixtures", os.path.join(HERE, "make_hdf5_fixtures.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)
ARRAYS = _gen.arrays()


@pytest.mark.parametrize("fname", sorted(ARRAYS))
def test_every_dataset_reads_back_exactly(fname):
    with h5lite.File(os.path.join(HERE, fname), "r") as f:
        for name, want in ARRAYS[fname].items():
            ds = f[name]
            want = np.asarray(want)
            assert ds.shape == want.shape and ds.dtype == want.dtype, (fname, name, ds)
            got = ds[()] if not ds.shape else ds[...]
            assert got.dtype == want.dtype and np.array_equal(got, want), (fname, name)
            assert np.array_equal(np.asarray(ds), want)
        top = set(k.split("/")[0] for k in ARRAYS[fname])
        assert set(f.keys()) == top and all(k in f for k in ARRAYS[fname]) and "nope" not in f
        with pytest.raises(KeyError):
            f["nope"]


def test_reference_recipe_file_is_sliced_like_h5py():
    """util.iterate_hdf5 (util.py:20-42) reads ``X_arr[b*bs:(b+1)*bs]`` and ``X_arr.shape[0]``"""
    want = ARRAYS["ref_layout.h5"]
    f = h5lite.File(os.path.join(HERE, "ref_layout.h5"))
    xt, yt = f["xt"], f["yt"]
    assert len(xt) == 6 and xt.shape == (6, 16, 16, 1) and yt.shape[-1] == 3 and xt.dtype == np.uint8
    for b in range(3):
        assert np.array_equal(xt[b * 2:(b + 1) * 2], want["xt"][b * 2:(b + 1) * 2])
        assert np.array_equal(yt[b * 2:(b + 1) * 2], want["yt"][b * 2:(b + 1) * 2])
    assert np.array_equal(xt[4:100], want["xt"][4:]) and xt[6:8].shape == (0, 16, 16, 1)
    assert np.array_equal(yt[-1], want["yt"][-1]) and np.array_equal(yt[[0, 3, 5]], want["yt"][[0, 3, 5]])
    assert np.array_equal(yt[1:5, ::2, 3], want["yt"][1:5, ::2, 3])
    f.close()


@pytest.mark.parametrize("name", ["plain", "gz", "gzshuf", "gzshuf_fl", "holes"])
def test_chunked_datasets_partial_reads(name):
    want = np.asarray(ARRAYS["chunked.h5"][name])
    with h5lite.File(os.path.join(HERE, "chunked.h5")) as f:
        ds = f[name]
        assert ds.chunks is not None
        n = want.shape[0]
        for lo, hi in [(0, 1), (1, 4), (2, n), (n - 1, n), (0, n), (3, 3)]:
            assert np.array_equal(ds[lo:hi], want[lo:hi]), (name, lo, hi)
        assert np.array_equal(ds[n - 2], want[n - 2]) and np.array_equal(ds[-1], want[-1])
        assert np.array_equal(ds[::2], want[::2]) and np.array_equal(ds[[0, 2, n - 1]], want[[0, 2, n - 1]])
        assert np.array_equal(ds[1:4, 1], want[1:4, 1]) and np.array_equal(ds[..., 0], want[..., 0])
        with pytest.raises(IndexError):
            ds[n]


def test_nested_groups_and_many_members():
    with h5lite.File(os.path.join(HERE, "chunked.h5")) as f:
        g = f["grp"]
        assert isinstance(g, h5lite.Group) and g.keys() == ["inner"] and "inner/f64" in g
        assert np.array_equal(f["grp"]["inner"]["f64"][...], ARRAYS["chunked.h5"]["grp/inner/f64"])
        assert f["grp/inner"].name == "/grp/inner"
    with h5lite.File(os.path.join(HERE, "many.h5")) as f:
        assert len(f) == 41 and sorted(f.keys()) == sorted(ARRAYS["many.h5"])
        deep = f["deep"]
        assert len(deep._chunk_index()) == 300            # a chunk per row: the index B-tree has internal nodes
        assert np.array_equal(deep[123:257], ARRAYS["many.h5"]["deep"][123:257])


def test_not_hdf5_and_unsupported_are_loud(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(IOError):
        h5lite.File(str(p))
    with pytest.raises(ValueError):
        h5lite.File(os.path.join(HERE, "ref_layout.h5"), "w")


def test_get_iterators_reads_an_hdf5_file_without_h5py(monkeypatch):
    """experiments.get_iterators (experiments.py:10-18) on a real .h5: the four arrays arrive through h5lite when
    h5py is not importable (host side only here: the iterator's source arrays)."""
    import builtins
    from gan_heightmaps_amd import experiments
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py is not installed")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_h5py)
    d = experiments.open_dataset(os.path.join(HERE, "ref_layout.h5"))
    want = ARRAYS["ref_layout.h5"]
    for k in ("xt", "yt", "xv", "yv"):
        assert d[k].shape == want[k].shape and np.array_equal(d[k][0:2], want[k][0:2])


@pytest.mark.gpu
def test_device_iterator_over_an_hdf5_file_equals_the_in_memory_one():
    """the training input path of experiments.py:10-18 end to end: .h5 on disk -> h5lite slices -> device batches"""
    from gan_heightmaps_amd import device, experiments
    from gan_heightmaps_amd import data as D
    dev = device.Device(0)
    try:
        want = ARRAYS["ref_layout.h5"]
        it_train, it_val = experiments.get_iterators(os.path.join(HERE, "ref_layout.h5"), 2, True, False, da=True,
                                                     in_shp=16, device=dev)
        imgen = D.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
        ref_train = D.Hdf5Iterator(want["xt"], want["yt"], 2, imgen, True, False, device=dev)
        ref_val = D.Hdf5Iterator(want["xv"], want["yv"], 2, imgen, True, False, device=dev)
        assert it_train.N == 6 and it_val.N == 2
        for it, ref in ((it_train, ref_train), (it_val, ref_val)):
            for _ in range(5):
                a, b = it.next()
                ra, rb = ref.next()
                assert a.shape == ra.shape and np.array_equal(a, ra) and np.array_equal(b, rb)
    finally:
        dev.close()
