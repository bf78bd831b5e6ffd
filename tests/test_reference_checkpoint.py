"""f2 pinned on the reference: tests/golden/reference_checkpoint.model was written by the REFERENCE's own
Pix2Pix.save_model (executed by tests/golden/make_reference_checkpoint.py).  This package's load_model must restore
exactly those values (all three modes), and a file written by this package's save_model must be readable by the
reference's load_model (checked where /root/reference exists)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from gan_heightmaps_amd import layers as L
from gan_heightmaps_amd.pix2pix import Pix2Pix

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "reference_checkpoint.model")
VALS = np.load(os.path.join(HERE, "golden", "reference_checkpoint.npz"))


def _gen():
    spec = importlib.util.spec_from_file_location("make_reference_checkpoint", os.path.join(HERE, "golden", "make_reference_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _model(seed):
    nets = _gen().build_nets(seed)
    m = Pix2Pix.__new__(Pix2Pix)
    m.dcgan, m.p2p = nets["dcgan"], nets["p2p"]
    return m


def _values(m):
    return {"%s/%s/%03d" % (a, b, i): v for a in ("dcgan", "p2p") for b in ("gen", "disc")
            for i, v in enumerate(L.get_all_param_values(getattr(m, a)[b]))}


@pytest.mark.parametrize("mode", ["both", "dcgan", "p2p"])
def test_load_model_reads_the_reference_written_checkpoint(mode):
    m = _model(seed=7)
    before = _values(m)
    assert any(not np.array_equal(before[k], VALS[k]) for k in VALS.files)
    m.load_model(CKPT, mode=mode)
    after = _values(m)
    assert sorted(after) == sorted(VALS.files)
    for k in VALS.files:
        loaded = mode == "both" or k.startswith(mode + "/")
        assert after[k].shape == VALS[k].shape and after[k].dtype == VALS[k].dtype
        assert np.array_equal(after[k], VALS[k] if loaded else before[k]), k


def test_save_model_layout_equals_the_reference_layout(tmp_path):
    """same dict structure, list order, shapes and dtypes as the reference-written file"""
    import gzip
    import pickle
    m = _model(seed=123)
    m.save_model(str(tmp_path / "mine.model"))
    mine = pickle.load(gzip.open(tmp_path / "mine.model"), encoding="latin1")
    ref = pickle.load(gzip.open(CKPT), encoding="latin1")
    assert sorted(mine) == sorted(ref) == ["dcgan", "p2p"]
    for a in ref:
        assert sorted(mine[a]) == sorted(ref[a]) == ["disc", "gen"]
        for b in ref[a]:
            assert len(mine[a][b]) == len(ref[a][b])
            for u, v in zip(mine[a][b], ref[a][b]):
                assert np.array_equal(u, v) and u.dtype == v.dtype       # seed 123 on both sides


def test_checkpoint_globals_resolve_in_the_reference_environment(tmp_path):
    """The reference runs Python 2 with numpy <= 1.16: every pickle GLOBAL of a checkpoint written here must name a
    module that exists there.  numpy >= 2 reduces an ndarray through numpy._core.multiarray._reconstruct, which does
    not; save_model writes numpy.core.multiarray instead (valid in every numpy up to 2.x)."""
    import gzip
    import pickle
    import pickletools
    m = _model(seed=11)
    m.save_model(str(tmp_path / "mine.model"))
    raw = gzip.open(tmp_path / "mine.model").read()
    globals_ = {arg for op, arg, _ in pickletools.genops(raw) if op.name == "GLOBAL"}
    assert globals_ == {"numpy.core.multiarray _reconstruct", "numpy ndarray", "numpy dtype", "_codecs encode"}, globals_
    assert raw[:2] == b"\x80\x02"                                     # protocol 2 == py2 HIGHEST_PROTOCOL
    # what a stock numpy-2 pickle would have contained (the defect this guards against)
    stock = {arg for op, arg, _ in pickletools.genops(pickle.dumps(np.zeros(2, np.float32), 2)) if op.name == "GLOBAL"}
    if int(np.__version__.split(".")[0]) >= 2:
        assert "numpy._core.multiarray _reconstruct" in stock
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        back = pickle.loads(raw, encoding="latin1")
    for u, v in zip(back["p2p"]["gen"], L.get_all_param_values(m.p2p["gen"])):
        assert np.array_equal(u, v) and u.dtype == v.dtype


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_reference_load_model_reads_our_checkpoint(tmp_path):
    G = _gen()
    P = G.load_reference_pix2pix()
    src = _model(seed=5)
    src.save_model(str(tmp_path / "ours.model"))
    nets = G.build_nets(9)
    r = P.Pix2Pix.__new__(P.Pix2Pix)
    r.dcgan, r.p2p = nets["dcgan"], nets["p2p"]
    r.load_model(str(tmp_path / "ours.model"), mode="both")
    want = _values(src)
    got = {"%s/%s/%03d" % (a, b, i): v for a in ("dcgan", "p2p") for b in ("gen", "disc")
           for i, v in enumerate(L.get_all_param_values(nets[a][b]))}
    for k in want:
        assert np.array_equal(got[k], want[k]), k
