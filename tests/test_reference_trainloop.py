"""SURVEY 8 a11 pinned on the reference: tests/golden/reference_trainloop.json is the observable behaviour of the
REFERENCE's Pix2Pix.train (+ generate_atob / generate_gz / util.plot_grid) executed by
tests/golden/make_reference_trainloop.py against recording stand-ins.  This package's Pix2Pix.train, driven by the
same stand-ins, must produce the same event sequence (iterator draws, sampler calls, compiled-function calls),
the same results.txt rows, the same files and the same checkpoint requests."""
import importlib.util
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "reference_trainloop.json")))


def _harness_module():
    spec = importlib.util.spec_from_file_location("make_reference_trainloop", os.path.join(HERE, "golden", "make_reference_trainloop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_train_loop_matches_the_reference_event_for_event(tmp_path):
    from gan_heightmaps_amd.pix2pix import Pix2Pix
    from gan_heightmaps_amd.step import TRAIN_KEYS
    M = _harness_module()
    h = M.Harness()
    m = Pix2Pix.__new__(Pix2Pix)
    h.attach(m, TRAIN_KEYS)
    assert list(TRAIN_KEYS) == ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_recon', 'p2p_disc']      # pix2pix.py:157
    out = h.run(m, str(tmp_path / "out"), str(tmp_path / "models"))
    assert out["header"] == FIX["header"]
    assert out["rows"] == FIX["rows"]
    assert out["saved"] == FIX["saved"]
    assert out["files"] == FIX["files"]
    got, ref = json.loads(json.dumps(out["log"])), FIX["log"]
    assert len(got) == len(ref)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a == b, "event %d: %r != %r" % (i, a, b)


def test_quick_run_and_no_dumps(tmp_path):
    """quick_run breaks after one minibatch per loop (pix2pix.py:209-210); dump_images=False leaves the iterators alone"""
    from gan_heightmaps_amd.pix2pix import Pix2Pix
    from gan_heightmaps_amd.step import TRAIN_KEYS
    M = _harness_module()
    h = M.Harness()
    m = Pix2Pix.__new__(Pix2Pix)
    h.attach(m, TRAIN_KEYS)
    out = h.run(m, str(tmp_path / "o"), None, quick_run=True, dump_images=False)
    kinds = [e[0] for e in out["log"]]
    assert kinds.count("train_fn") == 3 and kinds.count("loss_fn") == 3 and "gen_fn" not in kinds
    assert out["files"] == ["results.txt"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_fixture_is_reproducible_from_the_reference(tmp_path):
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_reference_trainloop.py")],
                         capture_output=True, text=True, env=dict(os.environ, GHM_FIXTURE_OUT=str(tmp_path / "g.json")))
    assert out.returncode == 0, out.stderr[-2000:]
    new = json.load(open(tmp_path / "g.json"))
    for k in ("header", "rows", "files", "saved", "log", "plateau"):
        assert new[k] == FIX[k], k


@pytest.mark.parametrize("case", range(4))
def test_reduce_lr_on_plateau_matches_the_reference_fixture(case):
    """the committed trajectories of the reference's ReduceLROnPlateau (reference_trainloop.json "plateau")"""
    from gan_heightmaps_amd.keras_ports import ReduceLROnPlateau
    from gan_heightmaps_amd.updates import shared
    import numpy as np
    M = _harness_module()
    rec = FIX["plateau"][case]
    assert rec["kw"] == M.PLATEAU_CASES[case]
    lr = shared(np.float32(0.01))
    cb = ReduceLROnPlateau(lr, **rec["kw"])
    cb.on_train_begin()
    for e, (v, want) in enumerate(zip(M.plateau_sequence(len(rec["kw"])), rec["trajectory"])):
        cb.on_epoch_end(v, e + 1)
        assert [float(lr.get_value()), cb.wait, cb.cooldown_counter] == want, (case, e)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("kw", [dict(), dict(mode='min', patience=2, factor=0.5), dict(mode='max', patience=1, cooldown=2),
                                dict(mode='min', patience=0, factor=0.1, min_lr=1e-3, epsilon=0.01)])
def test_reduce_lr_on_plateau_matches_the_reference_class(kw, monkeypatch):
    """keras_ports.ReduceLROnPlateau (keras_ports.py:7-111), executed via lib2to3, against this package's class:
    same learning-rate trajectory and internal counters on random monitor sequences"""
    import numpy as np
    from gan_heightmaps_amd.keras_ports import ReduceLROnPlateau
    from gan_heightmaps_amd.updates import shared
    monkeypatch.setattr(np, "Inf", np.inf, raising=False)      # the reference predates NumPy 2.0 (np.Inf removed)
    M = _harness_module()
    ref_mod = M.load_py2("/root/reference/keras_ports.py", "reference_keras_ports")
    rng = np.random.RandomState(len(kw))
    seq = np.round(1.5 + 0.2 * np.sin(np.arange(60) / 3.0) + 0.05 * rng.randn(60), 3).tolist()
    lr_a, lr_b = shared(np.float32(0.01)), shared(np.float32(0.01))
    a, b = ref_mod.ReduceLROnPlateau(lr_a, **kw), ReduceLROnPlateau(lr_b, **kw)
    a.on_train_begin(); b.on_train_begin()
    for e, v in enumerate(seq):
        a.on_epoch_end(v, e + 1); b.on_epoch_end(v, e + 1)
        assert float(lr_a.get_value()) == float(lr_b.get_value()), e
        assert (a.wait, a.cooldown_counter, a.best) == (b.wait, b.cooldown_counter, b.best), e
    assert float(lr_b.get_value()) < 0.01 or not kw
