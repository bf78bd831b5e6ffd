"""Rows a1-a3 pinned on the reference: tests/golden/reference_step.npz holds what the compiled functions defined by
the REFERENCE's Pix2Pix.__init__ return and do to the parameters when that code is executed on the oracle's ops
(tests/golden/make_reference_step.py + symtheano.py).  oracle/step.py -- the independently written restatement
the HIP path is tested against -- must agree: five losses of train_fn (two consecutive steps: optimiser state
and BN running statistics carry over), every parameter after the first step and at the end, loss_fn, and the
four generator functions, for three optimiser / mode / loss variants."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from oracle import step as S

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "reference_step.npz"))


def _gen():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    spec = importlib.util.spec_from_file_location("make_reference_step", os.path.join(HERE, "golden", "make_reference_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


G = _gen()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def state_values(state):
    out = {}
    for a, b in S.NET_ORDER:
        for i, v in enumerate(state['params'][a][b]):
            out["%s/%s/%03d" % (a, b, i)] = v
    return out


def compare_params(state, name, tag, cfg):
    sp = S.specs(cfg)
    got = state_values(state)
    for key in S.NET_ORDER:
        for i, kind in enumerate(sp[key].kinds):
            k = "%s/%s/%03d" % (key[0], key[1], i)
            ref = FIX["%s/%s/%s" % (name, tag, k)]
            assert got[k].shape == ref.shape, k
            # fp32 parameter storage on the reference side, float64 state here
            assert rel(got[k], ref) < 2e-6 or np.abs(got[k] - ref).max() < 1e-7, (name, tag, k, kind)


@pytest.mark.parametrize("name", sorted(G.VARIANTS))
def test_oracle_step_equals_the_reference_wiring(name):
    cfg = S.default_cfg(**dict(G.CFG_OVER, **G.VARIANTS[name]))
    state = S.init_state(cfg, G.SEED, np.float64)
    for step in range(2):
        Z, X, Y = S.synthetic_batch(G.BATCH, cfg, seed=100 + step)
        res = S.train_step(state, Z, X, Y, dtype=np.float64, update=True)
        assert rel(res['losses'], FIX["%s/train%d" % (name, step)]) < 1e-6, (name, step)
        if step == 0:
            compare_params(state, name, "after0", cfg)
    Z, X, Y = S.synthetic_batch(G.BATCH, cfg, seed=200)
    res = S.train_step(state, Z, X, Y, dtype=np.float64, update=False)
    assert rel(res['losses'], FIX[name + "/loss"]) < 1e-6
    fw = S.forward(state, Z, X, Y, np.float64, deterministic=True)
    assert rel(fw['gz'].v, FIX[name + "/z_fn_det"]) < 1e-6
    assert rel(fw['ux'].v, FIX[name + "/gen_fn_det"]) < 1e-6
    # z_fn / gen_fn use batch statistics and move the running statistics once more each (default_updates)
    fw = S.forward(state, Z, X, Y, np.float64)
    assert rel(fw['gz'].v, FIX[name + "/z_fn"]) < 1e-6
    assert rel(fw['ux'].v, FIX[name + "/gen_fn"]) < 1e-6
    S._apply_bn_running(state, fw)
    compare_params(state, name, "final", cfg)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_fixture_is_reproducible_from_the_reference(tmp_path):
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_reference_step.py")],
                         capture_output=True, text=True, env=dict(os.environ, GHM_FIXTURE_OUT=str(tmp_path / "g.npz")))
    assert out.returncode == 0, out.stderr[-2000:]
    new = np.load(tmp_path / "g.npz")
    assert sorted(new.files) == sorted(FIX.files)
    for k in FIX.files:
        assert np.array_equal(new[k], FIX[k], equal_nan=True), k
