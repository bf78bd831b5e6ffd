#!/usr/bin/env python
"""The BASELINE configuration, end to end from the reference's own files: experiments.py (as __main__,
`test1_nobn_bilin_both train`) hands its keyword arguments to the reference's Pix2Pix.__init__ (pix2pix.py:24-157,
via lib2to3), which builds the four 512x512 networks with the reference's architectures/{dcgan,p2p,layers}.py and
defines train_fn -- all executed here on tests/golden/symtheano.py (oracle ops, float64).  One train_fn call on
a seeded batch of 2 is recorded:

    python tests/golden/make_reference_step_fullsize.py     # build container only; ~10 min of CPU
    -> tests/golden/reference_step_fullsize.npz  (5 losses; per parameter tensor after the step: sum, L2 norm and
       eight sampled elements; the same for the tensors before the step)

tests/test_gpu_fullsize.py runs the same step through the HIP path on the MI355X and compares.
"""
import os
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

SEED, BATCH, DATA_SEED = 11, 2, 42


def summarise(values):
    out = {}
    for k, v in values.items():
        v64 = np.asarray(v, np.float64).ravel()
        idx = np.linspace(0, v64.size - 1, 8).astype(np.int64)
        out[k] = np.concatenate([[v64.sum(), np.sqrt((v64 * v64).sum())], v64[idx]])
    return out


def main():
    import symtheano as sym
    import make_reference_step as RS
    from gan_heightmaps_amd import init as INIT
    from oracle import step as S
    P, _ = RS.load_reference(sym)
    rec = {}

    class Recorder:
        def __init__(self, **kw):
            rec["kw"] = kw

        def train(self, *a, **k):
            pass
    sys.modules["pix2pix"].Pix2Pix = Recorder
    sys.modules["util"].Hdf5Iterator = lambda *a, **k: None      # the data path is not part of this fixture
    argv = sys.argv
    sys.argv = ["experiments.py", "test1_nobn_bilin_both", "train"]
    try:
        runpy.run_path(os.path.join(REF, "experiments.py"), run_name="__main__")
    finally:
        sys.argv = argv
    kw = dict(rec["kw"], verbose=False)
    assert kw["opt"] is sym.rmsprop            # experiments.py:116 picked lasagne.updates.rmsprop -> the stand-in
    INIT.set_rng(np.random.RandomState(SEED))
    m = P.Pix2Pix(**kw)
    cfg = S.default_cfg()
    Z, X, Y = S.synthetic_batch(BATCH, cfg, seed=DATA_SEED)
    out = {}
    for k, v in summarise(RS.all_values(m)).items():
        out["before/" + k] = v
    out["train0"] = np.asarray(m.train_fn(Z, X, Y), np.float64)
    print("losses", out["train0"], flush=True)
    for k, v in summarise(RS.all_values(m)).items():
        out["after/" + k] = v
    out["meta"] = np.array([SEED, BATCH, DATA_SEED], np.int64)
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(HERE, "reference_step_fullsize.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
