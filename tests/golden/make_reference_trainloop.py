#!/usr/bin/env python
"""Runs the REFERENCE's Pix2Pix.train / generate_atob / generate_gz (pix2pix.py:187-326), the host-side caller of
the hot path, against recording stand-ins for the compiled functions and the iterators, and writes
down everything observable: the call sequence (which function, which batch, which iterator it was drawn from,
sampler shapes), the results.txt rows, the files created and the checkpoints requested.

    python tests/golden/make_reference_trainloop.py        # build container only: needs /root/reference
    -> tests/golden/reference_trainloop.json

pix2pix.py is Python-2 source (print statements): it is translated in memory by lib2to3 when loaded -- nothing is
written or copied; its `open(..., "wb")` for results.txt becomes text mode, as Python 2 treated it -- and executed with theano/lasagne -> this package's vocabulary, `util` -> the reference's own
util.py, and recorder stubs for keras / keras_ports / skimage.io (absent here).  The object is created without
running __init__ (which would build the Theano graph); `train_keys` is set as pix2pix.py:157 sets it.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def load_py2(path, name):
    from lib2to3 import refactor
    fixers = refactor.get_fixers_from_package("lib2to3.fixes")
    tool = refactor.RefactoringTool(fixers)
    src = open(path).read()
    tree = tool.refactor_string(src if src.endswith("\n") else src + "\n", name)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    # Python 2 wrote str to files opened "wb" (pix2pix.py:234): give the module a text-mode open()
    mod.open = lambda p, mode="r": open(p, mode.replace("b", ""))
    exec(compile(str(tree), path, "exec"), mod.__dict__)
    return mod


class Harness:
    """the stand-ins, shared by the fixture generator and tests/test_reference_trainloop.py"""

    def __init__(self, n_train=6, n_val=4, bs=2, in_shp=8, latent=5):
        self.log = []
        self.bs, self.in_shp, self.latent = bs, in_shp, latent
        self.saved = []
        self.images = []
        self.it_train = self._iterator("train", n_train)
        self.it_val = self._iterator("val", n_val)
        self.calls = {"train_fn": 0, "loss_fn": 0}

    def _iterator(self, name, N):
        h = self

        class It:
            def __init__(self):
                self.N, self.k = N, 0

            def next(self):
                x = np.full((h.bs, 1, h.in_shp, h.in_shp), self.k / 100.0, np.float32)
                y = np.full((h.bs, 3, h.in_shp, h.in_shp), -self.k / 100.0, np.float32)
                h.log.append(["draw", name, self.k])
                self.k += 1
                return x, y
            __next__ = next

            def __iter__(self):
                return self
        return It()

    def fn(self, name):
        def call(Z, X, Y):
            i = self.calls[name]
            self.calls[name] = i + 1
            self.log.append([name, list(np.shape(Z)), str(np.asarray(Z).dtype), round(float(np.asarray(X).flat[0]) * 100)])
            base = 10.0 if name == "train_fn" else 20.0
            return [np.float32(base + i + 0.125 * j) for j in range(5)]
        return call

    def sampler(self, n, d):
        self.log.append(["sampler", int(n), int(d)])
        return np.zeros((n, d))

    def gen(self, name, channels):
        def call(X):
            self.log.append([name, list(np.shape(X))])
            return np.zeros((np.shape(X)[0], channels, self.in_shp, self.in_shp), np.float32)
        return call

    def attach(self, m, keys):
        m.train_keys = list(keys)
        m.train_fn, m.loss_fn = self.fn("train_fn"), self.fn("loss_fn")
        m.gen_fn, m.gen_fn_det = self.gen("gen_fn", 3), self.gen("gen_fn_det", 3)
        m.z_fn, m.z_fn_det = self.gen("z_fn", 1), self.gen("z_fn_det", 1)
        m.sampler, m.latent_dim, m.in_shp = self.sampler, self.latent, self.in_shp
        m.is_a_grayscale, m.is_b_grayscale = True, False
        m.train_mode, m.verbose = "both", False
        m.lr = types.SimpleNamespace(get_value=lambda: 0.0001)
        m.save_model = lambda filename: self.saved.append(os.path.basename(filename))
        m.load_model = lambda filename, mode="both": self.log.append(["load_model", os.path.basename(filename)])

    def run(self, m, out_dir, model_dir, **kw):
        m.train(self.it_train, self.it_val, batch_size=self.bs, num_epochs=3, out_dir=out_dir, model_dir=model_dir,
                save_every=2, **kw)
        rows = open(os.path.join(out_dir, "results.txt")).read().strip().split("\n")
        header = rows[0].split(",")
        t = header.index("time")
        body = [[("<t>" if i == t else v) for i, v in enumerate(r.split(","))] for r in rows[1:]]
        files = sorted(os.path.relpath(os.path.join(d, f), out_dir) for d, _, fs in os.walk(out_dir) for f in fs)
        return {"header": header, "rows": body, "files": files, "saved": list(self.saved), "log": self.log}


PLATEAU_CASES = [dict(), dict(mode='min', patience=2, factor=0.5), dict(mode='max', patience=1, cooldown=2),
                 dict(mode='min', patience=0, factor=0.1, min_lr=1e-3, epsilon=0.01)]


def plateau_sequence(seed):
    rng = np.random.RandomState(seed)
    return np.round(1.5 + 0.2 * np.sin(np.arange(60) / 3.0) + 0.05 * rng.randn(60), 3).tolist()


def main():
    import make_reference_graph as G
    import make_reference_iterator as I
    G.install_shims({})
    ref_util = I.load_reference_util()
    sys.modules["util"] = ref_util
    sys.modules["keras_ports"] = types.ModuleType("keras_ports")
    sys.modules["keras_ports"].ReduceLROnPlateau = object
    h = Harness()
    skimage = types.ModuleType("skimage")
    skimage.io = types.ModuleType("skimage.io")
    skimage.io.imsave = lambda fname=None, arr=None, **k: (open(fname, "wb").close(), None)[1]
    sys.modules.update({"skimage": skimage, "skimage.io": skimage.io})
    sys.path.insert(0, REF)
    P = load_py2(os.path.join(REF, "pix2pix.py"), "reference_pix2pix")
    m = P.Pix2Pix.__new__(P.Pix2Pix)
    h.attach(m, ['dcgan_gen', 'dcgan_disc', 'p2p_gen', 'p2p_recon', 'p2p_disc'])      # pix2pix.py:157
    with tempfile.TemporaryDirectory() as td:
        out = h.run(m, os.path.join(td, "out"), os.path.join(td, "models"))
    # keras_ports.ReduceLROnPlateau (keras_ports.py:7-111), executed: learning-rate trajectories on fixed sequences
    np.Inf = np.inf                                    # the reference predates NumPy 2.0
    KP = load_py2(os.path.join(REF, "keras_ports.py"), "reference_keras_ports")
    from gan_heightmaps_amd.updates import shared
    out["plateau"] = []
    for kw in PLATEAU_CASES:
        lr = shared(np.float32(0.01))
        cb = KP.ReduceLROnPlateau(lr, **kw)
        cb.on_train_begin()
        traj = []
        for e, v in enumerate(plateau_sequence(len(kw))):
            cb.on_epoch_end(v, e + 1)
            traj.append([float(lr.get_value()), int(cb.wait), int(cb.cooldown_counter)])
        out["plateau"].append({"kw": kw, "trajectory": traj})
    out["provenance"] = "python tests/golden/make_reference_trainloop.py (reference pix2pix.py executed via lib2to3, not copied)"
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(HERE, "reference_trainloop.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out["log"]), "events", out["files"][:6], out["saved"])
    for r in out["rows"]:
        print(",".join(r))


if __name__ == "__main__":
    main()
