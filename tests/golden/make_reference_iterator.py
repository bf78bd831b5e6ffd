#!/usr/bin/env python
"""Runs the REFERENCE's util.iterate_hdf5 (util.py:10-44), unmodified, on small uint8 arrays and records what it
yields -- golden vectors for the part of the data path that is the reference's own code: slice order
(RandomState(0) shuffle per pass), NHWC->NCHW, the two normalisation branches, and the flow seeds it draws.

    python tests/golden/make_reference_iterator.py        # build container only: needs /root/reference
    -> tests/golden/reference_iterator.npz

Stand-ins: `cPickle` -> pickle (Python 3), `keras.preprocessing.image.ImageDataGenerator` -> a recorder whose
`flow(x, None, batch_size, seed)` returns x unchanged and logs the seed (Keras itself is absent: what flow()
does to a batch is third-party behaviour restated in oracle/keras_aug.py, not reference code).
"""
import os
import pickle
import sys
import types
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"


class RecordingGenerator:
    def __init__(self, **kw):
        self.kw = kw
        self.seeds = []

    def flow(self, x, y=None, batch_size=None, seed=None):
        self.seeds.append(int(seed))
        out = types.SimpleNamespace()
        out.next = lambda: x
        return out


def load_reference_util():
    sys.modules.setdefault("cPickle", pickle)
    keras = types.ModuleType("keras")
    keras.preprocessing = types.ModuleType("keras.preprocessing")
    keras.preprocessing.image = types.ModuleType("keras.preprocessing.image")
    keras.preprocessing.image.ImageDataGenerator = RecordingGenerator
    sys.modules.update({"keras": keras, "keras.preprocessing": keras.preprocessing,
                        "keras.preprocessing.image": keras.preprocessing.image})
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_util", os.path.join(REF, "util.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CASES = {
    # name: (N, H, W, bs, is_a_grayscale, is_b_grayscale, with generator, steps)
    "gray_colour_gen": (10, 6, 5, 4, True, False, True, 8),       # experiments.py:16-17 configuration, ragged tail
    "colour_gray_gen": (7, 4, 4, 3, False, True, True, 7),
    "no_generator": (9, 5, 6, 2, True, False, False, 11),
}


def main():
    util = load_reference_util()
    out = {}
    for name, (N, H, W, bs, ga, gb, gen, steps) in CASES.items():
        rng = np.random.RandomState(zlib.crc32(name.encode()) % 1000)
        X = rng.randint(0, 256, (N, H, W, 1 if ga else 3)).astype(np.uint8)
        Y = rng.randint(0, 256, (N, H, W, 1 if gb else 3)).astype(np.uint8)
        g = RecordingGenerator() if gen else None
        it = util.iterate_hdf5(g, ga, gb, True)(X, Y, bs)
        out[name + "/X"], out[name + "/Y"] = X, Y
        for s in range(steps):
            a, b = next(it)
            out["%s/a%d" % (name, s)] = np.ascontiguousarray(a)
            out["%s/b%d" % (name, s)] = np.ascontiguousarray(b)
        out[name + "/seeds"] = np.array(g.seeds if gen else [], np.int64)
        out[name + "/cfg"] = np.array([N, H, W, bs, int(ga), int(gb), int(gen), steps], np.int64)
    # util.convert_to_rgb / compose_imgs (util.py:69-99): the image helpers behind generate_* 
    rng = np.random.RandomState(5)
    g = (rng.rand(1, 6, 7) * 1.6 - 0.3).astype(np.float32)          # grayscale, partly outside [0, 1]
    c = (rng.rand(3, 6, 7) * 2.6 - 1.3).astype(np.float32)          # tanh range, partly outside [-1, 1]
    out["rgb/g"], out["rgb/c"] = g, c
    out["rgb/g_gray"] = util.convert_to_rgb(g.copy(), is_grayscale=True)
    out["rgb/g_tanh"] = util.convert_to_rgb(g.copy(), is_grayscale=False)
    out["rgb/c_tanh"] = util.convert_to_rgb(c.copy(), is_grayscale=False)
    out["rgb/c_gray"] = util.convert_to_rgb(c.copy(), is_grayscale=True)
    out["rgb/compose"] = util.compose_imgs(g.copy(), c.copy(), is_a_grayscale=True, is_b_grayscale=False)
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(ROOT, "tests", "golden", "reference_iterator.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
