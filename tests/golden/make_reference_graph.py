#!/usr/bin/env python
"""Runs the REFERENCE's own files, unmodified, against this package's layer vocabulary and records what they
build -- the evidence that `test1_nobn_bilin_both` "runs unchanged" at the graph-construction level.

    python tests/golden/make_reference_graph.py      # in the build container only: needs /root/reference
    -> tests/golden/reference_graph.json

What is executed from /root/reference (read-only, nothing is copied):
  * experiments.py, as `__main__` with argv = [test1_nobn_bilin_both, train]  (experiments.py:98-131)
  * architectures/dcgan.py and architectures/p2p.py, imported by it               (its own `from architectures import`)
against stand-in modules for what this image lacks:
  theano / lasagne.*  -> gan_heightmaps_amd.{layers, nonlinearities, init, updates}   (the drop-in vocabulary)
  keras / h5py / util -> inert stubs (the iterators are not exercised here; tests/test_data_path.py does that)
  pix2pix             -> a recorder class: the reference's pix2pix.py is Python-2 only (print statements)
Python 2 -> 3: the architecture files compute filter counts with `/` (dcgan.py:19,39); the stand-in layer
constructors accept an integral float there, which is the only accommodation made.

The fixture holds, per network, the ordered layer list (class, output shape, parameter names and shapes) and the
keyword arguments experiments.py passed to Pix2Pix.  tests/test_reference_graph.py rebuilds the same networks
from this package's re-typed architectures / experiments and compares.
"""
import json
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np                                   # noqa: E402
from gan_heightmaps_amd import layers as L          # noqa: E402
from gan_heightmaps_amd import nonlinearities as NL  # noqa: E402
from gan_heightmaps_amd import init as INIT          # noqa: E402
from gan_heightmaps_amd import updates as UP         # noqa: E402


def _intify(v):
    return int(v) if isinstance(v, float) and float(v).is_integer() else v


def _py2_division(ctor):
    """layer constructor that takes 256.0 where Python 2's `512/2` gave 256"""
    def make(*a, **k):
        return ctor(*[_intify(x) for x in a], **{n: _intify(x) for n, x in k.items()})
    make.__name__ = getattr(ctor, "__name__", "layer")
    return make


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims(recorder):
    theano = _module("theano", shared=UP.shared)
    theano.tensor = _module("theano.tensor")
    public = lambda mod: {k: v for k, v in vars(mod).items() if not k.startswith("_")}
    layer_names = public(L)
    bases = tuple(b for b in (getattr(L, 'Layer', None), getattr(L, 'MergeLayer', None)) if b is not None)
    wrapped = {k: (_py2_division(v) if isinstance(v, type) and issubclass(v, L.Layer) and v not in bases else v)
               for k, v in layer_names.items()}      # the base classes stay classes: layers.py:13 subclasses Layer
    lasagne = _module("lasagne")
    lasagne.layers = _module("lasagne.layers", **wrapped)
    lasagne.nonlinearities = _module("lasagne.nonlinearities", theano=theano, **public(NL))
    lasagne.init = _module("lasagne.init", **public(INIT))
    lasagne.updates = _module("lasagne.updates", **public(UP))
    lasagne.objectives = _module("lasagne.objectives")
    lasagne.utils = _module("lasagne.utils", floatX=INIT.floatX)

    class ImageDataGenerator:
        def __init__(self, **kw):
            self.kw = kw
    keras = _module("keras")
    keras.preprocessing = _module("keras.preprocessing")
    keras.preprocessing.image = _module("keras.preprocessing.image", ImageDataGenerator=ImageDataGenerator)

    class _H5(dict):
        def __init__(self, path, mode):
            super().__init__(xt="xt", yt="yt", xv="xv", yv="yv")
            recorder["dataset"] = path
    _module("h5py", File=_H5)

    class Hdf5Iterator:
        def __init__(self, X, Y, bs, imgen, is_a_grayscale, is_b_grayscale):
            recorder.setdefault("iterators", []).append(
                dict(X=X, Y=Y, bs=bs, imgen=imgen.kw, is_a_grayscale=is_a_grayscale, is_b_grayscale=is_b_grayscale))
    _module("util", Hdf5Iterator=Hdf5Iterator)

    class Pix2Pix:
        def __init__(self, **kw):
            recorder["pix2pix_kwargs"] = kw

        def train(self, it_train, it_val, **kw):
            recorder["train_kwargs"] = kw
    _module("pix2pix", Pix2Pix=Pix2Pix)


# name -> (module, function, positional args, keyword args); nonlinearities by name
VARIANTS = {
    "dcgan_gen_bilinear": ("dcgan", "default_generator", (100, False), dict(bilinear_upsample=True)),
    "dcgan_gen_repeats": ("dcgan", "default_generator", (64, True), dict(num_repeats=1, nch=64, final_size=64,
                                                                          div=[2, 2, 4, 4])),
    "dcgan_gen_dropout": ("dcgan", "default_generator", (64, True), dict(dropout_p=0.25, nch=64, final_size=32,
                                                                          div=[2, 2, 4])),
    "dcgan_disc_bn_avg": ("dcgan", "default_discriminator", (512, False), dict(bn=True, pool_mode="average",
                                                                               nonlinearity="sigmoid")),
    "dcgan_disc_repeats": ("dcgan", "default_discriminator", (512, True), dict(num_repeats=2, nonlinearity="linear")),
    "unet_deconv": ("p2p", "g_unet", (512, True, False), dict(nf=32, act="tanh", bilinear_upsample=False)),
    "unet_repeats_dropout": ("p2p", "g_unet", (512, False, True), dict(nf=16, act="sigmoid", dropout=True,
                                                                       num_repeats=1, bilinear_upsample=True)),
    "unet_256": ("p2p", "g_unet_256", (256, True, False), dict(nf=32, act="tanh", dropout=0.5)),
    "patchgan_bn": ("p2p", "discriminator", (512, True, False), dict(nf=32, act="sigmoid", bn=True, num_repeats=1)),
    "patchgan2": ("p2p", "discriminator2", (512, False, False), dict(nf=16, act="linear", mul_factor=[1, 2, 4],
                                                                     num_repeats=1)),
    "fake_generator": ("p2p", "fake_generator", (True, False), dict(act="tanh")),
    "fake_discriminator": ("p2p", "fake_discriminator", (True, False), dict()),
}


def resolve(kws, NL):
    names = {"linear": NL.linear, "tanh": NL.tanh, "sigmoid": NL.sigmoid}
    return {k: (names[v] if k in ("act", "nonlinearity") and v in names else v) for k, v in kws.items()}


def describe(out_layer):
    rows = []
    for l in L.get_all_layers(out_layer):
        nl = getattr(l, "nonlinearity", None)
        rows.append({
            "class": type(l).__name__,
            "output_shape": [None if s is None else int(s) for s in l.output_shape],
            "params": [[p.name, list(map(int, p.shape)), sorted(p.tags)] for p in l.get_params()],
            "nonlinearity": None if nl is None else getattr(nl, "__name__", type(nl).__name__) +
            ("(%g)" % nl.leakiness if hasattr(nl, "leakiness") else ""),
        })
    return rows


def jsonable(v):
    if callable(v):
        return "<%s.%s>" % (getattr(v, "__module__", "?"), getattr(v, "__name__", type(v).__name__))
    if hasattr(v, "get_value"):
        return {"shared": float(v.get_value())}
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    return v


def main():
    rec = {}
    install_shims(rec)
    sys.path.insert(0, REF)                          # `from architectures import p2p, dcgan`, `from layers import ...`
    sys.path.insert(0, os.path.join(REF, "architectures"))
    argv = sys.argv
    sys.argv = ["experiments.py", "test1_nobn_bilin_both", "train"]
    try:
        runpy.run_path(os.path.join(REF, "experiments.py"), run_name="__main__")
    finally:
        sys.argv = argv
    from gan_heightmaps_amd import nonlinearities as NL
    kw = rec["pix2pix_kwargs"]
    INIT.set_rng(np.random.RandomState(0))
    nets = {
        "dcgan_gen": kw["gen_fn_dcgan"](kw["latent_dim"], kw["is_a_grayscale"], **kw["gen_params_dcgan"]),
        "dcgan_disc": kw["disc_fn_dcgan"](kw["in_shp"], kw["is_a_grayscale"], **kw["disc_params_dcgan"]),
        "p2p_gen": kw["gen_fn_p2p"](kw["in_shp"], kw["is_a_grayscale"], kw["is_b_grayscale"], **kw["gen_params_p2p"]),
    }
    pd = kw["disc_fn_p2p"](kw["in_shp"], kw["is_a_grayscale"], kw["is_b_grayscale"], **kw["disc_params_p2p"])
    nets["p2p_disc"] = pd["out"]
    # every architecture function of the reference, over the keyword variants it exposes
    import importlib
    ref_dcgan, ref_p2p = importlib.import_module("architectures.dcgan"), importlib.import_module("architectures.p2p")
    variants = {}
    for name, (mod, fn, args, kws) in VARIANTS.items():
        INIT.set_rng(np.random.RandomState(0))
        res = getattr(ref_dcgan if mod == "dcgan" else ref_p2p, fn)(*args, **resolve(kws, NL))
        variants[name] = describe(res["out"] if isinstance(res, dict) else res)
    out = {
        "variants": variants,
        "provenance": "python tests/golden/make_reference_graph.py  (reference files executed, not copied)",
        "experiment": "test1_nobn_bilin_both",
        "pix2pix_kwargs": jsonable(kw),
        "train_kwargs": jsonable(rec["train_kwargs"]),
        "dataset": rec["dataset"],
        "iterators": jsonable(rec["iterators"]),
        "p2p_disc_inputs": [list(l.shape) for l in pd["inputs"]],
        "networks": {k: describe(v) for k, v in nets.items()},
        "param_counts": {k: int(L.count_params(v)) for k, v in nets.items()},
    }
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(ROOT, "tests", "golden", "reference_graph.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out["networks"].items()}, out["param_counts"])


if __name__ == "__main__":
    main()
