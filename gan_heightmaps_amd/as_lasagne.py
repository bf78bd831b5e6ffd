"""Run the reference's own scripts unchanged on this backend.

    import gan_heightmaps_amd.as_lasagne as shim; shim.install()
    # from here on `import theano`, `from lasagne.layers import *`, `from pix2pix import Pix2Pix`,
    # `from util import Hdf5Iterator`, `from keras.preprocessing.image import ImageDataGenerator` ... resolve to
    # this package, so /path/to/reference/experiments.py and architectures/{dcgan,p2p}.py run as they are
    # (Python 3: the reference's pix2pix.py itself is Python-2 source and is the module that is replaced).

or from a shell, at the reference's root:   python -m gan_heightmaps_amd.as_lasagne experiments.py test1_nobn_bilin_both train

Only names are aliased; nothing of Theano/Lasagne/Keras is emulated beyond the vocabulary the reference's
experiments / architecture files use.  `h5py` is NOT aliased: a real h5py is what
`experiments.get_iterators` needs (datasets that slice like arrays).
tests/test_reference_graph.py executes the reference's files through the same aliases and checks every layer.
"""
import runpy
import sys
import types


def _public(mod):
    return {k: v for k, v in vars(mod).items() if not k.startswith("_")}


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    from . import data, init, layers, nonlinearities, pix2pix, updates, util
    from .architectures import layers as arch_layers
    theano = _module("theano", shared=updates.shared)
    theano.tensor = _module("theano.tensor")
    lasagne = _module("lasagne")
    lasagne.layers = _module("lasagne.layers", **_public(layers))
    # lasagne.nonlinearities binds the name `theano` too; experiments.py:117 relies on that star-import leak
    lasagne.nonlinearities = _module("lasagne.nonlinearities", theano=theano, **_public(nonlinearities))
    lasagne.init = _module("lasagne.init", **_public(init))
    lasagne.updates = _module("lasagne.updates", **_public(updates))
    lasagne.objectives = _module("lasagne.objectives")
    lasagne.utils = _module("lasagne.utils", floatX=init.floatX)
    keras = _module("keras")
    keras.preprocessing = _module("keras.preprocessing")
    keras.preprocessing.image = _module("keras.preprocessing.image", ImageDataGenerator=data.ImageDataGenerator)
    sys.modules["pix2pix"] = pix2pix
    sys.modules["util"] = util
    sys.modules["layers"] = arch_layers          # architectures/p2p.py:12 `from layers import BilinearUpsample2DLayer`
    return sys.modules


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    install()
    sys.argv = argv[1:]
    sys.path.insert(0, "")
    runpy.run_path(argv[1], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
