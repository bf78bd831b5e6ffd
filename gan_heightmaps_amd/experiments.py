"""Experiment registry with the reference's entry points (/root/reference/experiments.py:22-131):
``python experiments.py <experiment> <mode>``, experiments test1_nobn, test1_nobn_finetunep2p_bilin and
test1_nobn_bilin_both, modes train / interp / gen.

``get_iterators`` returns the device-side counterpart of util.Hdf5Iterator (gan_heightmaps_amd.data) over the
xt/yt/xv/yv uint8 NHWC arrays of an HDF5 file (h5py when importable, else this package's own reader h5lite.py) or of
an ``.npz`` with the same four keys.  A missing or unreadable dataset is an error; synthetic batches with the
reference's value ranges are served only when asked for (``dataset=None`` or ``GHM_DATASET=synthetic``).
"""
import os
import sys

import numpy as np

from .architectures import dcgan, p2p
from .nonlinearities import linear, tanh
from .pix2pix import Pix2Pix
from .updates import rmsprop, shared
from .init import floatX


class ArrayIterator:
    """Infinite shuffled batch generator over uint8 NHWC arrays (util.py:20-42 without augmentation):
    NHWC uint8 -> NCHW float32; A: /255 if grayscale else (x-127.5)/127.5; same for B."""

    def __init__(self, X, Y, bs, is_a_grayscale, is_b_grayscale, seed=0):
        assert X.shape[0] == Y.shape[0]
        self.X, self.Y, self.bs, self.N = X, Y, bs, X.shape[0]
        self.ga, self.gb = is_a_grayscale, is_b_grayscale
        self.rng = np.random.RandomState(seed)
        self._slices = []

    def __iter__(self):
        return self

    def __next__(self):
        if not self._slices:
            self._slices = [slice(b * self.bs, (b + 1) * self.bs) for b in range((self.N + self.bs - 1) // self.bs)]
            self.rng.shuffle(self._slices)
        sl = self._slices.pop(0)
        x = self.X[sl].astype("float32").transpose(0, 3, 1, 2)
        y = self.Y[sl].astype("float32").transpose(0, 3, 1, 2)
        x = x / 255.0 if self.ga else (x - 127.5) / 127.5
        y = y / 255.0 if self.gb else (y - 127.5) / 127.5
        return np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)

    next = __next__


def synthetic_arrays(n, in_shp, is_a_grayscale, is_b_grayscale, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.randint(0, 256, (n, in_shp, in_shp, 1 if is_a_grayscale else 3)).astype(np.uint8)
    Y = rng.randint(0, 256, (n, in_shp, in_shp, 1 if is_b_grayscale else 3)).astype(np.uint8)
    return X, Y


def open_dataset(dataset):
    """``h5py.File(dataset, "r")`` of experiments.py:11: a mapping with the uint8 NHWC arrays xt / yt / xv / yv that
    stay on disk and are sliced per batch.  ``.npz`` files are accepted too.  HDF5 files are opened with h5py when it is
    importable and with this package's own reader (h5lite.py: contiguous / chunked / gzip datasets) otherwise."""
    if not os.path.exists(dataset):
        raise FileNotFoundError("dataset %r does not exist (set GHM_DATASET to an .h5 / .npz file with xt, yt, xv, yv, "
                                "or to 'synthetic' for random batches)" % (dataset,))
    if dataset.endswith(".npz"):
        d = np.load(dataset)
    else:
        try:
            import h5py
            d = h5py.File(dataset, "r")
        except ImportError:
            from . import h5lite
            d = h5lite.File(dataset, "r")
    missing = [k for k in ("xt", "yt", "xv", "yv") if k not in d]
    if missing:
        raise KeyError("dataset %r has no array(s) %s (expected xt, yt, xv, yv as written by the reference's "
                       "notebooks/prototype_cropping_code.ipynb)" % (dataset, ", ".join(missing)))
    return d


def get_iterators(dataset, batch_size, is_a_grayscale, is_b_grayscale, da=True, in_shp=512, n_synthetic=8, device=None):
    """experiments.get_iterators (experiments.py:10-18): (it_train, it_val) over the xt/yt/xv/yv arrays of the
    dataset, augmented with flips + 360-degree rotation + reflect fill when ``da``.  ``dataset``: an HDF5 file as the
    reference opens it (``open_dataset``), an ``.npz`` with the same four keys, or 'synthetic' / None."""
    from .data import Hdf5Iterator, ImageDataGenerator
    if dataset is None or dataset == "synthetic":
        xt, yt = synthetic_arrays(n_synthetic, in_shp, is_a_grayscale, is_b_grayscale, 0)
        xv, yv = xt, yt
    else:
        d = open_dataset(dataset)
        xt, yt, xv, yv = d['xt'], d['yt'], d['xv'], d['yv']
    if da:
        imgen = ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
    else:
        imgen = ImageDataGenerator()
    it_train = Hdf5Iterator(xt, yt, batch_size, imgen, is_a_grayscale=is_a_grayscale, is_b_grayscale=is_b_grayscale,
                            device=device)
    it_val = Hdf5Iterator(xv, yv, batch_size, imgen, is_a_grayscale=is_a_grayscale, is_b_grayscale=is_b_grayscale,
                          device=device)
    return it_train, it_val


# kwargs of the three experiments (experiments.py:24-41, :63-79, :102-119)
_COMMON = dict(
    gen_fn_dcgan=dcgan.default_generator, disc_fn_dcgan=dcgan.default_discriminator,
    gen_params_dcgan={'num_repeats': 0, 'div': [2, 2, 4, 4, 8, 8, 8]},
    disc_params_dcgan={'num_repeats': 0, 'bn': False, 'nonlinearity': linear, 'div': [8, 4, 4, 4, 2, 2, 2]},
    gen_fn_p2p=p2p.g_unet, disc_fn_p2p=p2p.discriminator,
    disc_params_p2p={'nf': 64, 'bn': False, 'num_repeats': 0, 'act': linear, 'mul_factor': [1, 2, 4, 8]},
    in_shp=512, latent_dim=1000, is_a_grayscale=True, is_b_grayscale=False, lsgan=True, opt=rmsprop)


def experiment_kwargs(name):
    """The keyword arguments experiments.py passes to Pix2Pix for experiment ``name`` (checked against the
    reference's own file, executed, in tests/test_reference_graph.py)."""
    kw = dict(_COMMON)
    kw['opt_args'] = {'learning_rate': shared(floatX(1e-4))}
    if name == 'test1_nobn':
        kw['gen_params_p2p'] = {'nf': 64, 'act': tanh, 'num_repeats': 0}
    elif name == 'test1_nobn_finetunep2p_bilin':
        kw['gen_params_p2p'] = {'nf': 64, 'act': tanh, 'num_repeats': 0, 'bilinear_upsample': True}
        kw['train_mode'] = 'p2p'
    elif name == 'test1_nobn_bilin_both':
        kw['gen_params_p2p'] = {'nf': 64, 'act': tanh, 'num_repeats': 0, 'bilinear_upsample': True}
        kw['train_mode'] = 'both'
    else:
        raise KeyError(name)
    return kw


def make_model(name, **backend):
    """Construct the Pix2Pix of experiment ``name``; ``backend`` = device / comm / use_graph / seed / verbose."""
    kw = experiment_kwargs(name)
    kw.update(backend)
    return Pix2Pix(**kw)


DATASET = os.environ.get("GHM_DATASET", "/data/lisa/data/cbeckham/textures_v2_brown500.h5")


BASE_RUN = "test1_repeatnod_fixp2p_nobn"             # the run whose checkpoints the other entry points load
FINETUNE_RUN = "test1_repeatnod_fixp2p_nobn_finetunep2p_bilin"


def _train(model, out_name, num_epochs, dataset=None, **train_kw):
    bs = 4
    it_train, it_val = get_iterators(DATASET if dataset is None else dataset, bs, True, False, True,
                                     device=model.device)
    model.train(it_train, it_val, batch_size=bs, num_epochs=num_epochs, out_dir="output/%s" % out_name,
                model_dir="models/%s" % out_name, **train_kw)


def test1_nobn(mode, num_epochs=1000, models_dir="models", **kw):
    """experiments.py:22-55"""
    assert mode in ["train", "interp", "gen"]
    model = make_model('test1_nobn', **kw.pop('backend', {}))
    if mode == "train":
        _train(model, BASE_RUN, num_epochs, **kw)
    elif mode == "interp":
        model.load_model("%s/%s/600.model.bak" % (models_dir, BASE_RUN))
        zs = model.sampler(2, model.latent_dim)
        # the reference passes (z1, z2, "/tmp/test.png") against the signature (out_name, zsample1, zsample2)
        # (experiments.py:51 vs pix2pix.py:328): the evident intent is kept
        model.generate_interpolation("/tmp/test.png", floatX(zs[0]), floatX(zs[1]), mode='matrix')
    else:
        model.load_model("%s/%s/600.model.bak" % (models_dir, BASE_RUN))
        model.generate_gz(100, 10, "deleteme")
    return model


def test1_nobn_finetunep2p_bilin(mode, num_epochs=1000, models_dir="models", load_base=True, **kw):
    """experiments.py:58-93: the DCGAN half comes from the base run's checkpoint, only the pix2pix half trains"""
    assert mode in ["train", "interp", "gen"]
    model = make_model('test1_nobn_finetunep2p_bilin', **kw.pop('backend', {}))
    if load_base:
        model.load_model("%s/%s/1000.model.bak" % (models_dir, BASE_RUN), mode='dcgan')
    if mode == "train":
        _train(model, FINETUNE_RUN, num_epochs, **kw)
    elif mode == "interp":
        model.load_model("%s/%s/1000.model.bak" % (models_dir, BASE_RUN), mode='dcgan')
        model.load_model("%s/%s/1000.model.bak" % (models_dir, FINETUNE_RUN), mode='p2p')
        model.generate_interpolation_clip(100, 4, "output/%s/interp_clip_600_concat_bothdet/" % FINETUNE_RUN,
                                          concat=True, deterministic=True)
    return model


def test1_nobn_bilin_both(mode, num_epochs=1000, **kw):
    """experiments.py:99-127 (only 'train' does anything in the reference)"""
    assert mode in ["train", "interp", "gen"]
    model = make_model('test1_nobn_bilin_both', **kw.pop('backend', {}))
    if mode == "train":
        _train(model, "test1_nobn_bilin_both_deleteme", num_epochs, **kw)
    return model


def main(argv):
    fn = {'test1_nobn': test1_nobn, 'test1_nobn_finetunep2p_bilin': test1_nobn_finetunep2p_bilin,
          'test1_nobn_bilin_both': test1_nobn_bilin_both}[argv[1]]
    fn(argv[2])


if __name__ == '__main__':
    main(sys.argv)
