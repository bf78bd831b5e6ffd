"""Restatement of the data path that FEEDS the hot path (TEST INFRASTRUCTURE ONLY):
/root/reference/util.py:20-42 (`iterate_hdf5`) on top of Keras' ImageDataGenerator as configured at
/root/reference/experiments.py:13 -- horizontal_flip, vertical_flip, rotation_range=360, fill_mode="reflect".

PARITY STATUS: unpinned.  Keras is neither vendored nor pinned by the reference; what follows is the behaviour of
Keras 2.0.x (the 2017 release line the notebooks' "Using Theano backend." banner belongs to) for channels-first
arrays:  ImageDataGenerator.random_transform draws theta ~ U(-rot, rot) degrees, builds the rotation about the
image centre (transform_matrix_offset_center), resamples every channel with
scipy.ndimage.affine_transform(order=0, mode=fill_mode), then flips columns / rows each with probability 1/2
(np.random.random() < 0.5, columns first).  NumpyArrayIterator.next() reseeds the GLOBAL numpy RNG with
``seed`` (+ batches seen = 0 for a fresh flow), permutes the batch (shuffle=True), then transforms sample by
sample.  The reference calls flow(...).next() once per array with the SAME seed for A and B, so both get the same
permutation and the same transforms (util.py:38-40).
"""
import numpy as np
import scipy.ndimage as ndi


def get_slices(length, bs):
    """util._get_slices (util.py:10-18)"""
    return [slice(b * bs, (b + 1) * bs) for b in range((length + bs - 1) // bs)]


def normalise(arr_nhwc_u8, is_grayscale):
    """util.py:28-35: NHWC uint8 -> NCHW float32; /255 if grayscale else (x-127.5)/127.5"""
    x = arr_nhwc_u8.astype("float32").swapaxes(3, 2).swapaxes(2, 1)
    return (x / 255.0) if is_grayscale else (x - 127.5) / 127.5


def rotation_matrix_centered(theta, h, w):
    """Keras: rotation_matrix then transform_matrix_offset_center(matrix, h, w) (3x3, output -> input index map)"""
    rot = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
    o_x, o_y = float(h) / 2 + 0.5, float(w) / 2 + 0.5
    off = np.array([[1, 0, o_x], [0, 1, o_y], [0, 0, 1]])
    reset = np.array([[1, 0, -o_x], [0, 1, -o_y], [0, 0, 1]])
    return off @ rot @ reset


def draw_transform(rng, rotation_range=360.0, horizontal_flip=True, vertical_flip=True):
    """RNG draws of ImageDataGenerator.random_transform in Keras' order (only the enabled options draw)."""
    theta = np.pi / 180 * rng.uniform(-rotation_range, rotation_range) if rotation_range else 0.0
    hflip = bool(rng.random_sample() < 0.5) if horizontal_flip else False
    vflip = bool(rng.random_sample() < 0.5) if vertical_flip else False
    return theta, hflip, vflip


def apply_transform(x_chw, theta, hflip, vflip, order=0, mode='reflect'):
    c, h, w = x_chw.shape
    m = rotation_matrix_centered(theta, h, w)
    out = np.stack([ndi.affine_transform(ch, m[:2, :2], m[:2, 2], order=order, mode=mode, cval=0.0) for ch in x_chw])
    if hflip:
        out = out[:, :, ::-1]
    if vflip:
        out = out[:, ::-1, :]
    return np.ascontiguousarray(out)


def flow_first_batch(x_nchw, seed, augment=True):
    """imgen.flow(x, None, batch_size=len(x), seed=seed).next(): -> (augmented batch, permutation, params)"""
    rng = np.random.RandomState(seed)                 # == np.random.seed(seed) on the global RNG
    n = x_nchw.shape[0]
    perm = rng.permutation(n)
    out = np.empty_like(x_nchw)
    params = []
    for i, j in enumerate(perm):
        if augment:
            theta, hf, vf = draw_transform(rng)
            out[i] = apply_transform(x_nchw[j], theta, hf, vf)
        else:
            theta, hf, vf = 0.0, False, False
            out[i] = x_nchw[j]
        params.append((theta, hf, vf))
    return out, perm, params


class Hdf5IteratorOracle:
    """util.Hdf5Iterator / iterate_hdf5 with numpy arrays standing in for the h5py datasets."""

    def __init__(self, X, Y, bs, is_a_grayscale, is_b_grayscale, augment=True):
        assert X.shape[0] == Y.shape[0]
        self.X, self.Y, self.bs, self.N = X, Y, bs, X.shape[0]
        self.ga, self.gb, self.augment = is_a_grayscale, is_b_grayscale, augment
        self.rnd_state = np.random.RandomState(0)     # util.py:21 default argument
        self._pending = []

    def next(self):
        if not self._pending:
            self._pending = get_slices(self.N, self.bs)
            self.rnd_state.shuffle(self._pending)      # util.py:24-26
        sl = self._pending.pop(0)
        x, y = normalise(self.X[sl], self.ga), normalise(self.Y[sl], self.gb)
        # experiments.get_iterators always passes an ImageDataGenerator (an identity one when da=False), so the
        # seed is always drawn and the batch is always permuted by flow(shuffle=True) (util.py:37-40)
        seed = self.rnd_state.randint(0, 100000)
        x = flow_first_batch(x, seed, self.augment)[0]
        y = flow_first_batch(y, seed, self.augment)[0]
        return x, y

    __next__ = next
