"""The data path in front of the train step (/root/reference/util.py:10-62, experiments.py:10-18) on the MI355X
backend: `Hdf5Iterator` with the reference's constructor and `.N` / `.next()` protocol, and the Keras
`ImageDataGenerator` options the reference uses (horizontal_flip, vertical_flip, rotation_range, fill_mode='reflect').

A batch never exists as fp32 on the host: the uint8 NHWC slice (4x smaller) is uploaded and one kernel
(`ghm_image_batch`) normalises, transposes to NCHW and resamples it through the per-sample affine map.  The random
draws follow Keras 2.0.x (`oracle/keras_aug.py` restates and documents them; parity unpinned because the reference
pins no Keras version): per batch the slice order is shuffled with RandomState(0), a seed is drawn from it, and
with that seed -- for A and B alike -- the batch is permuted and each sample draws theta, then the column flip,
then the row flip.
"""
import ctypes as C

import numpy as np

from ._lib import call


class ImageDataGenerator:
    """keras.preprocessing.image.ImageDataGenerator, restricted to what experiments.py:13-15 configures."""

    def __init__(self, horizontal_flip=False, vertical_flip=False, rotation_range=0., fill_mode='nearest', **unsupported):
        if unsupported:
            raise NotImplementedError("ImageDataGenerator options without a kernel: %s" % sorted(unsupported))
        if rotation_range and fill_mode != 'reflect':
            raise NotImplementedError("rotation needs fill_mode='reflect' (the only border mode with a kernel)")
        self.horizontal_flip, self.vertical_flip = bool(horizontal_flip), bool(vertical_flip)
        self.rotation_range, self.fill_mode = float(rotation_range), fill_mode

    def draw(self, rng):
        """RNG draws of random_transform, in Keras' order; only enabled options draw."""
        theta = np.pi / 180 * rng.uniform(-self.rotation_range, self.rotation_range) if self.rotation_range else 0.0
        hflip = bool(rng.random_sample() < 0.5) if self.horizontal_flip else False
        vflip = bool(rng.random_sample() < 0.5) if self.vertical_flip else False
        return theta, hflip, vflip


def transform_row(theta, hflip, vflip, h, w):
    """{m00, m01, off0, m10, m11, off1, hflip, vflip} of ghm_image_batch for a rotation by theta about the image
    centre (Keras transform_matrix_offset_center: centre at (h/2 + 0.5, w/2 + 0.5))."""
    rot = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
    o_x, o_y = float(h) / 2 + 0.5, float(w) / 2 + 0.5
    m = np.array([[1, 0, o_x], [0, 1, o_y], [0, 0, 1]]) @ rot @ np.array([[1, 0, -o_x], [0, 1, -o_y], [0, 0, 1]])
    return [m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2], float(hflip), float(vflip)]


def plan_flow(imgen, n, seed, h, w):
    """imgen.flow(x, None, batch_size=n, seed=seed).next() as data: (permutation, [n, 8] transform table)."""
    rng = np.random.RandomState(seed)          # Keras reseeds the global numpy RNG with `seed`
    perm = rng.permutation(n)                  # NumpyArrayIterator(shuffle=True)
    rows = [transform_row(*imgen.draw(rng), h=h, w=w) for _ in range(n)]
    return perm, np.asarray(rows, np.float64).reshape(n, 8)


class Hdf5Iterator:
    """util.Hdf5Iterator: infinite generator of (A, B) batches from two array-likes of NHWC uint8 images."""

    def __init__(self, X, y, bs, imgen, is_a_grayscale, is_b_grayscale, is_uint8=True, device=None):
        assert X.shape[0] == y.shape[0]
        if not is_uint8:
            raise NotImplementedError("is_uint8=False (pre-normalised float datasets)")
        self.X, self.Y, self.bs, self.N = X, y, int(bs), int(X.shape[0])
        self.imgen = imgen if imgen is not None else None
        self.ga, self.gb = is_a_grayscale, is_b_grayscale
        self.rnd_state = np.random.RandomState(0)          # util.py:21
        self._pending = []
        self.dev = device
        self._bufs = {}

    def __iter__(self):
        return self

    # ---- host-side plan of one batch (pure; tested against oracle/keras_aug.py on CPU) ----
    def _refill(self):
        if not self._pending:
            self._pending = [slice(b * self.bs, (b + 1) * self.bs) for b in range((self.N + self.bs - 1) // self.bs)]
            self.rnd_state.shuffle(self._pending)          # util.py:24-26

    def peek_n(self):
        """number of samples the next batch will hold (the last slice of a pass may be ragged)"""
        self._refill()
        return len(range(*self._pending[0].indices(self.N)))

    def plan_next(self):
        self._refill()
        sl = self._pending.pop(0)
        n = len(range(*sl.indices(self.N)))
        h, w = self.X.shape[1:3]
        if self.imgen is not None:
            seed = self.rnd_state.randint(0, 100000)       # util.py:38
            perm, table = plan_flow(self.imgen, n, seed, h, w)
        else:
            perm = np.arange(n)
            table = np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0], np.float64), (n, 1))
        return sl, perm, table

    # ---- device side ----
    def _staging(self, name, nbytes, via=None):
        cur = self._bufs.get(name)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                # a copy or the augmentation kernel of the previous batch may still be reading it on the copy stream
                (via if via is not None else self.dev).sync()
                self.dev.free(cur[0])
            cur = (self.dev.alloc(nbytes), nbytes)
            self._bufs[name] = cur
        return cur[0]

    def next_into(self, x_dst, y_dst, via=None, pinned=None):
        """produce the next batch directly into two device tensors [n, C, H, W] (n = this slice's length).
        ``via`` / ``pinned``: issue the uploads and the augmentation kernel on that context (a copy stream) from page-locked
        staging kept in the dict ``pinned`` WITHOUT waiting for them (GanStep.produce_async: the batch of step i+1 is made
        while step i runs); default: on the destination's own context, synchronously."""
        if self.dev is None:
            self.dev = x_dst.dev
        sl, perm, table = self.plan_next()
        n = len(perm)
        dev = via if via is not None else None
        for name, arr, dst, gray in (("a", self.X, x_dst, self.ga), ("b", self.Y, y_dst, self.gb)):
            batch = np.ascontiguousarray(np.asarray(arr[sl])[perm])          # uint8 NHWC, permuted like flow()
            _, h, w, c = batch.shape
            assert dst.shape[1:] == (c, h, w) and dst.shape[0] >= n, (dst.shape, batch.shape)
            src = self._staging(name, batch.nbytes, via)
            xf = self._staging(name + "_xf", table.nbytes, via)
            d = dev if dev is not None else dst.dev
            if dev is None:
                d.h2d(src, batch)
                d.h2d(xf, table)
            else:
                from .device import PinnedArray
                for key, a, ptr in ((name, batch, src), (name + "_xf", np.ascontiguousarray(table), xf)):
                    pa = pinned.get(key)
                    if pa is None or pa.array.shape != a.shape or pa.array.dtype != a.dtype:
                        if pa is not None:
                            pa.close()
                        pa = pinned[key] = PinnedArray(a.shape, a.dtype)
                    np.copyto(pa.array, a)
                    d.h2d_async(ptr, pa)
            call("ghm_image_batch", d.h, C.c_void_p(src), n, h, w, c, C.c_void_p(xf), 0 if gray else 1,
                 C.c_void_p(dst.ptr), dst.nstride)
        return n

    def next(self):
        """reference protocol: -> (A [n, Ca, H, W], B [n, Cb, H, W]) float32 numpy arrays"""
        if self.dev is None:
            from .device import Device
            self.dev = Device(0)
        n = min(self.bs, self.N)
        h, w, ca = self.X.shape[1:4]
        cb = self.Y.shape[3]
        xa = self._tensor("out_a", (self.bs, ca, h, w))
        xb = self._tensor("out_b", (self.bs, cb, h, w))
        n = self.next_into(xa, xb)
        return xa.samples(0, n).numpy(), xb.samples(0, n).numpy()

    __next__ = next

    def _tensor(self, name, shape):
        t = self._bufs.get(name)
        if t is None or t.shape != tuple(shape):
            t = self.dev.empty(shape)
            self._bufs[name] = t
        return t
