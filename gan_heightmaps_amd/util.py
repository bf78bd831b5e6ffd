"""Host-side image helpers either side of the path (/root/reference/util.py:69-116) plus the iterator names
the reference exports from ``util`` (util.py:10-62 -> gan_heightmaps_amd.data).

Everything here is numpy on what the device has already produced; PNGs are written with PIL (the
reference's skimage is not in this image): a float image in [0,1] becomes uint8 by round(255*x), which is
what skimage.io.imsave does for float input.
"""
import os

import numpy as np

from .data import Hdf5Iterator, ImageDataGenerator      # noqa: F401  (util.Hdf5Iterator surface)


def convert_to_rgb(img, is_grayscale=False):
    """[C,H,W] network output -> [H,W,3] in [0,1] (util.py:69-84).

    One-channel images are replicated to three; ``is_grayscale=False`` means the values are in the tanh
    range and are mapped by (127.5*x + 127.5)/255; the result is clipped to [0,1]."""
    img = np.asarray(img)
    if img.ndim != 3:
        raise Exception("Image must have 3 dimensions (channels x height x width). Given %d" % img.ndim)
    ch = img.shape[0]
    if ch not in (1, 3):
        raise Exception("Unsupported number of channels. Must be 1 or 3, given %d." % ch)
    out = np.broadcast_to(img, (3,) + img.shape[1:]) if ch == 1 else img
    if not is_grayscale:
        out = (out * 127.5 + 127.5) / 255.
    return np.clip(np.transpose(out, (1, 2, 0)), 0, 1)


def compose_imgs(a, b, is_a_grayscale=True, is_b_grayscale=False):
    """a | b side by side as one [H,2W,3] image (util.py:87-99)."""
    left = convert_to_rgb(a, is_grayscale=is_a_grayscale)
    right = convert_to_rgb(b, is_grayscale=is_b_grayscale)
    if left.shape != right.shape:
        raise Exception("A and B must have the same size. %s != %s" % (left.shape, right.shape))
    return np.concatenate([left, right], axis=1).astype(np.float64)


_WRITES = [True]


class writes:
    """``with writes(False):`` -- imsave / plot_grid / makedirs below do everything but touch the file system
    (data-parallel ranks other than 0 still draw the batches and run the forward passes of the per-epoch dumps)."""

    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        _WRITES.append(self.enabled and _WRITES[-1])

    def __exit__(self, *exc):
        _WRITES.pop()


def makedirs(d):
    if _WRITES[-1]:
        os.makedirs(d, exist_ok=True)


def to_uint8(img01):
    """float [0,1] -> uint8 the way skimage's img_as_ubyte does (round half to even of 255*x)"""
    return np.rint(np.clip(np.asarray(img01, np.float64), 0, 1) * 255.0).astype(np.uint8)


def imsave(fname, arr):
    """PNG writer standing in for skimage.io.imsave(fname=..., arr=...) at pix2pix.py:303-304,325,418-422."""
    if not _WRITES[-1]:
        return
    from PIL import Image
    arr = np.asarray(arr)
    if arr.dtype != np.uint8:
        arr = to_uint8(arr)
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[:, :, 0]
    d = os.path.dirname(os.path.abspath(fname))
    os.makedirs(d, exist_ok=True)
    Image.fromarray(arr).save(fname)


def imread(fname):
    from PIL import Image
    return np.asarray(Image.open(fname))


def plot_grid(out_filename, itr, out_fn, is_a_grayscale, is_b_grayscale, N=4):
    """N x N figure of [A | B] pairs (util.py:101-116): one batch is drawn per cell, ``out_fn`` maps A -> B
    (``None`` shows the iterator's own B), only element 0 of each batch is shown."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt
    fig = plt.figure(figsize=(10, 6))
    for cell in range(N * N):
        a, b = next(itr) if hasattr(itr, '__next__') else itr.next()
        shown = b if out_fn is None else out_fn(a)
        ax = fig.add_subplot(N, N, cell + 1)
        ax.imshow(compose_imgs(a[0], shown[0], is_a_grayscale=is_a_grayscale, is_b_grayscale=is_b_grayscale))
        ax.axis('off')
    if _WRITES[-1]:
        fig.savefig(out_filename)
    plt.close('all')
