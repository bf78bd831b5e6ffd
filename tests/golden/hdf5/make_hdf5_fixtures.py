#!/usr/bin/env python
"""Writes the HDF5 fixtures of tests/test_h5lite.py with the REAL library (h5py 3.3 / HDF5 1.10.6, the conda
interpreter of the build container: /opt/conda/bin/python3.9 tests/golden/hdf5/make_hdf5_fixtures.py).  The product
interpreter has no h5py; gan_heightmaps_amd/h5lite.py must read these byte for byte.  Contents are functions of fixed
seeds (``arrays()``), so the tests rebuild the expected values without the library.

  ref_layout.h5      the reference's own recipe (notebooks/prototype_cropping_code.ipynb cell 17): create_dataset(name,
                     shape, dtype='uint8') for xt / yt / xv / yv, filled by slice assignment -- contiguous, old-style group
  chunked.h5         chunked datasets: plain, gzip, gzip+shuffle, gzip+shuffle+fletcher32, ragged edge chunks, a chunk
                     that was never written, float32 / int16 big-endian / float64, a nested group, a scalar, a compact one
  latest.h5          libver='latest': superblock v3, version-2 object headers, link messages, contiguous data
  many.h5            40 datasets in one group (several symbol-table nodes) and a dataset with enough chunks for a
                     two-level chunk B-tree
  wide.h5            300 scalar datasets in one group: a two-level group B-tree
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def arrays():
    r = np.random.RandomState(1234)
    a = {}
    a["ref_layout.h5"] = {"xt": r.randint(0, 256, (6, 16, 16, 1)).astype(np.uint8),
                          "yt": r.randint(0, 256, (6, 16, 16, 3)).astype(np.uint8),
                          "xv": r.randint(0, 256, (2, 16, 16, 1)).astype(np.uint8),
                          "yv": r.randint(0, 256, (2, 16, 16, 3)).astype(np.uint8)}
    a["chunked.h5"] = {"plain": r.randint(0, 256, (7, 10, 10, 3)).astype(np.uint8),
                       "gz": r.randint(0, 4, (7, 10, 10, 3)).astype(np.uint8),
                       "gzshuf": (r.randn(5, 9, 6) * 100).astype(np.float32),
                       "gzshuf_fl": r.randint(-3000, 3000, (9, 5)).astype(">i2"),
                       "holes": np.concatenate([r.randint(0, 256, (4, 6)), np.zeros((4, 6)), r.randint(0, 256, (4, 6))]
                                               ).astype(np.uint8),
                       "grp/inner/f64": r.randn(3, 4),
                       "scalar": np.float32(2.5),
                       "compact": np.arange(12, dtype=np.int32).reshape(3, 4)}
    a["latest.h5"] = {"xt": r.randint(0, 256, (5, 8, 8, 1)).astype(np.uint8),
                      "yt": r.randint(0, 256, (5, 8, 8, 3)).astype(np.uint8),
                      "f32": r.randn(4, 7).astype(np.float32)}
    many = {"d%02d" % i: r.randint(0, 256, (3, 4)).astype(np.uint8) for i in range(40)}
    many["deep"] = r.randint(0, 256, (300, 4)).astype(np.uint8)
    a["many.h5"] = many
    a["wide.h5"] = {"s%03d" % i: np.uint8(r.randint(0, 256)) for i in range(300)}
    return a


def main():
    import h5py
    A = arrays()
    d = A["ref_layout.h5"]
    with h5py.File(os.path.join(HERE, "ref_layout.h5"), "w") as f:
        for k in ("xt", "yt", "xv", "yv"):
            f.create_dataset(k, d[k].shape, dtype="uint8")
        for k in ("xt", "yt", "xv", "yv"):
            f[k][0:len(d[k])] = d[k]
    d = A["chunked.h5"]
    with h5py.File(os.path.join(HERE, "chunked.h5"), "w") as f:
        f.create_dataset("plain", data=d["plain"], chunks=(2, 4, 10, 3))
        f.create_dataset("gz", data=d["gz"], chunks=(3, 5, 5, 3), compression="gzip", compression_opts=4)
        f.create_dataset("gzshuf", data=d["gzshuf"], chunks=(2, 4, 6), compression="gzip", shuffle=True)
        f.create_dataset("gzshuf_fl", data=d["gzshuf_fl"], chunks=(4, 5), compression="gzip", shuffle=True,
                         fletcher32=True)
        h = f.create_dataset("holes", (12, 6), dtype="uint8", chunks=(4, 6))
        h[0:4] = d["holes"][0:4]
        h[8:12] = d["holes"][8:12]                      # rows 4..7: a chunk that is never allocated
        f.create_group("grp").create_group("inner").create_dataset("f64", data=d["grp/inner/f64"])
        f.create_dataset("scalar", data=d["scalar"])
        dc = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dc.set_layout(h5py.h5d.COMPACT)
        sid = h5py.h5s.create_simple((3, 4))
        did = h5py.h5d.create(f.id, b"compact", h5py.h5t.NATIVE_INT32, sid, dcpl=dc)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, np.ascontiguousarray(d["compact"]))
    d = A["latest.h5"]
    with h5py.File(os.path.join(HERE, "latest.h5"), "w", libver="latest") as f:
        for k, v in d.items():
            f.create_dataset(k, data=v)
    d = A["many.h5"]
    with h5py.File(os.path.join(HERE, "many.h5"), "w") as f:
        for k, v in d.items():
            if k == "deep":
                f.create_dataset(k, data=v, chunks=(1, 4))
            else:
                f.create_dataset(k, data=v)
    with h5py.File(os.path.join(HERE, "wide.h5"), "w") as f:
        for k, v in A["wide.h5"].items():
            f.create_dataset(k, data=v)
    for name in A:
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
