#!/usr/bin/env python
"""End-to-end train images/s THROUGH the data path (uint8 upload over PCIe + ghm_image_batch augmentation + step +
5-float read-back per step), next to bench.py's resident-input number.   python tools/train_throughput.py [steps] [f32|bf16|f16]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device, data  # noqa: E402
from gan_heightmaps_amd.experiments import make_model, synthetic_arrays  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
dev = device.Device(0)
model = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, dtype=dtype, use_graph='recorded')
X, Y = synthetic_arrays(64, 512, True, False, 0)
imgen = data.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
it = data.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
z = lambda n: np.random.rand(n, 1000).astype(np.float32)
for _ in range(3):
    model.engine.run_from_iterator(it, z)
t0 = time.perf_counter()
hist = []
for _ in range(steps):
    losses = model.engine.run_from_iterator(it, z)
    hist.append([float(v) for v in losses])
dt = time.perf_counter() - t0
hist = np.asarray(hist)
assert np.isfinite(hist).all(), "non-finite loss at step %d" % int(np.argwhere(~np.isfinite(hist))[0][0])
print("%s through the data path: %.1f img/s (%.2f ms/step, synchronous: upload -> augment -> step -> read losses)"
      % (dtype, 4 * steps / dt, 1e3 * dt / steps))
# the product's loop (Pix2Pix.train with prefetch=True): batch i+1 is drawn, uploaded (page-locked staging, copy stream) and
# augmented while step i runs; the five losses are still read back every step
it2 = data.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
for _ in model.engine.train_pipelined_from_iterator(it2, z, 6):
    pass
t0 = time.perf_counter()
hist2 = [[float(v) for v in l] for l in model.engine.train_pipelined_from_iterator(it2, z, steps)]
dt = time.perf_counter() - t0
assert np.isfinite(np.asarray(hist2)).all()
print("%s through the data path, pipelined: %.1f img/s (%.2f ms/step: batch i+1 uploaded + augmented under step i)"
      % (dtype, 4 * steps / dt, 1e3 * dt / steps))
print("  losses first  %s\n  losses last   %s\n  mean of last 20 %s" % (hist[0].round(4).tolist(), hist[-1].round(4).tolist(),
                                                                        hist[-20:].mean(0).round(4).tolist()))
