#!/usr/bin/env python
"""End-to-end train images/s THROUGH the data path (uint8 upload over PCIe + ghm_image_batch augmentation + step +
5-float read-back per step), next to bench.py's resident-input number.   python tools/train_throughput.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device, data  # noqa: E402
from gan_heightmaps_amd.experiments import make_model, synthetic_arrays  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = device.Device(0)
model = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False)
X, Y = synthetic_arrays(64, 512, True, False, 0)
imgen = data.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
it = data.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
z = lambda n: np.random.rand(n, 1000).astype(np.float32)
for _ in range(3):
    model.engine.run_from_iterator(it, z)
t0 = time.perf_counter()
for _ in range(steps):
    losses = model.engine.run_from_iterator(it, z)
dt = time.perf_counter() - t0
print("through the data path: %.1f img/s (%.2f ms/step, synchronous: upload -> augment -> step -> read losses)"
      % (4 * steps / dt, 1e3 * dt / steps), [float(v) for v in losses])
