#!/usr/bin/env python
"""The same training run in two arithmetic modes, step by step: identical initial weights, data order, augmentation draws,
noise and dropout masks; the five losses of every step side by side.  Training is chaotic, so the runs separate eventually --
what the table shows is WHEN and how fast: fp32 on the fp32 matrix instruction against fp32 by operand splitting (bf16x3, the
default) should separate like two fp32 implementations do (summation order), not like a reduced-precision run.
    python tools/trajectory_agreement.py [steps] [modeA] [modeB]        (default 40 f32 bf16x3)
A mode is a dtype, optionally followed by +NAME=VALUE tuning switches set for that run only, e.g.
    python tools/trajectory_agreement.py 40 bf16x3 bf16x3+GHM_NO_RANK_ONE=1
(the DCGAN generator's gradient as a per-sample multiple of the discriminator-loss pass, DESIGN 4e, against the two separate passes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device, data  # noqa: E402
from gan_heightmaps_amd.experiments import make_model, synthetic_arrays  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
modes = (sys.argv[2] if len(sys.argv) > 2 else 'f32', sys.argv[3] if len(sys.argv) > 3 else 'bf16x3')


def run(mode):
    dtype, *switches = mode.split('+')
    for sw in switches:
        os.environ[sw.split('=')[0]] = sw.split('=', 1)[1]
    try:
        return _run(dtype)
    finally:
        for sw in switches:
            os.environ.pop(sw.split('=')[0], None)


def _run(dtype):
    dev = device.Device(0)
    model = make_model('test1_nobn_bilin_both', device=dev, seed=0, verbose=False, dtype=dtype, use_graph='recorded')
    X, Y = synthetic_arrays(64, 512, True, False, 0)
    np.random.seed(1234)                      # data order, augmentation draws and the generator's noise
    imgen = data.ImageDataGenerator(horizontal_flip=True, vertical_flip=True, rotation_range=360, fill_mode="reflect")
    it = data.Hdf5Iterator(X, Y, 4, imgen, True, False, device=dev)
    z = lambda n: np.random.rand(n, 1000).astype(np.float32)
    hist = [[float(v) for v in model.engine.run_from_iterator(it, z)] for _ in range(steps)]
    dev.close()
    return np.asarray(hist)


a, b = run(modes[0]), run(modes[1])
names = ("D(dcgan)", "G(dcgan)", "D(p2p)", "G(p2p)", "recon")
print("step  " + "  ".join("%-22s" % n for n in names) + "   (%s | relative difference of %s)" % (modes[0], modes[1]))
for i in range(steps):
    rel = np.abs(a[i] - b[i]) / np.maximum(np.abs(a[i]), 1e-12)
    print("%4d  " % i + "  ".join("%10.6f %9.1e  " % (a[i][k], rel[k]) for k in range(5)))
rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-12)
for n in (1, 5, 10, 20, steps):
    if n <= steps:
        print("worst relative difference of any loss over the first %3d steps: %.2e" % (n, rel[:n].max()))
assert np.isfinite(a).all() and np.isfinite(b).all()
