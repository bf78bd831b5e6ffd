#!/usr/bin/env python
"""How long does the HOST need to issue one recorded step (ghm_step_run) against how long the GPU needs to run it?
    python tools/host_issue_time.py [dtype]
If the two are close the step is issue-bound, not kernel-bound."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device  # noqa: E402
from gan_heightmaps_amd.experiments import make_model  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = device.Device(0)
model = make_model('test1_nobn_bilin_both', device=dev, use_graph='recorded', seed=0, verbose=False, dtype=dtype)
eng = model.engine
b = eng.built(4)
rng = np.random.RandomState(0)
eng._upload(b, rng.rand(4, 1000).astype(np.float32), rng.rand(4, 1, 512, 512).astype(np.float32),
            rng.rand(4, 3, 512, 512).astype(np.float32) * 2 - 1)
for _ in range(4):
    eng.enqueue_train(b)
eng.sync()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    eng.enqueue_train(b)
t1 = time.perf_counter()
eng.sync()
t2 = time.perf_counter()
seq = eng._sequence(b, 'train')
print("%s: %d program entries per step; host issue %.3f ms/step, GPU step %.3f ms (issue / step = %.2f)"
      % (dtype, len(seq), 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n, (t1 - t0) / (t2 - t0)))
# the same with empty queues: one step issued, then drained, five times -- the host's own cost of a replay
ts = []
for _ in range(5):
    eng.sync()
    t0 = time.perf_counter()
    eng.enqueue_train(b)
    ts.append(time.perf_counter() - t0)
    eng.sync()
print("%s: host issue of ONE step into empty queues: %s ms" % (dtype, " ".join("%.2f" % (1e3 * t) for t in ts)))
