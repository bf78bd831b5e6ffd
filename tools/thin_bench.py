#!/usr/bin/env python
"""the thin first-layer forwards of the step, isolated and warm, through the q entry points:
    python tools/thin_bench.py [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gan_heightmaps_amd import device as D
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = D.Device(0); ops = D.Ops(dev)
rng = np.random.RandomState(0)
def bench(name, fn, nbytes, reps=100):
    t0 = time.time()
    while (time.time() - t0) < 0.06:
        for _ in range(10): fn()
        dev.sync()
    dev.timer_start(0)
    for _ in range(reps): fn()
    dev.timer_stop(0)
    ms = dev.timer_ms(0) / reps
    print("%-34s %8.1f us   %6.0f GB/s of %5.1f MB written+read" % (name, ms * 1e3, nbytes / ms / 1e6, nbytes / 1e6))
for (N, C, H, K, k, s, pad, pooled, q, f32) in [(8, 1, 512, 64, 5, 1, 2, True, True, False), (8, 1, 512, 64, 5, 1, 2, True, True, True),
                                                (4, 1, 512, 64, 3, 2, 1, False, False, True), (8, 4, 512, 64, 3, 2, 1, False, True, True)]:
    d = D.conv_desc(N, C, H, H, K, k, k, s, pad)
    x = dev.tensor(rng.randn(N, C, H, H).astype(np.float32))
    w = dev.tensor((rng.randn(C * k * k * K) * 0.05).astype(np.float32))
    b = dev.tensor(rng.randn(K).astype(np.float32))
    if pooled:
        shp = (N, K, H // 2, H // 2)
        y = dev.empty(shp) if f32 else None
        yq = D.QTensor.empty(dev, shp, dt)
        mask = dev.alloc(int(np.prod(shp)))
        nb = 4 * N * C * H * H + int(np.prod(shp)) * (1 + 2 + (4 if f32 else 0))
        bench("pool N%d C%d %d K%d k%d %s" % (N, C, H, K, k, "f32+q" if f32 else "q"),
              lambda: ops.conv2d_fwd_pool_thin_q(d, x, w, b, y, mask, yq, 'lrelu', 0.2), nb)
    else:
        shp = (N, K, d.Ho, d.Wo)
        y = dev.empty(shp) if f32 else None
        yq = D.QTensor.empty(dev, shp, dt) if q else None
        nb = 4 * N * C * H * H + int(np.prod(shp)) * ((2 if q else 0) + (4 if f32 else 0))
        if q:
            fn = lambda: ops.conv2d_fwd_thin_q(d, x, w, b, y, yq, 'lrelu', 0.2)
        else:
            fn = lambda: ops.conv2d_fwd(d, x, w, b, y, 'linear', 0.0)
        bench("N%d C%d %d K%d k%d s%d %s" % (N, C, H, K, k, s, ("f32+q" if f32 else "q") if q else "f32"), fn, nb)
dev.close()
