#!/usr/bin/env python
"""HBM-bound kernels of the step at one representative geometry, HIP-event timed through the C ABI:
algorithmic bytes (tensor passes) / time.   python tools/ew_bench.py [N C H W]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device as D  # noqa: E402


def main():
    N, C, H, W = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (4, 64, 256, 256)
    dev = D.Device(0)
    ops = D.Ops(dev)
    rng = np.random.RandomState(0)
    full = lambda: dev.tensor(rng.randn(N, C, H, W).astype(np.float32))       # noqa: E731
    x, y, g, dx = full(), full(), full(), dev.empty((N, C, H, W))
    half = dev.tensor(rng.randn(N, C, H // 2, W // 2).astype(np.float32))
    half2 = dev.empty((N, C, H // 2, W // 2))
    vec = lambda: dev.tensor(rng.rand(1, C, 1, 1).astype(np.float32) + 0.5)  # noqa: E731
    m, iv, gam, bet, dg, db = vec(), vec(), vec(), vec(), vec(), vec()
    ws = dev.alloc(ops.bn_workspace(C))
    nb = N * C * H * W * 4

    def t(fn, reps=20):
        for _ in range(3):
            fn()
        dev.sync()
        dev.timer_start(0)
        for _ in range(reps):
            fn()
        dev.timer_stop(0)
        return dev.timer_ms(0) / reps

    rows = [
        ("bn_stats      (1 pass)", 1.0, lambda: ops.bn_stats(x, m, iv, ws)),
        ("bn_apply      (2)", 2.0, lambda: ops.bn_apply(x, y, m, iv, gam, bet, 'relu')),
        ("bn_backward   (3 + 4)", 7.0, lambda: ops.bn_backward(g, y, x, dx, m, iv, gam, dg, db, ws, 'relu')),
        ("act_fwd       (2)", 2.0, lambda: ops.act_fwd(x, y, 'lrelu', 0.2)),
        ("act_bwd       (3)", 3.0, lambda: ops.act_bwd(g, y, dx, 'lrelu', 0.2)),
        ("channel_sum   (1)", 1.0, lambda: ops.channel_sum(g, db)),
        ("maxpool2_fwd  (1.25)", 1.25, lambda: ops.maxpool2_fwd(x, half2)),
        ("maxpool2_bwd  (2.5)", 2.5, lambda: ops.maxpool2_bwd(x, half2, half, dx)),
        ("bilinear_fwd  (1.25)", 1.25, lambda: ops.upsample_bilinear2_fwd(half, y)),
        ("bilinear_bwd  (1.25)", 1.25, lambda: ops.upsample_bilinear2_bwd(g, half2)),
        ("copy_view     (2)", 2.0, lambda: ops.copy_view(x, y)),
    ]
    print("N%d C%d %dx%d  (%.0f MB per tensor pass)" % (N, C, H, W, nb / 1e6))
    for name, passes, fn in rows:
        ms = t(fn)
        print("%-24s %.3f ms  %.2f TB/s" % (name, ms, passes * nb / ms / 1e9))


if __name__ == "__main__":
    main()
