#!/usr/bin/env python
"""Isolated timing of the d_conv1 backward from the pooled operands (csrc/conv_pool_bwd.hip) against the materialised path
it replaces (mask pass + dense thin kernels):  python tools/pool_bwd_bench.py [N H W K] [--reps R]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("geom", type=int, nargs="*", default=[8, 512, 512, 64])
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    N, H, W, K = args.geom
    dev = D.Device(0)
    ops = D.Ops(dev)
    rng = np.random.RandomState(0)
    d = D.conv_desc(N, 1, H, W, K, 5, 5, 1, 2)
    x = dev.tensor(rng.randn(N, 1, H, W).astype(np.float32))
    wp = dev.tensor((rng.randn(25 * K) * 0.1).astype(np.float32))
    b = dev.tensor(rng.randn(K).astype(np.float32))
    pooled = dev.empty((N, K, H // 2, W // 2))
    mask = dev.alloc(N * K * (H // 2) * (W // 2))
    ops.conv2d_fwd_pool(d, x, wp, b, pooled, mask, 'lrelu', 0.2, 'f32')
    gp = dev.tensor(rng.randn(N, K, H // 2, W // 2).astype(np.float32))
    dw, db, dx = dev.zeros((1, 25 * K, 1, 1)), dev.zeros((1, K, 1, 1)), dev.zeros((N, 1, H, W))
    Gf = dev.empty((N, K, H, W))
    ws = dev.alloc(max(ops.wgrad_workspace(d), ops.pool_wgrad_sparse_workspace(d), 16))
    fns = {
        "wgrad sparse": lambda: ops.conv2d_pool_wgrad_sparse(d, x, mask, pooled, gp, dw, db, ws, 'lrelu', 0.2),
        "dgrad sparse": lambda: ops.conv2d_pool_dgrad_sparse(d, mask, pooled, gp, wp, dx, 'lrelu', 0.2),
        "mask pass (+bias)": lambda: ops.maxpool2_mask_bwd(mask, pooled, gp, Gf, 'lrelu', 0.2, db),
        "wgrad dense": lambda: ops.conv2d_wgrad(d, x, Gf, dw, ws),
        "dgrad dense": lambda: ops.conv2d_dgrad(d, Gf, wp, dx),
    }
    for name, fn in fns.items():
        for _ in range(10):
            fn()
        dev.sync()
        dev.timer_start(0)
        for _ in range(args.reps):
            fn()
        dev.timer_stop(0)
        print("%-20s %8.3f ms" % (name, dev.timer_ms(0) / args.reps))
    dev.close()


if __name__ == "__main__":
    main()
