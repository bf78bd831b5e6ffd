#!/usr/bin/env python
"""The 5x5 weight gradient of a conv -> activation -> MaxPool2D(2) layer, dense split kernel against the sparse-instruction form
(ghm_conv2d_wgrad_pooled_split), each alone, HIP-event timed, with the distance of both from a float64 contraction on a small case.
    python tools/pooled_wgrad_bench.py [--ties FRACTION_OF_ROWS]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gan_heightmaps_amd import device as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ties", type=float, default=0.0, help="fraction of pooled rows given tied windows (those rows run densely)")
ap.add_argument("--dtype", default="bf16x3")
args = ap.parse_args()
dev = D.Device(0)
ops = D.Ops(dev)


def setup(N, C, K, H, W, dtype, ties, seed):
    rng = np.random.RandomState(seed)
    Ho, Wo = H // 2, W // 2
    x = rng.randn(N, C, H, W).astype(np.float32)
    gp = rng.randn(N, K, Ho, Wo).astype(np.float32)
    mask = (1 << rng.randint(0, 4, size=(N, K, Ho, Wo))).astype(np.uint8)
    for i in range(Ho):
        if rng.rand() < ties:
            mask[:, ::3, i, ::5] = 0b1111
    d = D.conv_desc(N, C, H, W, K, 5, 5, 1, 2)
    xq = D.QTensor.empty(dev, x.shape, dtype)
    ops.q_pack(dev.tensor(x), xq)
    mptr = dev.alloc(mask.size)
    dev.h2d(mptr, mask)
    gpt = dev.tensor(gp)
    dyq, dx = D.QTensor.empty(dev, (N, K, H, W), dtype), dev.empty((N, K, H, W))
    ops.maxpool2_mask_bwd_q(mptr, None, gpt, dx, dyq, 'linear', 0.0)
    cq = D.QTensor.empty(dev, (N, K, H, W // 2), dtype)
    idx, flags = dev.alloc(N * (K // 8) * H * (W // 32) * 16 + 256), dev.alloc(N * H * 4 + 256)
    comp = lambda: ops.maxpool2_mask_bwd_compress_q(mptr, None, gpt, cq, idx, flags, 'linear', 0.0)  # noqa: E731
    comp()
    fl = np.zeros(N * H, np.int32)
    dev.sync()
    dev.d2h(fl, flags, fl.nbytes)
    ws = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
    a = dev.zeros((1, C * 25 * K, 1, 1))
    dense = lambda: ops.conv2d_wgrad_lp_q(d, xq, dyq, a, ws, dtype)  # noqa: E731
    sparse = lambda: ops.conv2d_wgrad_pooled_split(d, xq, dyq, cq, idx, flags, a, ws, dtype)  # noqa: E731
    return d, x, dx, a, dense, sparse, comp, float((fl != 0).mean())


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    dev.sync()
    dev.timer_start(0)
    for _ in range(reps):
        fn()
    dev.timer_stop(0)
    return dev.timer_ms(0) / reps


d, x, dx, a, dense, sparse, comp, _ = setup(2, 64, 128, 32, 64, args.dtype, args.ties, 1)
xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (2, 2), (2, 2)))
dyd = dx.numpy().astype(np.float64)
ref = np.stack([np.einsum('nchw,nkhw->ck', xp[:, :, ta:ta + 32, tb:tb + 64], dyd) for ta in range(5) for tb in range(5)], 1).ravel()
dense()
A = a.numpy().ravel().copy()
sparse()
B = a.numpy().ravel()
print("N2 C64 K128 32x64 %s: |dense - f64| / max %.2e, |sparse - f64| / max %.2e" % (args.dtype, np.abs(A - ref).max() / np.abs(ref).max(),
                                                                                      np.abs(B - ref).max() / np.abs(ref).max()))
for (N, C, K, H, W) in [(8, 64, 128, 256, 256), (8, 128, 128, 128, 128), (8, 128, 128, 64, 64), (8, 128, 256, 32, 32)]:
    d, x, dx, a, dense, sparse, comp, frac = setup(N, C, K, H, W, args.dtype, args.ties, 9)
    fl = 2.0 * N * H * W * C * K * 25
    td, ts, tc = timed(dense), timed(sparse), timed(comp)
    print("N%d C%d %dx%d K%d k5 (dense rows %.0f %%): dense %.3f ms %.1f TFLOP/s | sparse form %.3f ms %.1f TFLOP/s by the dense count | operand pass %.3f ms"
          % (N, C, H, W, K, 100 * frac, td, fl / td / 1e9, ts, fl / ts / 1e9, tc))
