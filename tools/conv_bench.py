#!/usr/bin/env python
"""Micro-benchmark of one convolution geometry through the C ABI (fwd / dgrad / wgrad), HIP-event timed.
    python tools/conv_bench.py N C H W K k stride pad [--reps R] [--kinds fwd,dgrad,wgrad]
Used for kernel tuning and for rocprofv3 --pmc runs on a single layer."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_heightmaps_amd import device as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("geom", type=int, nargs=8, help="N C H W K k stride pad")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--warm-ms", type=float, default=60.0,
                    help="untimed warm-up: the GPU idles at ~160 MHz and takes milliseconds to reach its 2.4 GHz "
                         "shader clock, a 20-launch measurement from idle under-reports by 5-10 %%")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--q", default="", choices=["", "f32", "q", "both"],
                    help="low-precision kinds through the *_q entry points: operands are pre-packed q tensors, the "
                         "result is written as fp32, as a q tensor, or both")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16", "split", "split2"],
                    help="bf16 / f16: the *_lp entry points (kinds fwd, dgrad_t, wgrad); split: fp32 on the bf16 matrix "
                         "cores by operand splitting (kinds fwd, dgrad_t, pack_x; --q q: operands already split)")
    args = ap.parse_args()
    N, C, H, W, K, k, s, pad = args.geom
    dev = D.Device(0)
    ops = D.Ops(dev)
    d = D.conv_desc(N, C, H, W, K, k, k, s, pad)
    rng = np.random.RandomState(0)
    if os.environ.get('GHM_BENCH_ZERO'):        # all-zero operands: the same instruction stream without data toggling (power / clock probe)
        class _Z:
            randn = staticmethod(lambda *s: np.zeros(s))
        rng = _Z()
    x = dev.tensor(rng.randn(N, C, H, W).astype(np.float32))
    y = dev.tensor(rng.randn(N, K, d.Ho, d.Wo).astype(np.float32))
    w = dev.tensor((rng.randn(C * k * k * K) * 0.05).astype(np.float32))
    b = dev.tensor(rng.randn(K).astype(np.float32))
    dx = dev.empty((N, C, H, W))
    dw = dev.zeros((1, C * k * k * K, 1, 1))
    ws = dev.alloc(ops.wgrad_workspace(d))
    flops = 2.0 * N * K * d.Ho * d.Wo * C * k * k
    wT = dev.empty((1, C * k * k * K, 1, 1))
    if ops.dgrad_t_supported(d):
        ops.transpose_weights(d, w, wT)
    fns = {"fwd": lambda: ops.conv2d_fwd(d, x, w, b, y, 'lrelu', 0.2),
           "dgrad_t": lambda: ops.conv2d_dgrad_t(d, y, wT, dx),
           "dgrad": lambda: ops.conv2d_dgrad(d, y, w, dx),
           "wgrad": lambda: ops.conv2d_wgrad(d, x, y, dw, ws)}
    if args.dtype == "f32" and min(C, K) <= 4 and ops.thin_fwd_q_supported(d, 'lrelu', False, 'bf16x3'):
        # a first layer in the split step: fp32 result + the three-plane q copy in one pass (conv2d_fwd_thin_q)
        yq3 = D.QTensor.empty(dev, y.shape, 'bf16x3')
        fns["fwd_q"] = lambda: ops.conv2d_fwd_thin_q(d, x, w, b, y, yq3, 'lrelu', 0.2)
    if args.dtype in ("split", "split2"):
        npc = 3 if args.dtype == "split" else 2
        qdt = "bf16x3" if npc == 3 else "bf16x2"
        wq = dev.alloc(ops.split_weight_bytes(d, False, npc))
        wqT = dev.alloc(ops.split_weight_bytes(d, True, npc))
        ops.split_pack_weights(d, w, wq, False, npc)
        ops.split_pack_weights(d, w, wqT, True, npc)
        xs = dev.alloc(3 * N * C * H * W * 2)
        ys = dev.alloc(3 * N * K * d.Ho * d.Wo * 2)
        xq = (xs,) + ops.split_pack(x, xs, pieces=npc)
        yq = (ys,) + ops.split_pack(y, ys, pieces=npc)
        pre = args.q == "q"
        fns = {"pack_x": lambda: ops.split_pack(x, xs, pieces=npc), "pack": lambda: ops.split_pack_weights(d, w, wq, False, npc)}
        if ops.split_supported(d, 0):
            fns["fwd"] = lambda: ops.conv2d_fwd_split(d, x, wq, b, y, 'lrelu', 0.2, xq=xq if pre else None, pieces=npc)
        if ops.split_supported(d, 1):
            fns["dgrad_t"] = lambda: ops.conv2d_dgrad_split(d, y, wqT, dx, dyq=yq if pre else None, pieces=npc)
        if ops.split_supported(d, 2):
            xQ = D.QTensor(dev, xs, x.shape, qdt)
            yQ = D.QTensor(dev, ys, y.shape, qdt)
            ws_sp = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
            fns["wgrad"] = lambda: ops.conv2d_wgrad_lp_q(d, xQ, yQ, dw, ws_sp, qdt)
    elif args.dtype != "f32":
        dt = args.dtype
        wq = dev.alloc(ops.lp_weight_bytes(d, False))
        wqT = dev.alloc(ops.lp_weight_bytes(d, True))
        ops.lp_pack_weights(d, w, wq, dt, False)
        ops.lp_pack_weights(d, w, wqT, dt, True)
        ws_lp = dev.alloc(max(ops.wgrad_lp_workspace(d), 16))
        fns = {}
        if ops.lp_supported(d, 0, dt):
            fns["fwd"] = lambda: ops.conv2d_fwd_lp(d, x, wq, b, y, dt, 'lrelu', 0.2)
        if ops.lp_supported(d, 1, dt):
            fns["dgrad_t"] = lambda: ops.conv2d_dgrad_lp(d, y, wqT, dx, dt)
        if ops.lp_supported(d, 2, dt):
            fns["wgrad"] = lambda: ops.conv2d_wgrad_lp(d, x, y, dw, ws_lp, dt)
        if args.q:
            xq, yq, dxq = (D.QTensor.empty(dev, t.shape, dt) for t in (x, y, dx))
            ops.q_pack(x, xq)
            ops.q_pack(y, yq)
            o32 = args.q in ("f32", "both")
            oq = args.q in ("q", "both")
            if "fwd" in fns:
                fns["fwd"] = lambda: ops.conv2d_fwd_lp_q(d, xq, wq, b, y if o32 else None, yq if oq else None, dt, 'lrelu', 0.2)
            if "dgrad_t" in fns:
                fns["dgrad_t"] = lambda: ops.conv2d_dgrad_lp_q(d, yq, wqT, dx if o32 else None, dxq if oq else None, dt)
            if ops.lp_wgrad_q_supported(d, dt):
                fns["wgrad"] = lambda: ops.conv2d_wgrad_lp_q(d, xq, yq, dw, ws_lp, dt)
        fns["pack"] = lambda: ops.lp_pack_weights(d, w, wq, dt, False)
        fns["pack_t"] = lambda: ops.lp_pack_weights(d, w, wqT, dt, True)
    for i, kind in enumerate(args.kinds.split(",")):
        if kind not in fns:
            print("%-6s not served in %s" % (kind, args.dtype))
            continue
        fn = fns[kind]
        import time
        t0 = time.time()
        while True:
            for _ in range(10):
                fn()
            dev.sync()
            if (time.time() - t0) * 1e3 >= args.warm_ms:
                break
        dev.timer_start(0)
        for _ in range(args.reps):
            fn()
        dev.timer_stop(0)
        ms = dev.timer_ms(0) / args.reps
        name = ("split" if args.dtype == "split" else "lp<%s>" % args.dtype) if args.dtype != "f32" else \
            ops.conv_variant(d, ["fwd", "dgrad", "wgrad", "dgrad_t"].index(kind if kind != "fwd_q" else "fwd"))
        print("%-7s %-34s %8.3f ms  %7.1f TFLOP/s  (%.1f GFLOP)  N%d C%d %dx%d K%d k%d s%d" %
              (kind, name, ms, flops / ms / 1e9, flops / 1e9, N, C, H, W, K, k, s))
    dev.close()


if __name__ == "__main__":
    main()
