#!/usr/bin/env python
"""The BASELINE configuration at the reference's own batch size (bs = 4, experiments.py:121): one joint train step of
the four 512x512 test1_nobn_bilin_both networks on the float64 oracle (oracle/step.py), plus the same step in float32
(what any fp32 implementation -- the reference runs floatX=float32 -- can be expected to reproduce).

    python tests/golden/make_reference_step_fullsize_b4.py     # build container only; ~40 min of CPU
    -> tests/golden/reference_step_fullsize_b4.npz:
       losses64 / losses32           the five losses
       gz_sample64, ux_sample64      G(z) and U(X) on a fixed pixel lattice (every 8th row and column) + one
                                     full-resolution 64x64 window per image                     (+ *_32)
       grad/<net>/<i>                per trainable tensor of the float64 gradients: [sum, L2 norm, 8 sampled elements]
       grad32/<net>/<i>              the same from the float32 run (the conditioning spread)
       after/<net>/<i>               per parameter tensor after the RMSprop step: [sum, L2 norm, 8 sampled elements]
       meta                          [seed, batch, data seed, lattice stride, window]
tests/test_gpu_fullsize.py compares the HIP step on the MI355X against this file (no oracle run on the GPU box).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SEED, BATCH, DATA_SEED, STRIDE, WIN = 11, 4, 42, 8, 64


def summary(v):
    v64 = np.asarray(v, np.float64).ravel()
    idx = np.linspace(0, v64.size - 1, 8).astype(np.int64)
    return np.concatenate([[v64.sum(), np.sqrt((v64 * v64).sum())], v64[idx]])


def sample_image(a):
    """[B,C,512,512] -> (lattice [B,C,64,64], window [B,C,WIN,WIN] at a per-image offset)"""
    a = np.asarray(a, np.float64)
    lat = a[:, :, ::STRIDE, ::STRIDE].copy()
    win = np.stack([a[n, :, 37 * (n + 1):37 * (n + 1) + WIN, 53 * (n + 1):53 * (n + 1) + WIN] for n in range(a.shape[0])])
    return lat, win


def main():
    from oracle import step as S
    cfg = S.default_cfg()
    Z, X, Y = S.synthetic_batch(BATCH, cfg, seed=DATA_SEED)
    out = {"meta": np.array([SEED, BATCH, DATA_SEED, STRIDE, WIN], np.int64)}
    for tag, dt in (("64", np.float64), ("32", np.float32)):
        t0 = time.time()
        st = S.init_state(cfg, SEED, np.float32)
        res = S.train_step(st, Z, X, Y, dtype=dt, want=('gz', 'ux'))
        print("float%s: %.0f s, losses %r" % (tag, time.time() - t0, res['losses']), flush=True)
        out["losses" + tag] = np.asarray(res['losses'], np.float64)
        for k in ('gz', 'ux'):
            lat, win = sample_image(res[k])
            out["%s_sample%s" % (k, tag)] = lat
            out["%s_window%s" % (k, tag)] = win
        for key in S.NET_ORDER:
            for i, g in enumerate(res['grads'][key]):
                out["grad%s/%s_%s/%03d" % ("" if tag == "64" else "32", key[0], key[1], i)] = summary(g)
        if tag == "64":
            for key in S.NET_ORDER:
                for i, v in enumerate(st['params'][key[0]][key[1]]):
                    out["after/%s_%s/%03d" % (key[0], key[1], i)] = summary(v)
    path = os.environ.get("GHM_FIXTURE_OUT") or os.path.join(HERE, "reference_step_fullsize_b4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
