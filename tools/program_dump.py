#!/usr/bin/env python
"""The train program of the headline workload in issue order, stream by stream, every entry timed alone (median of three
instrumented steps): what a stage stream's dependent launch chain is made of.
    python tools/program_dump.py [--dtype bf16] [--mode both]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gan_heightmaps_amd import device  # noqa: E402
from gan_heightmaps_amd.experiments import make_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=4)
args = ap.parse_args()
dev = device.Device(0)
model = make_model('test1_nobn_bilin_both', device=dev, use_graph=False, seed=0, verbose=False, dtype=args.dtype)
eng = model.engine
B = args.batch
rs = np.random.RandomState(0)
b = eng.built(B)
eng._upload(b, rs.rand(B, 1000).astype(np.float32), rs.rand(B, 1, 512, 512).astype(np.float32),
            rs.rand(B, 3, 512, 512).astype(np.float32) * 2 - 1)
for _ in range(2):
    eng.enqueue_train(b)
eng.sync()
runs = [eng.profile_train(B) for _ in range(3)]
tot = {}
for r in zip(*runs):
    label, meta, lane = r[0][0], r[0][2], r[0][3]
    ms = sorted(x[1] for x in r)[1]
    tot[lane] = tot.get(lane, 0.0) + ms
    small = ""
    print("%-2s %-18s %8.4f ms  %-44s %s" % (lane, label, ms, meta["kernel"] if meta else "", meta["geom"] if meta else ""))
print("lanes:", {k: round(v, 3) for k, v in tot.items()}, "entries:", len(runs[0]))
