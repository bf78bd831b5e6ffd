#!/usr/bin/env python
"""ms per step over a long run, in windows of 20 steps (lr = 0 unless --train: every step then computes the same thing), beside the
shader clock / power / temperature rocm-smi reports -- is the step's speed-up over the first hundreds of steps the data or the chip?
    python tools/step_time_series.py [--dtype bf16x3] [--windows 40] [--train]"""
import argparse, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from gan_heightmaps_amd import device
from gan_heightmaps_amd.experiments import make_model


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
        rows = [r for r in out.strip().splitlines() if r and not r.startswith("WARNING")]
        hdr, val = rows[0].split(","), rows[1].split(",")
        keep = [i for i, h in enumerate(hdr) if any(t in h.lower() for t in ("sclk", "power", "junction", "edge", "mclk", "fclk"))]
        return " ".join("%s=%s" % (hdr[i].split("(")[0].strip()[:22], val[i]) for i in keep)
    except Exception as e:
        return "smi: %s" % e


ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16x3")
ap.add_argument("--windows", type=int, default=40)
ap.add_argument("--train", action="store_true")
ap.add_argument("--sleep-after", type=float, default=0.0, help="idle seconds after the series, then one more window (does the gain survive a pause?)")
args = ap.parse_args()
dev = device.Device(0)
model = make_model('test1_nobn_bilin_both', device=dev, use_graph='recorded', seed=0, verbose=False, dtype=args.dtype)
eng = model.engine
if not args.train:
    eng.set_lr(0.0)
Z, X, Y = synthetic_batch(4, 1000, 512, seed=1000)
b = eng.built(4)
eng._upload(b, Z, X, Y)
for _ in range(3):
    eng.enqueue_train(b)
eng.sync()
print("before:", smi(), flush=True)
t_start = time.perf_counter()
for w in range(args.windows):
    t0 = time.perf_counter()
    for _ in range(20):
        eng.enqueue_train(b)
    eng.sync()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    print("steps %4d-%4d  %7.3f ms  %6.1f img/s  t=%5.1fs  %s" % (20 * w, 20 * w + 19, ms, 4e3 / ms, time.perf_counter() - t_start,
                                                                smi() if w % 4 == 0 else ""), flush=True)
if args.sleep_after:
    time.sleep(args.sleep_after)
    for w in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            eng.enqueue_train(b)
        eng.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        print("after %.0f s idle: %7.3f ms  %6.1f img/s  %s" % (args.sleep_after, ms, 4e3 / ms, smi()), flush=True)
