#!/usr/bin/env python
"""Timeline of ONE steady-state train step from a rocprofv3 --kernel-trace CSV: per-queue busy time, the union busy time,
the concurrency histogram and (with --dump) every kernel with its start offset.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline
    python tools/timeline.py gpurun_out/tl [--marker upconv_collapse] [--step -2] [--dump]
A step is delimited by two consecutive launches of the marker kernel (one launch per step)."""
import argparse
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--marker", default="upconv_collapse")
    ap.add_argument("--step", type=int, default=-2, help="which marker interval (negative: from the end)")
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--gaps", type=float, default=15.0, help="report per-queue idle gaps longer than this many microseconds")
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % args.dir)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if args.marker in r[3]]
    if len(marks) < 3:
        sys.exit("marker %r seen %d times" % (args.marker, len(marks)))
    a, b = marks[args.step - 1 if args.step < 0 else args.step], marks[args.step if args.step < 0 else args.step + 1]
    step = rows[a:b]
    t0 = step[0][0]
    t1 = rows[b][0]
    span = (t1 - t0) / 1e3
    print("step = %d kernels, %.1f us between two marker launches" % (len(step), span))
    queues = sorted({r[2] for r in step})
    for q in queues:
        ks = [r for r in step if r[2] == q]
        busy = sum(r[1] - r[0] for r in ks) / 1e3
        print("  queue %-4s %4d kernels  busy %8.1f us (%.0f %%)  first at +%.1f  last ends +%.1f" %
              (q, len(ks), busy, 100 * busy / span, (ks[0][0] - t0) / 1e3, (max(r[1] for r in ks) - t0) / 1e3))
        prev = None
        for r in ks:
            if prev is not None and (r[0] - prev[1]) / 1e3 > args.gaps:
                print("      idle %7.1f us at +%8.1f  after %-40s before %s" %
                      ((r[0] - prev[1]) / 1e3, (prev[1] - t0) / 1e3, short(prev[3])[:40], short(r[3])[:40]))
            if prev is None or r[1] > prev[1]:
                prev = r
    # union busy time and concurrency histogram
    ev = []
    for r in step:
        ev.append((r[0], 1))
        ev.append((min(r[1], t1), -1))
    ev.sort()
    level, last, hist = 0, t0, {}
    for t, d in ev:
        hist[level] = hist.get(level, 0) + (t - last)
        last = t
        level += d
    hist[level] = hist.get(level, 0) + max(0, t1 - last)
    print("  kernels in flight:  " + "  ".join("%d: %.0f us (%.0f %%)" % (k, v / 1e3, 100 * v / 1e3 / span) for k, v in sorted(hist.items())))
    print("  sum of kernel durations %.1f us = %.2f x the step" % (sum(r[1] - r[0] for r in step) / 1e3, sum(r[1] - r[0] for r in step) / 1e3 / span))
    if args.dump:
        for r in step:
            print("%9.1f %8.1f  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], short(r[3])))


if __name__ == "__main__":
    main()
