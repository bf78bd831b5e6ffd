#!/usr/bin/env python
"""HBM (fabric) traffic per launch of every conv kernel of the bench step, from rocprofv3 PMC passes.

Run on the GPU box (two separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; counters only,
no --sys-trace / --stats next to --pmc):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --one-stream --issue eager [--dtype bf16]
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --one-stream --issue eager [--dtype bf16]
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/r02_pmc_traffic.json
Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B, and
on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read.  The guide calibrates that only for
16 B per lane, so the access patterns of THESE kernels were calibrated on the box (tools/pmc_calibrate.hip, 1 GiB
streamed once; profiles/r02_pmc_calibration.txt): FETCH_SIZE * 1024 / bytes = 0.500 for plain 4-byte loads, plain
16-byte loads, 4-byte global->LDS DMA and 16-byte global->LDS DMA alike; WRITE_SIZE * 1024 / bytes = 1.000 for 4- and
16-byte stores.  So the read side is doubled for every kernel and the write side is taken as reported.  The counters
sit on the memory side of L2, so Infinity-Cache hits are included: this is fabric traffic, an upper bound on HBM bytes.
"""
import collections
import csv
import glob
import json
import re
import sys

LP_DT = {"1": "bf16", "2": "f16"}


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"void (\w+<[^>]*>)", name)
    s = m.group(1) if m else re.sub(r"^void ", "", name).split("(")[0]
    m = re.match(r"(lp_dgrad_s2_kernel)<(\d)", s)
    if m:
        return "%s<%s, 3, 2>" % (m.group(1), LP_DT.get(m.group(2), m.group(2)))
    m = re.match(r"(lp_(?:conv|wgrad|wgrad_q)_kernel)<(\d), (\d+), (\d+)", s)
    if m:                                   # the label bench.py / engine.conv_meta use: <dtype, k, stride>
        return "%s<%s, %s, %s>" % (m.group(1).replace("wgrad_q", "wgrad"), LP_DT.get(m.group(2), m.group(2)), m.group(3), m.group(4))
    m = re.match(r"(lp_dgrad_s2_kernel)<(\d)", s)
    if m:
        return "%s<%s, 3, 2>" % (m.group(1), LP_DT.get(m.group(2), m.group(2)))
    return s


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    agg = collections.defaultdict(lambda: [[], []])
    for k, v in fetch.items():
        agg[short(k)][0] += v
    for k, v in write.items():
        agg[short(k)][1] += v
    out = {}
    for k, (fv, wv) in agg.items():
        if not fv:
            continue
        raw = sum(fv) / len(fv) * 1024
        wr = (sum(wv) / len(wv) * 1024) if wv else 0.0
        out[k] = {"launches_sampled": len(fv), "fetch_counter_bytes": raw, "fetch_bytes_per_launch": 2 * raw,
                  "write_bytes_per_launch": wr, "hbm_bytes_per_launch": 2 * raw + wr}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
