#!/usr/bin/env python
"""HBM traffic per launch of every conv kernel of the bench step, from rocprofv3 PMC passes.

Run on the GPU box (two separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --one-stream
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --one-stream
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/r01_pmc_traffic.json
Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B as reported by rocprofv3 (hbm_bytes = value * 1024); on gfx950 FETCH_SIZE under-reports wide coalesced
streaming reads by exactly 2x, so the read side is doubled.  WRITE_SIZE is taken as reported (uncalibrated).
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    m = re.match(r"void (\w+<[^>]*>)", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        fr = sum(fetch[k]) / len(fetch[k]) * 1024 * 2.0          # gfx950: x2 on coalesced reads
        wr = (sum(write[k]) / len(write[k]) * 1024) if k in write else 0.0
        out[short(k)] = {"launches_sampled": len(fetch[k]), "fetch_bytes_per_launch": fr, "write_bytes_per_launch": wr,
                         "hbm_bytes_per_launch": fr + wr}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
