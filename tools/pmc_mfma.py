#!/usr/bin/env python
"""MFMA utilisation per kernel of the bench step from rocprofv3 PMC passes (SQ counters; one pass, <= 8 SQ slots).

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 \
        SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
        --output-format csv -d <dir> -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary [--dtype bf16]
    python tools/pmc_mfma.py <dir> [<kernel-trace dir for durations>] > profiles/r04_pmc_mfma_f32.json

Reading the counters (MI355X_MICROARCH.md "rocprofv3 PMC slots", per-instruction table):
  * SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles a SIMD's matrix pipe is busy, summed over SIMDs (and XCDs);
  * SQ_BUSY_CYCLES counts cycles the SQ (per shader engine) has any wave, so  mfma_busy / (busy x SIMDs-per-SQ-instance)
    depends on the aggregation; the robust form used here is
        mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel wall cycles x 1024 SIMDs)
    with wall cycles = GRBM_GUI_ACTIVE / 8 of the same dispatch (the counter arrives summed over the 8 XCDs: value /
    duration = 8 x the shader clock), else duration x 2.4 GHz;
  * SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16,F16}: matrix operations issued, in units of 512 FLOP-pairs (guide: gfx94x formula);
    reported raw per launch and as a cross-check  flops = MOPS x 512 x 2 .
rocprofv3 SERIALISES the dispatches of a process while it collects counters (the bench line printed under --pmc shows it:
the three-stream step runs at the one-stream rate), so every figure here is the kernel ALONE on the GPU -- the matrix pipe's
busy fraction each kernel reaches by itself, to hold against profiles/*_kernel_table*.txt (isolated durations) and against
the in-step durations of the --kernel-trace --stats run."""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import short  # noqa: E402

SIMDS = 256 * 4


def main():
    f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and "End_Timestamp" in r and r["Counter_Name"] == "SQ_BUSY_CYCLES":
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, c in acc.items():
        n = len(c.get("SQ_BUSY_CYCLES", [])) or 1
        mean = {name: sum(v) / len(v) for name, v in c.items()}
        ns = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
        row = {"launches_sampled": n, "avg_ns_under_pmc": ns, **{name: round(v, 1) for name, v in sorted(mean.items())}}
        mf = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if ns and mf:
            # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (checked: value / duration = 8 x the shader clock)
            wall = mean["GRBM_GUI_ACTIVE"] / 8.0 if mean.get("GRBM_GUI_ACTIVE") else ns * 2.4
            row["wall_cycles"] = round(wall, 1)
            row["effective_clock_ghz"] = round(wall / ns, 3)
            row["mfma_busy_frac_of_all_simds"] = round(mf / (wall * SIMDS), 4)
        if mean.get("SQ_BUSY_CYCLES"):
            row["mfma_busy_over_sq_busy"] = round(mf / mean["SQ_BUSY_CYCLES"], 4)
        out[k] = row
    json.dump(dict(sorted(out.items(), key=lambda kv: -(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["launches_sampled"]))),
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
