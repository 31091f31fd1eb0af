#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv ...
    python tools/mfma_util.py <counter_collection.csv> <kernel_trace.csv> out.csv

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-core busy cycles summed over the chip's SIMDs (32 per v_mfma_f32_32x32x16_bf16 and
per v_mfma_f32_16x16x4_f32, 64 per v_mfma_f32_32x32x2_f32: MI355X_MICROARCH.md).  Utilisation = busy cycles / (1024 SIMDs x the
dispatch's TIME x 2.4 GHz): the denominator is the nominal clock, because a per-dispatch cycle count is not available for
dispatches this short -- round 2 derived one from GRBM_GUI_ACTIVE / 8 and got 3.1-5.3 "GHz" (the counter does not divide by
8 on dispatches of a few microseconds: VERDICT r2), so that column is gone.  The chip holds MFMA-dense loops at 1.5-1.7 GHz
(guide, 'DVFS give-back'), so a kernel that keeps its matrix cores busy 100 % of the time reads ~0.65-0.7 here; for the
row-local chain kernels, which run on 16-20 of the 256 CUs, the last column rescales to the CUs the grid occupies.
Medians over the recorded dispatches; durations are those of the counter run itself."""
import collections
import csv
import statistics
import sys


def main():
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur, grid = collections.defaultdict(list), {}
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        try:
            grid[r["Kernel_Name"]] = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        except (KeyError, ValueError):
            pass
    rows = []
    for k, c in cnt.items():
        busy = statistics.median(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
        insts = statistics.median(c.get("SQ_INSTS_MFMA", [0.0]))
        if insts <= 0:
            continue
        us = statistics.median(dur[k]) if k in dur else float("nan")
        wgs = grid.get(k, 0)
        cus = min(wgs, 256) if wgs else 256
        util = busy / (1024.0 * us * 2400.0) if us == us and us > 0 else 0.0
        rows.append([k, len(c.get("SQ_INSTS_MFMA", [])), us, insts, busy, busy / insts, wgs, util, util * 256.0 / cus])
    rows.sort(key=lambda r: -r[4])
    with open(sys.argv[3], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Dispatches", "us_median(counter run)", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                    "busy_cycles_per_mfma", "workgroups", "mfma_util_chip(busy/(1024 SIMDs*us*2400))",
                    "mfma_util_of_occupied_CUs(x 256/min(workgroups,256))"])
        w.writerows(rows)
    for r in rows[:14]:
        print("%-64s %6.1f us  %8.0f mfma  %.0f cyc each  util chip %.3f, occupied CUs %.3f (%d wg)" %
              (r[0][:64], r[2], r[3], r[5], r[7], r[8], r[6]))


if __name__ == "__main__":
    main()
