#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (one pass per counter) into the per-kernel HBM traffic table that
bench.py reads for roofline.traffic:  python tools/pmc_summary.py FETCH.csv WRITE.csv out.json out.csv

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are
in KiB; FETCH_SIZE tallies 128-byte requests as 64 bytes for 16-byte-per-lane streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Per launch = median over the recorded dispatches."""
import collections
import csv
import json
import statistics
import sys


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out, rows = {}, []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(sum(fetch.get(k, [0])) + sum(write.get(k, [0])))):
        f = statistics.median(fetch[k]) if k in fetch else 0.0
        w = statistics.median(write[k]) if k in write else 0.0
        rd, wr = 2.0 * f * 1024.0, w * 1024.0
        out[k] = dict(launches=len(fetch.get(k, write.get(k, []))), fetch_size_kib=f, write_size_kib=w,
                      hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr)
        rows.append([k, out[k]["launches"], f, w, rd, wr, rd + wr])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    with open(sys.argv[4], "w", newline="") as fh:
        cw = csv.writer(fh)
        cw.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KiB_median", "WRITE_SIZE_KiB_median",
                     "HBM_read_bytes(2xFETCH)", "HBM_write_bytes", "HBM_bytes_per_launch"])
        cw.writerows(rows)


if __name__ == "__main__":
    main()
