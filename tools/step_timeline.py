#!/usr/bin/env python3
"""Prints the kernel timeline of one steady-state step from a rocprofv3 rocpd database (kernel-trace run of bench.py):
start offset, duration, queue, gap to the previous kernel's end."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "adamw_dropout" in r[0]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0, prev_end = rows[a][2], rows[a][2]
for r in rows[a + 1:b + 1]:
    print(f"{(r[1] - t0) / 1e3:8.1f} +{(r[1] - prev_end) / 1e3:6.1f} {(r[2] - r[1]) / 1e3:7.1f} q{r[3]} {r[0][:70]}")
    prev_end = max(prev_end, r[2])
print("step", (rows[b][2] - rows[a][2]) / 1e3, "us; kernels", b - a)
