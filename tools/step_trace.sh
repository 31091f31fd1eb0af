#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of the default bench, prints the launches of one step (duration, gap to the
# previous kernel, name, grid) into gpurun_out/$1/one_step.txt.   tools/step_trace.sh <outdir> [bench flags...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline "$@" > $O/stats.log 2>&1; echo "trace rc=$?"
cd $O
cp $(find stats -name "*kernel_stats.csv" | head -1) kernel_stats.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("stats/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_dropout_fin" in r["Kernel_Name"]]      # (the step's variant; bench.py's
if len(idx) < 3:                                                                        #  roofline timing loop runs the plain one)
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev = None
with open("one_step.txt", "w") as out:
    tot = 0.0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) if prev else 0
        tot += (e - s) / 1e3
        out.write("%6.2f us  gap %5.2f  %s grid=%s wg=%s\n" % ((e - s) / 1e3, gap / 1e3, r["Kernel_Name"][:70],
                                                              r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")))
        prev = e
    out.write("launches %d, sum of durations %.1f us\n" % (b - a, tot))
PY
find . -name "*kernel_trace.csv" -delete
cat one_step.txt
