#!/bin/bash
# Runs ON the GPU box: round-2 erank / AdamW measurements (plain, rocprofv3 kernel stats, and the two PMC passes).
#   tools/r02_measure.sh <tag> [erank|adamw ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-m1}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  timeout -k 10 240 python3 $R/tools/r02_profile.py $W --out $O/$W.json > $O/$W.log 2>&1 || { echo "$W plain failed"; tail -5 $O/$W.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_stats -o k -- python3 $R/tools/r02_profile.py $W > $O/${W}_stats.log 2>&1 || { echo "$W stats failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${W}_fetch -o f -- python3 $R/tools/r02_profile.py $W > $O/${W}_fetch.log 2>&1 || { echo "$W fetch failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${W}_write -o w -- python3 $R/tools/r02_profile.py $W > $O/${W}_write.log 2>&1 || { echo "$W write failed"; exit 1; }
  F=$(find $O/${W}_fetch -name "*counter_collection.csv" | head -1); WW=$(find $O/${W}_write -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py "$F" "$WW" $O/${W}_pmc_hbm.json $O/${W}_pmc_hbm.csv
  cp $(find $O/${W}_stats -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
  cat $O/$W.log
done
