#!/bin/bash
# Runs ON the GPU box (gpurun): the round's judged artefacts into gpurun_out/final/ -- GPU test log, smoke, the default
# bench line (with cpu_baseline + erank field), rocprofv3 kernel stats of the same command, the two PMC passes (separate
# runs, no trace domains besides kernel-trace) that tools/pmc_summary.py turns into the HBM-traffic table, and the
# effective-rank / AdamW harness (tools/r02_profile.py) plain, under --stats and under the two PMC passes.
#   tools/refresh_profiles.sh [notests]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
if [ "$1" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_gpu.log
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1; echo "write rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py "$F" "$W" $O/pmc_hbm.json $O/pmc_hbm.csv && echo "pmc ok"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for W in erank adamw; do
  timeout -k 10 240 python3 $R/tools/r02_profile.py $W --out $O/$W.json > $O/$W.log 2>&1; echo "$W rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_stats -o k -- python3 $R/tools/r02_profile.py $W > $O/${W}_stats.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${W}_fetch -o f -- python3 $R/tools/r02_profile.py $W > $O/${W}_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${W}_write -o w -- python3 $R/tools/r02_profile.py $W > $O/${W}_write.log 2>&1
  F=$(find $O/${W}_fetch -name "*counter_collection.csv" | head -1); WW=$(find $O/${W}_write -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py "$F" "$WW" $O/${W}_pmc_hbm.json $O/${W}_pmc_hbm.csv
  cp $(find $O/${W}_stats -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv
done
for c in cfg3 cfg4 cfg5; do timeout -k 10 200 python3 $R/bench.py --config $c --steps 50 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null; cut -c1-200 $O/bench_$c.json; done
timeout -k 10 200 python3 $R/bench.py --steps 200 --no-cpu-baseline --erank-weight 0.05 > $O/bench_erank_in_step.json 2>/dev/null; cut -c1-200 $O/bench_erank_in_step.json
timeout -k 10 300 python3 $R/bench.py --config cfg4 --steps 50 --no-cpu-baseline --erank-weight 0.05 > $O/bench_cfg4_erank_in_step.json 2>/dev/null; cut -c1-200 $O/bench_cfg4_erank_in_step.json
timeout -k 10 200 python3 $R/tools/chain_timeline.py --graph > $O/chain_timeline.txt 2>&1
# matrix-core utilisation of every MFMA kernel of the step: one PMC pass (no trace domains besides kernel-trace)
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
MC=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1); MT=$(find $O/pmc_mfma -name "*kernel_trace.csv" | head -1)
python3 $R/tools/mfma_util.py "$MC" "$MT" $O/mfma_util.csv > $O/mfma_util.log 2>&1; tail -12 $O/mfma_util.log
bash $R/tools/step_trace.sh final/trace > $O/trace.log 2>&1; cp $O/trace/one_step.txt $O/step_trace.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O | head -50
