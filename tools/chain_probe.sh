#!/bin/bash
# Runs ON the GPU box: stage timelines of the forward chain kernel with parts of its N = 512 stage stubbed out
# (R3D_FC_PROBE bits: 1 no u / f1 global stores, 2 no LDS write of f1, 4 no erf, 8 no weight loads in the stage, 16 no LDS
# staging of the weights) -- results are wrong by construction, only the timeline is read.  tools/chain_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for v in 0 1 3 7 8 16 24 31; do
  R3D_EXTRA_DEFS="-DR3D_FC_PROBE=$v" python -m r3d_amd.build > $O/build_$v.log 2>&1 || { echo "build $v failed"; tail -3 $O/build_$v.log; continue; }
  timeout -k 10 120 python tools/chain_timeline.py --graph 2>/dev/null | grep -A1 "fwd_chain" | tail -1 | sed "s/^/probe $v: /"
done
python -m r3d_amd.build --force > /dev/null 2>&1
