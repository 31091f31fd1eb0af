#!/bin/bash
# Runs ON the GPU box: the forward depth projection (bf16x3 NT kernel) at the headline shape with parts stubbed out
# (R3D_NT_PROBE bits, csrc/gemm_bf3.hip) -- results are wrong by construction, only the durations are read.
#   tools/nt_stub_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for v in 0 1 8 2 4 6 5 7; do
  R3D_EXTRA_DEFS="-DR3D_NT_PROBE=$v" python -m r3d_amd.build > $O/build_$v.log 2>&1 || { echo "build $v failed"; tail -3 $O/build_$v.log; continue; }
  timeout -k 10 120 python tools/nt_probe.py 128 128 50176 2>/dev/null | grep -E "^\{\}|'splitk': 196" | sed "s/^/probe $v: /" | cut -c1-150
done
python -m r3d_amd.build --force > /dev/null 2>&1
