#!/bin/bash
# Runs ON the GPU box (gpurun), after tools/refresh_profiles.sh: the round's A/B bench lines (one box, one clock state), the
# in-kernel timeline of the forward depth projection and the two stand-alone probes -> gpurun_out/extras/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/extras
rm -rf $O; mkdir -p $O
cd $R
# launch-fusion A/B lines of this round (same box, same clocks)
for f in "--no-pair-embeddings" "--no-ride-planes" "--chain-fp32" "--no-decoder-chain" "--no-fuser-chain" "--overlap-planes" "--overlap-param-tail" "--no-fuser-chain --no-decoder-chain"; do
  echo "== bench.py $f" >> $O/ab_lines.txt
  timeout -k 10 200 python3 $R/bench.py --steps 400 --no-cpu-baseline $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab_lines.txt
done
echo "== bench.py (default)" >> $O/ab_lines.txt
timeout -k 10 200 python3 $R/bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/ab_lines.txt
# in-kernel timeline of the forward depth projection (builds a probe variant, restores the product build)
bash $R/tools/nt_timeline.sh > $O/nt_timeline.txt 2>&1
mkdir -p $R/tools/_build
[ -x $R/tools/_build/nt_stream_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $R/tools/_build/nt_stream_probe $R/tools/nt_stream_probe.hip
[ -x $R/tools/_build/split_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/r3d_amd/csrc -I $R/include -o $R/tools/_build/split_probe $R/tools/split_probe.hip
timeout -k 10 60 $R/tools/_build/nt_stream_probe > $O/nt_stream_probe.txt 2>&1
timeout -k 10 60 $R/tools/_build/split_probe > $O/split_probe.txt 2>&1
ls $O
