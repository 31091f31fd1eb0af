#!/bin/bash
# Runs ON the GPU box: builds with the timeline marks, prints the timelines, restores the product build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
R3D_EXTRA_DEFS="-DR3D_NT_PROBE=16" python -m r3d_amd.build > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python tools/nt_timeline.py 8 61 2>/dev/null
python tools/nt_timeline.py 9 196 2>/dev/null
python tools/nt_timeline.py 11 196 2>/dev/null
python tools/tn_timeline.py 2>/dev/null
python -m r3d_amd.build --force > /dev/null 2>&1
