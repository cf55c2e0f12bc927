#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 200 python3 tools/experiments/x1w_trunk_packing.py > gpurun_out/r4/x1w_trunk_packing.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/x1w_trunk_packing.txt
