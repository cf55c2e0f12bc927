#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 200 python3 tools/latency_sweep.py > gpurun_out/r4/latency_sweep.txt 2>&1; cat gpurun_out/r4/latency_sweep.txt | grep -v amdgpu.ids
