#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 500 python3 tools/soak.py ${SECS:-90} > gpurun_out/r4/soak.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r4/soak.txt | tail -12
