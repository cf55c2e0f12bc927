#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 200 python3 tools/timeline_ws_slot.py 8 > gpurun_out/r4/timeline_ws_slot_8.txt 2>gpurun_out/r4/timeline_ws_slot_8.err; echo "rc=$?"; cat gpurun_out/r4/timeline_ws_slot_8.txt; tail -3 gpurun_out/r4/timeline_ws_slot_8.err
