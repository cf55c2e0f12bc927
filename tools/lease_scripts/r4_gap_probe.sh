#!/bin/bash
# what the weight-stationary trunk's slot costs beyond its MFMA stream (tools/mfma_gap_probe.hip, tools/mfma_slot_probe.hip, built into build_abl/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
for p in ${PROBES:-slot_probe}; do
  timeout -k 10 120 build_abl/$p > gpurun_out/r4/$p.txt 2>&1; echo "$p rc=$?"; cat gpurun_out/r4/$p.txt
done
