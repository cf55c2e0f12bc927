#!/bin/bash
# Round 3, item 1: the multi-rank tests (blob export/import, stub-RCCL two-rank receive path, bench rehearsal lines).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/multi
rm -rf $out && mkdir -p $out
timeout -k 10 800 python3 -m pytest tests/test_gpu_multi.py -m gpu -x -q > $out/pytest_multi.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_multi.log
