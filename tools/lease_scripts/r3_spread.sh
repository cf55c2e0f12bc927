#!/bin/bash
# Spread B-fragment reads (one per gap) vs the four in a bunch at a region's start: bit-identity suite, then same-box A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3/ab && rm -f gpurun_out/r3/ab/*
timeout -k 10 600 python3 -m pytest tests/test_gpu_ws.py -x -q > gpurun_out/r3/ab/pytest_ws.log 2>&1; rc=$?; tail -3 gpurun_out/r3/ab/pytest_ws.log
[ $rc -eq 0 ] || exit $rc
bash tools/r3_ab_lib.sh spread mri_inr_amd/libmsiren.so bunched build_abl/libmsiren_bunched.so
