#!/bin/bash
# Round 5: windows of one array from two threads, buffers page-locked in part (bounce buffer), then the soak (stream modes 1-3) with its host-call phase
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/hostsafety
rm -rf $out && mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "page_locked_in_part or windows_of_one_array or two_threads" > $out/pytest.log 2>&1; rc=$?; tail -15 $out/pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 python3 -m pytest tests/test_gpu_split.py -q -x > $out/pytest_split.log 2>&1; rc=$?; tail -3 $out/pytest_split.log
[ $rc -eq 0 ] && timeout -k 10 500 python3 tools/soak.py ${SECS:-60} > $out/soak.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids $out/soak.txt | tail -14
