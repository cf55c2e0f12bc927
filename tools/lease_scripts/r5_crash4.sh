#!/bin/bash
# Round 5: after the fix (only whole pages inside the caller's tiles are registered; outputs in place only where the caller's memory is page-locked):
# the diagnostic run that failed (no capture, glibc heap checks: arrays from the brk heap), then the plain suite, then the host-call latencies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/crash
mkdir -p $out
LIBC_FATAL_STDERR_=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $out/nocapture_fixed.log 2>&1
rc=$?; echo "brk-heap run rc=$rc"; grep -n -i "passed\|failed\|Memory access fault\|Aborted\|Error" $out/nocapture_fixed.log | cut -c1-220 | head
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $out/plain_fixed.log 2>&1
rc=$?; echo "plain run rc=$rc"; grep -n -i "passed\|failed\|Memory access fault\|Aborted" $out/plain_fixed.log | cut -c1-220 | head
[ $rc -ne 0 ] && exit $rc
python3 tools/latency.py 2>&1 | grep -v amdgpu | tail -3
python3 tools/host_reconstruct_ab.py 2>&1 | grep -v amdgpu | grep "default\|device call + sync  "
