#!/bin/bash
# Round 5: synchronous one-chunk host calls look at the domain flag on the host instead of queueing the conditional launch (MSIREN_HOST_CHECK=0: as before)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/hostcheck
rm -rf $out && mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_ws.py -q -x > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
  echo "host check:"; python3 tools/latency.py 2>&1 | grep -v amdgpu | tail -3
  echo "conditional launch (MSIREN_HOST_CHECK=0):"; MSIREN_HOST_CHECK=0 python3 tools/latency.py 2>&1 | grep -v amdgpu | tail -3
done | tee $out/latency_ab.txt
