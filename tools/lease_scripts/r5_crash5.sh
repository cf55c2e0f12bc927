#!/bin/bash
# Round 5: per-call page-locking of caller memory OFF by default: the plain suite twice (capture off so that a runtime message would show), host-call latencies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/crash
mkdir -p $out
for k in 1 2; do
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $out/plain_noreg_$k.log 2>&1
rc=$?; echo "plain run $k rc=$rc"; grep -n -i "passed\|failed\|Memory access fault\|Aborted" $out/plain_noreg_$k.log | cut -c1-220 | head -4
[ $rc -ne 0 ] && exit $rc
done
python3 tools/latency.py 2>&1 | grep -v amdgpu | tail -3
python3 tools/host_pinned_ab.py 2>&1 | grep -v amdgpu | grep "B= 400"
python3 tools/host_reconstruct_ab.py 2>&1 | grep -v amdgpu | grep "default\|device call + sync  "
