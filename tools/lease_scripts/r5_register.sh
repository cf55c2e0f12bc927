#!/bin/bash
# Round 5: pageable caller buffers page-locked for the duration of a one-chunk host call (then read / written in place): tests, latency A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/register
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_cabi.py tests/test_c_consumer.py -x -q 2>&1 | tail -3
for reg in 1 0; do echo "MSIREN_HOST_REGISTER=$reg"; MSIREN_HOST_REGISTER=$reg timeout -k 10 200 python3 tools/latency_sweep.py 2>&1 | grep -v amdgpu.ids | tail -6; done
MSIREN_TRACE_HOST=1 timeout -k 10 100 python3 tools/host_trace.py 50 2>&1 | grep -v amdgpu | tail -3
