#!/bin/bash
# Round 5: the synchronous host-pointer call pipelining itself (chunk 0 small, copies beside the other chunk's kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/${OUT:-host2}
rm -rf $out && mkdir -p $out
timeout -k 10 120 python3 tools/host_trace.py 50 > $out/pipelined.txt 2> $out/pipelined.err; cat $out/pipelined.txt; grep msiren_forward $out/pipelined.err | tail -3
MSIREN_HOST_CHUNKS=1 timeout -k 10 120 python3 tools/host_trace.py 50 > $out/chunks1.txt 2> $out/chunks1.err; cat $out/chunks1.txt; grep msiren_forward $out/chunks1.err | tail -1
timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep.txt 2>&1; grep -v amdgpu.ids $out/latency_sweep.txt | tail -5
timeout -k 10 300 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_prologue.py -x -q 2>&1 | tail -3
