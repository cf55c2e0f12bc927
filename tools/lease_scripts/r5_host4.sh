#!/bin/bash
# Round 5: streamed upload of the host-pointer call (conv kernel per upload piece): tests, latency by piece size, timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/${OUT:-host4}
rm -rf $out && mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_prologue.py tests/test_cabi.py tests/test_c_consumer.py -x -q 2>&1 | tail -2
for up in 0 50 100 134 200; do
  echo "MSIREN_HOST_UP=$up"; MSIREN_HOST_UP=$up timeout -k 10 120 python3 tools/latency_sweep.py 2>&1 | grep -v amdgpu.ids | tail -3
done
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tl -- python3 tools/host_call_timeline.py run > $out/run.log 2>&1; python3 tools/host_call_timeline.py show $out/tl | tee $out/timeline.txt | tail -14
B=3200 timeout -k 10 250 python3 tools/host_plan_ab.py 2>&1 | grep -v amdgpu.ids | head -4
