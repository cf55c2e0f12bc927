#!/bin/bash
# Round 5: accuracy check of the split-fp16 prologue + rocprofv3 kernel stats of the one-stream and two-stream commands with it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/run3
rm -rf $out && mkdir -p $out
timeout -k 10 300 python3 tools/prologue_check.py > $out/check_new.txt 2>&1; echo "check rc=$?"; grep -v amdgpu.ids $out/check_new.txt | tail -25
MSIREN_PROLOGUE_F16X3=0 timeout -k 10 300 python3 tools/prologue_check.py > $out/check_old.txt 2>&1; grep -v amdgpu.ids $out/check_old.txt | tail -22
for s in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_s$s -- python3 bench.py --streams $s --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/trace_s$s.log 2>&1
  f=$(find $out/trace_s$s -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_s$s.csv; cut -c1-150 $out/kernel_stats_s$s.csv | head -8
done
