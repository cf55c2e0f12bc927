#!/bin/bash
# Round 5: rocprofv3 kernel stats of the large one-stream calls (uncut since MSIREN_SPLIT_MIN defaults to 0): 64 and 8 slices per call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/stats64
rm -rf $out && mkdir -p $out
stats() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/bench_under_rocprof_$name.json 2> $out/prof_$name.err
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$name.csv; rm -rf $out/prof_$name; echo "kernel stats ($name): done"; }
stats strong64_streams1 --total-slices 64 --streams 1 --steps 30 --warmup 3
stats slices8_streams1 --slices 8 --streams 1 --steps 100 --warmup 10
for f in $out/kernel_stats_*.csv; do echo $f; head -7 $f | cut -c1-170; done
