#!/bin/bash
# kernel-level timing of one single-stream bench run (encoder / modulator / trunk split)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/stage_prof
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --streams 1 --steps 100 --no-cpu-baseline > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cut -d, -f1-6 "$f" | head -20
