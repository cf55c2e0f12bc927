#!/bin/bash
# kernel-level timing of the slice pipeline (bench.py --pipeline reconstruct); args are passed to bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/recon_prof
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --pipeline reconstruct --steps 100 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | head -14
python3 -c "import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
