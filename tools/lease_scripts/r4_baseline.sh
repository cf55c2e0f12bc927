#!/bin/bash
# Round-4 baseline on this round's first box: kernel stats of the 64-slice single call (where the prologue goes at
# throughput sizes), 8-slice and default lines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/base
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; }
run bench_default --no-cpu-baseline
run bench_strong64 --total-slices 64 --steps 40 --warmup 5 --no-cpu-baseline --no-extras
run bench_strong64_s1 --total-slices 64 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras
run bench_slices8 --slices 8 --no-cpu-baseline --no-extras
run bench_slices8_s1 --slices 8 --streams 1 --no-cpu-baseline --no-extras
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof64 -- python3 bench.py --total-slices 64 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/prof64.json 2> $out/prof64.err
f=$(find $out/prof64 -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_strong64_s1.csv; rm -rf $out/prof64
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof8 -- python3 bench.py --slices 8 --streams 1 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > $out/prof8.json 2> $out/prof8.err
f=$(find $out/prof8 -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_slices8_s1.csv; rm -rf $out/prof8
cut -c1-200 $out/kernel_stats_strong64_s1.csv | head -14
cut -c1-200 $out/kernel_stats_slices8_s1.csv | head -14
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/base/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; t=d.get('roofline_timed_mode',{})
        print(f.split('/')[-1], round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), 'timed', round(t.get('frac',0),3))
    except Exception as e: print(f, 'ERR', e)
PY
