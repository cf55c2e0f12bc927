#!/bin/bash
# Round 5: config 5 (10 x 512 residual, bf16) with the one-launch prologue ((NPH, NPZ) = (4, 1) instance) against the launches per layer
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/cfg5
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
A="--model deep_residual --precision bf16 --steps 300 --warmup 20 --no-cpu-baseline --no-extras"
run new_s2 $A
MSIREN_PROLOGUE_F16X3=0 run old_s2 $A
run new_s1 --streams 1 $A
MSIREN_PROLOGUE_F16X3=0 run old_s1 --streams 1 $A
run new_s2_8 --slices 8 --model deep_residual --precision bf16 --steps 60 --warmup 5 --no-cpu-baseline --no-extras
MSIREN_PROLOGUE_F16X3=0 run old_s2_8 --slices 8 --model deep_residual --precision bf16 --steps 60 --warmup 5 --no-cpu-baseline --no-extras
run morlet_s2 --activation morlet --steps 600 --warmup 30 --no-cpu-baseline --no-extras
run morlet_s1 --activation morlet --streams 1 --steps 600 --warmup 30 --no-cpu-baseline --no-extras
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone') or {}
        print(f.split('/')[-1].ljust(18), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), 'alone', round(a.get('frac',0),3), [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --streams 1 $A > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_s1.csv; cut -c1-150 $out/kernel_stats_s1.csv | head -6
