#!/bin/bash
# Round 5, first lease: starting lines of the tree as round 4 left it, and -- with no code change -- what the large-call
# split of round 4 does when its threshold is lowered to ONE slice (MSIREN_SPLIT_MIN=300) on a one-stream handle,
# for several shares of the first part; host numpy -> numpy latency beside it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/run1
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run s1_base --streams 1 --steps 600 --warmup 30 --no-cpu-baseline --no-extras
for pct in 8 12 14 18 24; do MSIREN_SPLIT_MIN=300 MSIREN_SPLIT_PCT=$pct run s1_split_p$pct --streams 1 --steps 600 --warmup 30 --no-cpu-baseline --no-extras; done
run s2_base --steps 600 --warmup 30 --no-cpu-baseline --no-extras
run strong64_s1 --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/run1/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3),
              '| alone', a.get('kernel'), round(a.get('frac',0),3), '|', [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep.txt 2>&1; grep -v amdgpu.ids $out/latency_sweep.txt
MSIREN_SPLIT_MIN=300 MSIREN_SPLIT_PCT=14 timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep_split.txt 2>&1; grep -v amdgpu.ids $out/latency_sweep_split.txt
