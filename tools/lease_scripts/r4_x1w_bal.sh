#!/bin/bash
# config 5, two streams: how many CUs should the trunk leave to the neighbour stream?  MSIREN_X1_GRID sweep, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w_bal2
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for g in 256 252 248 244 240 232 225; do
  MSIREN_X1_GRID=$g run g${g}_2s --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --steps 300
done
run auto_1s --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
MSIREN_X1_BALANCE=0 run all_1s --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
MSIREN_X1_GRID=256 run g256_2s_again --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --steps 300
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w_bal2/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']; a=d.get('roofline_kernel_alone') or {}
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', round(r['frac'],4), '| alone', round(a.get('frac',0),4), round(a.get('avg_launch_ms',0),4))
    except Exception as e: print(f, 'ERR', e)
PY
