#!/bin/bash
# Round 5: config 5 on three streams by the trunk's grid cap (MSIREN_X1_GRID): with the prologue off the critical path, do full rounds on fewer CUs pack better?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/c5grid
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --model deep_residual --precision bf16 --steps 300 --warmup 20 "$@" --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run g256_s3 --streams 3
for g in 250 240 232 225 200 150 128; do MSIREN_X1_GRID=$g run g${g}_s3 --streams 3; done
MSIREN_X1_GRID=225 run g225_s2 --streams 2
run g256_s3_again --streams 3
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3))
    except Exception as e: print(f, 'ERR', e)
PY
