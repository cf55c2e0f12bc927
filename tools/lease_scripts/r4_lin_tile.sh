#!/bin/bash
# prologue at one slice per call: the 32x32-tile Linear kernel from fewer rows (MSIREN_LINEAR_TILE_MIN), config 5 and config 2, one stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/lin_tile
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for t in 1024 256 1; do
  export MSIREN_LINEAR_TILE_MIN=$t
  run c5_1s_t$t --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
  run c5_2s_t$t --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --steps 300
  run c2_1s_t$t --no-cpu-baseline --no-extras --streams 1 --steps 600
  run c2_2s_t$t --no-cpu-baseline --no-extras --steps 600
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/lin_tile/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', round(r['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
