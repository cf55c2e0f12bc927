#!/bin/bash
# same-box A/B of two builds of libmsiren.so through MSIREN_LIB: usage  r3_ab_lib.sh <name> <lib> [<name> <lib> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/ab
mkdir -p $out
for round in 1 2; do
  set -- "$@"
  args=("$@")
  i=0
  while [ $i -lt ${#args[@]} ]; do
    name=${args[$i]}; lib=${args[$((i+1))]}; i=$((i+2))
    MSIREN_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 bench.py --streams 1 --no-cpu-baseline --no-extras --check --steps 600 --warmup 100 > $out/${name}_s1_r$round.json 2> $out/${name}.err || echo "$name failed"
    MSIREN_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 bench.py --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 100 --warmup 20 > $out/${name}_sl8_r$round.json 2>> $out/${name}.err || echo "$name failed"
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value'],1),'Mpx/s', r['kernel'], round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f,'ERR',e)
PY
