#!/bin/bash
# headline (two streams, register-resident trunk): persistent grid cap sweep (MSIREN_GRID), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/grid_n
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for g in 256 240 225 200 256; do
  MSIREN_GRID=$g run g${g}_$RANDOM --no-cpu-baseline --no-extras --steps 1000
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r4/grid_n/*.json'), key=os.path.getmtime):
    d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(18), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', round(r['frac'],4))
PY
