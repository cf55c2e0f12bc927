#!/bin/bash
# config 5: does the weight stream's L2 behaviour cost time?  shipped build against -DMSIREN_X1W_ABL=8 (every layer reads layer 1's
# 512 KB: an L2-resident stream; results wrong, timing only), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w_l2
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  run shipped_1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_x1w_abl8.so run abl8_1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
  run shipped_8_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --slices 8 --steps 60
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_x1w_abl8.so run abl8_8_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --slices 8 --steps 60
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w_l2/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(20), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), round(r['avg_launch_ms'],4))
    except Exception as e: print(f, 'ERR', e)
PY
