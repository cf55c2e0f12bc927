#!/bin/bash
# Round 5 experiment (reverted; patch under tools/experiments): the conditional launch of one-stream msiren_forward_tiles_dev calls on a side stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/condaside
rm -rf $out && mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_ws.py -q -x > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -ne 0 ] && exit 1
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for rep in 1 2; do
run aside_s1_$rep --streams 1
MSIREN_COND_ASIDE=0 run own_s1_$rep --streams 1
done
run aside_morlet_s1 --streams 1 --activation morlet
MSIREN_COND_ASIDE=0 run own_morlet_s1 --streams 1 --activation morlet
run aside_8_s1 --streams 1 --slices 8 --steps 200 --warmup 20
MSIREN_COND_ASIDE=0 run own_8_s1 --streams 1 --slices 8 --steps 200 --warmup 20
run aside_s2 --streams 2
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3))
    except Exception as e: print(f, 'ERR', e)
PY
