#!/bin/bash
# Round 5 experiment: the weight-stationary trunk on a three-stream handle (MSIREN_WS_TWO=1): back-to-back launches of the faster kernel,
# the prologues of the calls after next in the tails
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/ws3
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run n_s2 --streams 2
MSIREN_WS_TWO=1 run w_s2 --streams 2
MSIREN_WS_TWO=1 run w_s3 --streams 3
run n_s3 --streams 3
MSIREN_WS_TWO=1 MSIREN_GRID=248 run w_s3_g248 --streams 3
MSIREN_WS_TWO=1 MSIREN_GRID=240 run w_s3_g240 --streams 3
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(18), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), [(k['kernel'][:28], k['launches'], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
