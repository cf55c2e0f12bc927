#!/bin/bash
# full GPU suite + the default bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/suite
rm -rf $out && mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; }
run bench_default --no-cpu-baseline
run bench_streams1 --streams 1 --no-cpu-baseline
run bench_slices8_s1 --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 200 --warmup 20
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/suite/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value'],1),'Mpx/s', round(d['ms_per_step'],4),'ms;', r['kernel'], round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), d.get('extra'))
    except Exception as e: print(f,'ERR',e)
PY
python3 tools/latency.py 2>&1 | tail -4
