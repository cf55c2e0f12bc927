#!/bin/bash
# weight-stationary trunk: parity tests, then A/B against the register-resident kernel in one process each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/ws
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ws.py -m gpu -x -q > $out/pytest_ws.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $out/pytest_ws.log
[ $rc -ne 0 ] && exit $rc
for ws in 0 1; do
  MSIREN_F16_WS=$ws timeout -k 10 300 python3 bench.py --streams 1 --no-cpu-baseline --no-extras --check --steps 600 --warmup 100 > $out/bench_ws${ws}_s1.json 2> $out/bench_ws${ws}_s1.err || echo "ws=$ws failed"
  MSIREN_F16_WS=$ws timeout -k 10 300 python3 bench.py --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 100 --warmup 20 > $out/bench_ws${ws}_s1_sl8.json 2> $out/bench_ws${ws}_s1_sl8.err || echo "ws=$ws failed"
  MSIREN_F16_WS=$ws timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 1000 --warmup 100 > $out/bench_ws${ws}_s2.json 2> $out/bench_ws${ws}_s2.err || echo "ws=$ws failed"
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/ws/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value'],1),'Mpx/s', round(d['ms_per_step'],4),'ms trunk', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f,'ERR',e)
PY
