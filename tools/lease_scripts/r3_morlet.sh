#!/bin/bash
# Morlet (BASELINE configs[3]): register-resident vs weight-stationary trunk, single stream, same box; default two-stream line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/morlet
rm -rf $out && mkdir -p $out
python3 -m pytest tests/test_gpu_ws.py -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
for ws in 0 1; do
  MSIREN_F16_WS=$ws timeout -k 10 300 python3 bench.py --activation morlet --streams 1 --no-cpu-baseline --no-extras --check --steps 600 --warmup 100 > $out/bench_morlet_ws${ws}_s1.json 2> $out/err_$ws.txt || echo "ws=$ws failed"
done
timeout -k 10 300 python3 bench.py --activation morlet --no-cpu-baseline --check > $out/bench_morlet_default.json 2> $out/err_d.txt || echo failed
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/morlet/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value'],1),'Mpx/s', round(d['ms_per_step'],4),'ms;', r['kernel'], round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f,'ERR',e)
PY
