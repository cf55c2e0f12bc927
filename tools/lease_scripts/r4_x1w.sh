#!/bin/bash
# Round 4: config 5's trunk weight-stationary (siren_trunk_x1w.hip.h) against the register-resident one (MSIREN_X1_WS=0), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "config5 or residual or deep_model" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  MSIREN_X1_WS=0 run x1n_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
  run x1w_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
  MSIREN_X1_WS=0 run x1n_s1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
  run x1w_s1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
done
run x1w_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
MSIREN_X1_WS=0 run x1n_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
run x1w_8slices --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60
MSIREN_X1_WS=0 run x1n_8slices --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(24), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), '| alone', a.get('kernel'), round(a.get('frac',0),3), round(a.get('avg_launch_ms',0),4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
