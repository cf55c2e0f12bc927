#!/bin/bash
# Round 4: config 5's trunk on 16x16x32 tiles (siren_trunk_x1n.hip.h) against round 1's 32x32x16 kernel (MSIREN_X1_TILE=32), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1n
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "config5 or residual or deep_model" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  MSIREN_X1_TILE=32 run x1_tile32_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
  run x1n_tile16_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
  MSIREN_X1_TILE=32 run x1_tile32_s1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
  run x1n_tile16_s1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
done
run x1n_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
MSIREN_X1_TILE=32 run x1_f16_tile32 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1n/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(24), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), '| alone', a.get('kernel'), round(a.get('frac',0),3), round(a.get('avg_launch_ms',0),4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
