#!/bin/bash
# x1w as the default config-5 trunk: full GPU suite, then the stand-alone config-5 lines against the register-resident kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w2
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  run x1w_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
  MSIREN_X1_WS=0 run x1n_r$r --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
done
run x1w_s1 --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
MSIREN_X1_WS=0 run x1n_s1 --model deep_residual --precision bf16 --no-cpu-baseline --streams 1 --steps 300
run x1w_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
MSIREN_X1_WS=0 run x1n_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
run x1w_8 --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60
MSIREN_X1_WS=0 run x1n_8 --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60
run x1w_64 --model deep_residual --precision bf16 --no-cpu-baseline --total-slices 64 --steps 8 --warmup 2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w2/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(16), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), '| alone', a.get('kernel'), round(a.get('frac',0),3), round(a.get('avg_launch_ms',0),4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
