#!/bin/bash
# x1w: correctness of the overlapped build, then timing of ablation builds (results wrong there) at 8 slices per launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w_abl
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "config5 or residual or deep_model" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run x1w_8 --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60 --streams 1 --check
run x1w_1 --model deep_residual --precision bf16 --no-cpu-baseline --steps 300 --streams 1
MSIREN_X1_WS=0 run x1n_8 --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60 --streams 1
for a in 1 3 4; do MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_x1w_abl$a.so run abl${a}_8 --model deep_residual --precision bf16 --no-cpu-baseline --slices 8 --steps 60 --streams 1; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w_abl/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(16), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), round(r['avg_launch_ms'],4), 'check', d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
