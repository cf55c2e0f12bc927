#!/bin/bash
# config 5, same box: the slot-to-slot prefetch build (bias rows / first B fragments / first epilogue loads fetched during the slot
# before) against the build without it (build_abl/libmsiren_x1w_nopf.so), one stream, kernel alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w_pf
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2 3; do
  run pf_1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_x1w_nopf.so run nopf_1_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --steps 300
  run pf_8_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --slices 8 --steps 60
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_x1w_nopf.so run nopf_8_r$r --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --streams 1 --slices 8 --steps 60
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/x1w_pf/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(20), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],4), round(r['avg_launch_ms'],4))
    except Exception as e: print(f, 'ERR', e)
PY
