#!/bin/bash
# Morlet on the weight-stationary trunk: the ten instructions of sin * exp2 spread over seven gaps (normal slots) against the build with
# all ten in one gap (build_abl/libmsiren_morlet_before.so), same box; Morlet / bit-identity tests first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/morlet
rm -rf $out && mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_ws.py tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -k "morlet or same_bits or instance" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  run new_1s_r$r --activation morlet --streams 1 --no-cpu-baseline --no-extras --check
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_morlet_before.so run old_1s_r$r --activation morlet --streams 1 --no-cpu-baseline --no-extras --check
  run new_8s_r$r --activation morlet --streams 1 --slices 8 --no-cpu-baseline --no-extras --steps 100
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_morlet_before.so run old_8s_r$r --activation morlet --streams 1 --slices 8 --no-cpu-baseline --no-extras --steps 100
  run new_2s_r$r --activation morlet --no-cpu-baseline --no-extras
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_morlet_before.so run old_2s_r$r --activation morlet --no-cpu-baseline --no-extras
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/morlet/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']; a=d.get('roofline_kernel_alone') or {}
        print(f.split('/')[-1].ljust(18), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],4), round(r['avg_launch_ms'],4), '| alone', round(a.get('frac',0),4), d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
