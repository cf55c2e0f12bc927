#!/bin/bash
# Round 4, fourth build: the conditional exact-fp32 launch as a SMALL kernel (32 KB LDS, <= 96 registers: dispatched beside the
# other stream's trunk), the weight-stationary trunk's prologue in one memory round trip.  Tests of both, the cost of the
# guard (MSIREN_RANGE_RERUN), same-box A/B of the prologue (build_abl/libmsiren_oldpro.so = -DMSIREN_WS_OLD_PROLOGUE), and the
# launch-level timeline of the single-slice launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/run4
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_ws.py tests/test_gpu_split.py tests/test_gpu_parity.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
OLD=$GRAFT_REPO_ROOT/build_abl/libmsiren_oldpro.so
for r in 1 2; do
  MSIREN_RANGE_RERUN=0 run noguard_s2_r$r --no-cpu-baseline --no-extras
  run guard_s2_r$r --no-cpu-baseline --no-extras
  MSIREN_RANGE_RERUN=0 run noguard_s1_r$r --streams 1 --no-cpu-baseline --no-extras
  run guard_s1_r$r --streams 1 --no-cpu-baseline --no-extras
  MSIREN_LIB=$OLD run oldpro_s1_r$r --streams 1 --no-cpu-baseline --no-extras
  MSIREN_LIB=$OLD run oldpro_s2_r$r --no-cpu-baseline --no-extras
done
python3 tools/timeline_ws_launch.py 1 > $out/timeline_launch_1slice.txt 2>&1
MSIREN_LIB=$OLD python3 tools/timeline_ws_launch.py 1 > $out/timeline_launch_1slice_oldpro.txt 2>&1
python3 tools/timeline_ws_launch.py 8 > $out/timeline_launch_8slices.txt 2>&1
tail -12 $out/timeline_launch_1slice.txt; tail -10 $out/timeline_launch_1slice_oldpro.txt; tail -10 $out/timeline_launch_8slices.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/run4/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), round(r['avg_launch_ms'],4), '| alone', a.get('kernel'), round(a.get('frac',0),3), round(a.get('avg_launch_ms',0),4))
    except Exception as e: print(f, 'ERR', e)
PY
