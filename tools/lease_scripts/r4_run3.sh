#!/bin/bash
# Round 4, third build: conditional exact-fp32 launch behind every f16x3 trunk launch (identical semantics outside the fp16
# domain), per-rank device table, scaling selftest.  Full GPU suite, then what the conditional launch costs (same-box A/B
# through MSIREN_RANGE_RERUN), and the selftest.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/run3
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for r in 1 2; do
  MSIREN_RANGE_RERUN=0 run noguard_s2_r$r --no-cpu-baseline --no-extras
  run guard_s2_r$r --no-cpu-baseline --no-extras
  MSIREN_RANGE_RERUN=0 run noguard_s1_r$r --streams 1 --no-cpu-baseline --no-extras
  run guard_s1_r$r --streams 1 --no-cpu-baseline --no-extras
done
timeout -k 10 300 python3 bench.py --scaling-selftest --steps 300 --warmup 20 > $out/selftest.json 2> $out/selftest.err; echo "selftest rc=$?"; cat $out/selftest.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/run3/*guard*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), d['config']['ranks'])
    except Exception as e: print(f, 'ERR', e)
PY
