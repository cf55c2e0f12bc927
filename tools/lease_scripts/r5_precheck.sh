#!/bin/bash
# Round 5: the conditional exact-fp32 launch beside the trunk instead of behind it (prechecked launches): tests, then same-box A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/${OUT:-precheck}
rm -rf $out && mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ws.py tests/test_gpu_prologue.py tests/test_gpu_split.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
A="--steps 600 --warmup 30 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  run s1_pre_$rep --streams 1 $A
  MSIREN_PRECHECK=0 run s1_behind_$rep --streams 1 $A
done
run s2 $A
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep.txt 2>&1; grep -v amdgpu.ids $out/latency_sweep.txt | tail -4
MSIREN_PRECHECK=0 timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep_behind.txt 2>&1; grep -v amdgpu.ids $out/latency_sweep_behind.txt | tail -4
