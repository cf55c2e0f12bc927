#!/bin/bash
# Round 5: L2-prefetch workgroups in the one-launch prologue's grid (latency sizes): same-box A/B + kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/${OUT:-prefetch}
rm -rf $out && mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_prologue.py -x -q 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
A="--steps 600 --warmup 30 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  run s1_pf_$rep --streams 1 $A
  MSIREN_EM_PREFETCH=0 run s1_nopf_$rep --streams 1 $A
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms; trunk launch', round(r['timed_region_kernels'][0]['avg_launch_ms'],4))
PY
for cfg in "pf:" "nopf:MSIREN_EM_PREFETCH=0"; do
  IFS=: read name envs <<< "$cfg"
  env $envs timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py --streams 1 --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/trace_$name.log 2>&1
  f=$(find $out/trace_$name -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$name.csv; cut -c1-150 $out/kernel_stats_$name.csv | head -5
done
