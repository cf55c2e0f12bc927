#!/bin/bash
# Round 5 quick loop: accuracy of the split-fp16 prologue (three lines), bench lines one / two streams / 64 slices, and the
# rocprofv3 kernel stats of the one-stream command.  OUT=<dir under gpurun_out/r5> ENVS="A=1 B=2" tools/lease_scripts/r5_quick.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/${OUT:-quick}
rm -rf $out && mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_prologue.py -x -q > $out/check_new.txt 2>&1; echo "check rc=$?"; tail -2 $out/check_new.txt
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
A="--steps 600 --warmup 30 --no-cpu-baseline --no-extras"
run s1 --streams 1 $A
MSIREN_PROLOGUE_F16X3=0 run s1_old --streams 1 $A
run s2 $A
MSIREN_PROLOGUE_F16X3=0 run s2_old $A
run strong64_s1 --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras
MSIREN_PROLOGUE_F16X3=0 run strong64_s1_old --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
for cfg in "s1:--streams 1" "s2:--streams 2"; do
  IFS=: read name args <<< "$cfg"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py $args --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/trace_$name.log 2>&1
  f=$(find $out/trace_$name -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$name.csv; cut -c1-150 $out/kernel_stats_$name.csv | head -5
done
