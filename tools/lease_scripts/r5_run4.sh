#!/bin/bash
# Round 5: split-fp16 prologue -- ring depth 8 / 4 on one stream; s_setprio beside the register-resident trunk on two; and the
# experiment "weight-stationary trunk on two streams with a few CUs left to the prologue" (MSIREN_WS_TWO=1, MSIREN_GRID=n).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/run4
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
A="--steps 600 --warmup 30 --no-cpu-baseline --no-extras"
run s1_d8 --streams 1 $A
MSIREN_EM_DEPTH=4 run s1_d4 --streams 1 $A
MSIREN_PROLOGUE_F16X3=0 run s1_old --streams 1 $A
run s2_prio $A
MSIREN_PROLOGUE_F16X3=0 run s2_old $A
for g in 256 252 248 240 232; do
  MSIREN_WS_TWO=1 MSIREN_GRID=$g MSIREN_EM_DEPTH=8 run s2_ws_g${g}_d8 $A
done
MSIREN_WS_TWO=1 MSIREN_GRID=248 MSIREN_EM_DEPTH=4 run s2_ws_g248_d4 $A
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/run4/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
for cfg in "s1:--streams 1:" "s2ws248:--streams 2:MSIREN_WS_TWO=1 MSIREN_GRID=248 MSIREN_EM_DEPTH=8"; do
  IFS=: read name args envs <<< "$cfg"
  env $envs timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py $args --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/trace_$name.log 2>&1
  f=$(find $out/trace_$name -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$name.csv; cut -c1-150 $out/kernel_stats_$name.csv | head -6
done
