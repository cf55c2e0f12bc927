#!/bin/bash
# config 5, two streams: kernel timeline (rocprofv3 --kernel-trace) with the trunk on every CU and capped to 225 workgroups
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/x1w_trace
rm -rf $out && mkdir -p $out
for g in 256 225; do
  export MSIREN_X1_GRID=$g
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t$g -- python3 bench.py --model deep_residual --precision bf16 --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $out/bench_$g.json 2> $out/bench_$g.err
  f=$(find $out/t$g -name "*kernel_trace.csv" | head -1); python3 - "$f" $g <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'msiren' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tr=[r for r in rows if 'x1w' in r['Kernel_Name']]
print('grid',sys.argv[2],'trunk launches',len(tr))
# steady-state: last 12 trunks: print start, end (us rel), duration, and for the prologue kernels between
t0=int(tr[-12]['Start_Timestamp'])
sel=[r for r in rows if int(r['Start_Timestamp'])>=t0-200000]
for r in sel[:140]:
    s=(int(r['Start_Timestamp'])-t0)/1000; e=(int(r['End_Timestamp'])-t0)/1000
    nm=r['Kernel_Name'].split('(')[0].replace('void msiren::','').replace('msiren::','')[:40]
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f} q{r.get('Queue_Id','?')} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))} {nm}")
PY
done
