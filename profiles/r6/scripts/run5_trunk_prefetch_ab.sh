#!/bin/bash
# Round 6, lease 5: the next call's prologue weights pulled into the L2s by 64 extra workgroups at the END of the
# weight-stationary trunk's grid (MSIREN_TRUNK_PREFETCH=1; =2: and the tail's own in-grid prefetch workgroups off).  Same box, interleaved.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/pf; rm -rf $out; mkdir -p $out
# correctness first: the instances' bit-identity tests and the host-call tests with the experiment on
MSIREN_TRUNK_PREFETCH=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_ws.py tests/test_gpu_host_calls.py -q -m gpu -x > $out/pytest_pf1.log 2>&1; rc=$?; echo "pytest pf=1 rc $rc"; tail -3 $out/pytest_pf1.log
if [ $rc -ne 0 ]; then exit $rc; fi
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0])
k={x['kernel']:x for x in d['roofline']['timed_region_kernels']}
print('$1', round(d['value'],2), 'Mpixel/s', round(d['ms_per_step']*1000,1), 'us/step; trunk', {n:round(x['avg_launch_ms']*1000,1) for n,x in k.items()}, 'h2h', d.get('host_to_host',{}).get('value'))
"; }
for rep in 1 2 3; do
  for v in 0 1 2; do
    MSIREN_TRUNK_PREFETCH=$v timeout -k 10 200 python3 bench.py --streams 1 --steps 2000 --warmup 50 --no-cpu-baseline --no-extras > $out/s1_pf${v}_$rep.json 2> $out/err.txt || exit 1
    line $out/s1_pf${v}_$rep.json
  done
done
# host -> host (the synchronous pattern the prefetch is for): extras of the one-stream command
for v in 0 1 0 1; do
  MSIREN_TRUNK_PREFETCH=$v timeout -k 10 300 python3 bench.py --streams 1 --steps 500 --warmup 50 --no-cpu-baseline > $out/h2h_pf${v}_$RANDOM.json 2> $out/err.txt || exit 1
done
for f in $out/h2h_pf*.json; do python3 -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); e=d['extra']
print('$f', round(d['value'],1), 'h2h', round(e['host_to_host_mpixel_s'],1), 'pinned', round(e['host_to_host_pinned_mpixel_s'],1), 'slice->slice host', round(e['host_slice_to_slice_mpixel_s'],1), 'dev recon', round(e['reconstruct_mpixel_s'],1))
"; done
# kernel durations under rocprofv3, both variants
for v in 0 1; do
  MSIREN_TRUNK_PREFETCH=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -- python3 bench.py --streams 1 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras > $out/prof_$v.json 2> $out/prof_$v.err
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_streams1_pf$v.csv; rm -rf $out/prof_$v; echo "pf=$v"; head -5 $out/kernel_stats_streams1_pf$v.csv | cut -c1-160
done
