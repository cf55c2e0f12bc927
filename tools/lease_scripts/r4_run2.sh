#!/bin/bash
# Round 4, second build: 32 x 32-tile Linear layers at throughput sizes (bit-identical to the 16 x 16 kernel), the split on
# one-stream handles only.  Tests, then same-box A/B: tiles off / on x split off / on at 64 and 8 slices per call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/run2
rm -rf $out && mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_ws.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
for s in 1 2; do
  for t in 0 1024; do
    MSIREN_LINEAR_TILE_MIN=$t MSIREN_SPLIT_MIN=0 run uncut64_t${t}_s$s --total-slices 64 --streams $s --steps 30 --warmup 3 --no-cpu-baseline --no-extras
    MSIREN_LINEAR_TILE_MIN=$t MSIREN_SPLIT_MIN=0 run uncut8_t${t}_s$s --slices 8 --streams $s --steps 200 --warmup 10 --no-cpu-baseline --no-extras
  done
done
for pct in 6 9 12; do MSIREN_SPLIT_PCT=$pct run split64_p${pct}_s1 --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras; done
for pct in 9 12 16; do MSIREN_SPLIT_PCT=$pct run split8_p${pct}_s1 --slices 8 --streams 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof64 -- python3 bench.py --total-slices 64 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/prof64.json 2> $out/prof64.err
f=$(find $out/prof64 -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_strong64_s1.csv; rm -rf $out/prof64
cut -c1-180 $out/kernel_stats_strong64_s1.csv | head -8
run default --cpu-seconds 3
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/run2/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3),
              '| alone', a.get('kernel'), round(a.get('frac',0),3), '|', [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
d=json.loads([l for l in open('gpurun_out/r4/run2/default.json').read().strip().splitlines() if l.startswith('{')][-1])
for k,v in d['extra']['configs'].items():
    if isinstance(v,dict): print(k, round(v['value'],1), round(v['ms_per_step'],4), v['kernel'], round(v['kernel_alone_frac'],3), round(v['timed_frac'],3), v.get('one_stream'))
PY
