#!/bin/bash
# Bench lines of every BASELINE configuration + kernel stats, for profiles/ (run via gpurun).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; tail -c 300 $out/$name.json | head -c 0; }
run bench_default
run bench_streams1 --streams 1 --no-cpu-baseline
run bench_morlet --activation morlet --no-cpu-baseline --check
run bench_slices8 --slices 8 --no-cpu-baseline
run bench_reconstruct --pipeline reconstruct --no-cpu-baseline --check
run bench_fp32 --precision fp32 --no-cpu-baseline --check
run bench_config5_bf16 --model deep_residual --precision bf16 --no-cpu-baseline --check
run bench_config5_f16 --model deep_residual --precision f16 --no-cpu-baseline --check
run bench_config5_fp32 --model deep_residual --precision fp32 --no-cpu-baseline --check --steps 50
for m in default streams1; do
  args=""; [ $m = streams1 ] && args="--streams 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$m -- python3 bench.py $args --no-cpu-baseline > $out/prof_$m.json 2> $out/prof_$m.err
  f=$(find $out/prof_$m -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$m.csv; rm -rf $out/prof_$m
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['achieved'],1),'TF', round(r['frac'],3), d.get('check_nerr_vs_fp64_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
