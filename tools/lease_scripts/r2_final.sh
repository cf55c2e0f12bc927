#!/bin/bash
# Round-2 record: full GPU parity suite, bench lines of every BASELINE configuration, rocprofv3 kernel stats and the
# PMC passes of the default command (run via gpurun; outputs under gpurun_out/r2/final, summaries copied to profiles/r2).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2/final
rm -rf $out && mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; }
run bench_default
run bench_driver_like --steps 20 --warmup 5
run bench_streams1 --streams 1 --no-cpu-baseline
run bench_morlet --activation morlet --no-cpu-baseline --check
run bench_slices8 --slices 8 --no-cpu-baseline
run bench_strong64_n1 --total-slices 64 --steps 40 --warmup 5 --no-cpu-baseline
MSIREN_BENCH_BACKEND=gloo run bench_strong64_gloo4_one_card --gpus 4 --total-slices 64 --steps 10 --warmup 3 --no-cpu-baseline
run bench_reconstruct --pipeline reconstruct --no-cpu-baseline --check
run bench_reconstruct_mask --pipeline reconstruct --brain-mask --no-cpu-baseline --check
run bench_fp32 --precision fp32 --no-cpu-baseline --check --steps 300
run bench_config5_bf16 --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
for m in default streams1; do
  args=""; [ $m = streams1 ] && args="--streams 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$m -- python3 bench.py $args --no-cpu-baseline --no-extras > $out/prof_$m.json 2> $out/prof_$m.err
  f=$(find $out/prof_$m -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$m.csv; rm -rf $out/prof_$m
done
bash tools/profile.sh $out/pmc --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $out/pmc.log 2>&1
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt; rm -rf $out/pmc
python3 tools/latency.py > $out/latency.txt 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/final/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], d['n_gpus'], d['scaling'], round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['achieved'],1),'TF', round(r['frac'],3), d.get('check_nerr_vs_fp64_oracle'), d.get('extra'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -A30 "siren_trunk_f16x3n" $out/pmc_summary.txt | head -32
cat $out/latency.txt | tail -8
