#!/bin/bash
# Round-4 record: rocprofv3 kernel stats of the default (two streams), the single-stream and the 64-slice single-stream
# (cut call) commands; PMC passes (separate runs) of the default command (register-resident trunk of the timed region) and
# of the single-stream command (weight-stationary trunk); traffic.json from FETCH_SIZE / WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/prof
rm -rf $out && mkdir -p $out
stats() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/bench_under_rocprof_$name.json 2> $out/prof_$name.err
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$name.csv; rm -rf $out/prof_$name; echo "kernel stats ($name): done"; }
stats default_two_streams
stats streams1 --streams 1
stats strong64_streams1 --total-slices 64 --streams 1 --steps 30 --warmup 3
stats config5_bf16 --model deep_residual --steps 200 --warmup 10
stats morlet_streams1 --activation morlet --streams 1
bash tools/profile.sh $out/pmc_default --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/pmc_default.log 2>&1
python3 tools/pmc_summary.py $out/pmc_default > $out/pmc_summary_default_two_streams.txt; rm -rf $out/pmc_default; echo "pmc default: done"
bash tools/profile.sh $out/pmc_s1 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $out/pmc_s1.log 2>&1
python3 tools/pmc_summary.py $out/pmc_s1 > $out/pmc_summary_streams1.txt; rm -rf $out/pmc_s1; echo "pmc s1: done"
bash tools/profile.sh $out/pmc_c5 --steps 20 --warmup 5 --model deep_residual --no-cpu-baseline --no-extras > $out/pmc_c5.log 2>&1
python3 tools/pmc_summary.py $out/pmc_c5 > $out/pmc_summary_config5_bf16.txt; rm -rf $out/pmc_c5; echo "pmc config5: done"
for f in $out/kernel_stats_*.csv; do echo $f; head -6 $f | cut -c1-170; done
grep -A30 "f16x3n_kernel" $out/pmc_summary_default_two_streams.txt | head -34
