#!/bin/bash
# Round-5 record: rocprofv3 kernel stats of the default (two streams), the single-stream, the 64-slice single-stream (cut call), the
# config-5 and the Morlet commands; PMC passes (separate runs, tools/profile.sh) of the default, the single-stream and the config-5
# commands; the config-5 command once more on the -DMSIREN_X1W_ABL=8 build (every layer reads layer 1's weights: an L2-resident
# weight stream) for the counter delta the round-4 review asked for.  traffic.json is assembled from the summaries afterwards.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/prof
rm -rf $out && mkdir -p $out
stats() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/bench_under_rocprof_$name.json 2> $out/prof_$name.err
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$name.csv; rm -rf $out/prof_$name; echo "kernel stats ($name): done"; }
stats default_two_streams
stats streams1 --streams 1
stats strong64_streams1 --total-slices 64 --streams 1 --steps 30 --warmup 3
stats config5_bf16 --model deep_residual --steps 200 --warmup 10
stats morlet_streams1 --activation morlet --streams 1
pmc() { name=$1; shift
  bash tools/profile.sh $out/pmc_$name --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/pmc_$name.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$name > $out/pmc_summary_$name.txt; rm -rf $out/pmc_$name; echo "pmc $name: done"; }
pmc default_two_streams
pmc streams1 --streams 1
pmc config5_bf16 --model deep_residual
MSIREN_LIB=$GRAFT_REPO_ROOT/build_ab/libmsiren_x1w_abl8.so pmc config5_bf16_l2_resident_weights --model deep_residual
for f in $out/kernel_stats_*.csv; do echo $f; head -6 $f | cut -c1-170; done
grep -A3 "x1w_kernel" $out/pmc_summary_config5_bf16.txt | head -5; grep -A3 "x1w_kernel" $out/pmc_summary_config5_bf16_l2_resident_weights.txt | head -5
