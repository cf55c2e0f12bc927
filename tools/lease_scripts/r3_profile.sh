#!/bin/bash
# Round-3 record: rocprofv3 kernel stats of the default (two streams) and single-stream commands, PMC passes of the
# single-stream command (the weight-stationary trunk alone) and of the register-resident trunk forced (MSIREN_F16_WS=0).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/prof
rm -rf $out && mkdir -p $out
for m in default streams1; do
  args=""; [ $m = streams1 ] && args="--streams 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$m -- python3 bench.py $args --no-cpu-baseline --no-extras > $out/prof_$m.json 2> $out/prof_$m.err
  f=$(find $out/prof_$m -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_$m.csv; rm -rf $out/prof_$m
  echo "kernel stats ($m): done"
done
bash tools/profile.sh $out/pmc_ws --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $out/pmc_ws.log 2>&1
python3 tools/pmc_summary.py $out/pmc_ws > $out/pmc_summary_ws.txt; rm -rf $out/pmc_ws
echo "pmc ws: done"
MSIREN_F16_WS=0 bash tools/profile.sh $out/pmc_n --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $out/pmc_n.log 2>&1
python3 tools/pmc_summary.py $out/pmc_n > $out/pmc_summary_n.txt; rm -rf $out/pmc_n
echo "pmc n: done"
head -12 $out/kernel_stats_default.csv | cut -c1-160
head -12 $out/kernel_stats_streams1.csv | cut -c1-160
grep -A32 "f16x3w" $out/pmc_summary_ws.txt | head -40
