#!/bin/bash
# Round-3 baseline on this round's box: GPU suite + default bench lines of the code as round 2 left it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/baseline
rm -rf $out && mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; }
run bench_default --no-cpu-baseline
run bench_streams1 --streams 1 --no-cpu-baseline
run bench_morlet --activation morlet --no-cpu-baseline
run bench_slices8 --slices 8 --no-cpu-baseline
python3 tools/latency.py > $out/latency.txt 2>&1
tail -n 3 $out/*.json $out/latency.txt
