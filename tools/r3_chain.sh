#!/bin/bash
# Modulator chain (one launch for conv3 + Linear + Modulator layers): parity suite, then chain on / off on the same box:
# latency (B = 1 / 8 / 400) and the single-stream bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/chain
rm -rf $out && mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_chain.py -x -q > $out/pytest_chain.log 2>&1; rc=$?; tail -5 $out/pytest_chain.log
[ $rc -eq 0 ] || exit $rc
for c in 1 0; do
  MSIREN_CHAIN=$c timeout -k 10 200 python3 tools/latency.py > $out/latency_chain$c.txt 2>&1 || exit 1
  MSIREN_CHAIN=$c timeout -k 10 300 python3 bench.py --streams 1 --no-cpu-baseline --no-extras > $out/bench_streams1_chain$c.json 2> $out/bench_streams1_chain$c.err || exit 1
done
python3 - <<'PY'
import json
for c in (1, 0):
    print(f"chain={c}")
    print(open(f"gpurun_out/r3/chain/latency_chain{c}.txt").read().strip())
    d = json.loads(open(f"gpurun_out/r3/chain/bench_streams1_chain{c}.json").read().strip().splitlines()[-1])
    print(f"  streams1: {d['value']:.1f} Mpixel/s, {d['ms_per_step']*1e3:.1f} us per slice, trunk {d['roofline']['avg_launch_ms']*1e3:.1f} us")
PY
