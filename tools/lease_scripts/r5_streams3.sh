#!/bin/bash
# Round 5: msiren_set_streams(h, 3) -- call k+2's prologue no longer queues behind call k's trunk.  Config 5 (a trunk that owns its CUs,
# 1.76 rounds per slice) and the default line, two against three streams, same box; then the stream tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/streams3
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
C5="--model deep_residual --precision bf16 --steps 300 --warmup 20"
for rep in 1 2; do
run c5_s2_$rep --streams 2 $C5
run c5_s3_$rep --streams 3 $C5
run def_s2_$rep --streams 2
run def_s3_$rep --streams 3
done
run morlet_s3 --streams 3 --activation morlet
run c5_8_s2 --streams 2 --model deep_residual --precision bf16 --slices 8 --steps 60 --warmup 5
run c5_8_s3 --streams 3 --model deep_residual --precision bf16 --slices 8 --steps 60 --warmup 5
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(18), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3))
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "stream or config5 or x1" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
