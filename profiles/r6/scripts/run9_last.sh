#!/bin/bash
# Round 6, lease 9: the GPU suite once more on the very last tree (host-side changes since lease 6: visual_error's image writers and error file,
# flat scalar copies in the bench line, the C consumer's runtime line), and the run-to-run spread of the driver's 20-step timed region: ten
# runs of `bench.py --gpus 1 --steps 20 --warmup 5` (side measurements off) on one box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/last; rm -rf $out; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 $out/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/d_$i.json 2>> $out/err.txt || exit 1
done
python3 - <<'PY' | tee gpurun_out/r6/last/spread.txt
import json, glob, statistics as st
v=[json.loads([l for l in open(f) if l.startswith('{')][0]) for f in sorted(glob.glob('gpurun_out/r6/last/d_*.json'))]
vals=[d['value'] for d in v]; fr=[d['roofline']['frac'] for d in v]
print('ten runs of the driver command (20 timed steps = 5.3 ms each), Mpixel/s:', ' '.join(f'{x:.1f}' for x in vals))
print(f'min {min(vals):.1f} median {st.median(vals):.1f} max {max(vals):.1f}  spread (max-min)/median {100*(max(vals)-min(vals))/st.median(vals):.2f} %  stdev {st.pstdev(vals):.2f}')
print('roofline.frac:', ' '.join(f'{x:.4f}' for x in fr))
PY
