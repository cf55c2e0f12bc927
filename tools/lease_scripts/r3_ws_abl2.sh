#!/bin/bash
# un-stamped launch time of the weight-stationary trunk, shipped build against ablation builds (results wrong, timing only):
# 7 = MFMAs only (no epilogue, no B-fragment reads, no barrier), 15 = not even the MFMAs: what is left is the slots' control flow
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3/abl
for a in shipped abl7 abl15; do
  lib=$GRAFT_REPO_ROOT/build_abl/libmsiren_$a.so; [ $a = shipped ] && lib=$GRAFT_REPO_ROOT/mri_inr_amd/libmsiren.so
  MSIREN_LIB=$lib timeout -k 10 300 python3 bench.py --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 60 --warmup 10 > gpurun_out/r3/abl/$a.json 2> gpurun_out/r3/abl/$a.err || echo "$a failed"
done
python3 - <<'PY'
import json
for a in ("shipped", "abl7", "abl15"):
    d = json.loads(open(f"gpurun_out/r3/abl/{a}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    ms = r["avg_launch_ms"]; slots = 8 * 7200 * 4 / 256.0
    print(f"{a:8s} launch {ms:.4f} ms for 8 slices = {ms * 1e3 / slots:.3f} us per slot and CU")
PY
