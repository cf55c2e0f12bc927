#!/bin/bash
# Attribution of the weight-stationary config-3 trunk's 22 % over its bare MFMA stream: single- and double-bit ablation builds
# (-DMSIREN_WS_ABL: 1 no epilogue, 2 no B-fragment reads, 4 no slot barrier; results wrong, timing only), 8 slices, one stream.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/ws_abl
rm -rf $out && mkdir -p $out
for a in shipped abl1 abl2 abl4 abl3 abl5 abl6 abl7 shipped2; do
  lib=$GRAFT_REPO_ROOT/build_abl/libmsiren_$a.so; case $a in shipped*) lib=$GRAFT_REPO_ROOT/mri_inr_amd/libmsiren.so;; esac
  MSIREN_LIB=$lib timeout -k 10 200 python3 bench.py --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 60 --warmup 10 > $out/$a.json 2> $out/$a.err || { echo "$a failed"; tail -2 $out/$a.err; }
done
python3 - <<'PY'
import json
for a in ("shipped", "abl1", "abl2", "abl4", "abl3", "abl5", "abl6", "abl7", "shipped2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4/ws_abl/{a}.json").read().strip().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
        ms = r["avg_launch_ms"]; slots = 8 * 7200 * 4 / 256.0
        print(f"{a:9s} launch {ms:.4f} ms for 8 slices = {ms * 1e3 / slots:.3f} us per slot and CU   {r['kernel']}")
    except Exception as e: print(a, "ERR", e)
PY
