#!/bin/bash
# instruction-cache behaviour of the trunk kernels (rocprofv3 PMC pass, counters only). Usage: tools/pmc_icache.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/r2/icache}; shift || true
ARGS=${@:---steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc" -- python3 bench.py $ARGS > "$OUT/pmc.log" 2>&1 || echo "pmc pass failed"
python3 tools/pmc_summary.py "$OUT/pmc" | grep -A10 "siren_trunk" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
