#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run via gpurun).  Usage: tools/profile.sh <outdir> [bench args...]
# Kernel trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses --pmc combined
# with other trace domains); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slot budget).
set -u
OUT=${1:-gpurun_out/prof}; shift || true
ARGS=${@:---steps 10 --warmup 3 --no-cpu-baseline --no-extras}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run trace --kernel-trace --stats
run pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE
run pmc_sq3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*.csv" | head -40
