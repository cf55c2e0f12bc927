#!/bin/bash
# full GPU suite (what the driver runs at round end), log under gpurun_out/r4/suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/suite
rm -rf $out && mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
