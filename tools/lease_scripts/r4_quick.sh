#!/bin/bash
# quick GPU loop: the tests named in $K (pytest -k expression), log under gpurun_out/r4/quick
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/quick
rm -rf $out && mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "${K:-config5}" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log | cut -c1-220
