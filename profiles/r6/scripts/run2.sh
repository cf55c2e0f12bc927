#!/bin/bash
# Round 6, lease 2: the whole GPU suite ONCE on the slimmer tree (host side cut into units, losers and their knobs deleted), then smoke.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/run2
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $out/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -15 $out/pytest.log
if [ $rc -ge 124 ]; then echo "suite killed: no further GPU step in this call"; exit $rc; fi
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.log
