#!/bin/bash
# ablation builds of the weight-stationary trunk (results wrong, timing only): slot timeline of each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in 1 3 7; do
  echo "=== MSIREN_WS_ABL=$a (1 = no epilogue, 2 = no B-fragment reads, 4 = no barrier)"
  MSIREN_LIB=$GRAFT_REPO_ROOT/build_abl/libmsiren_abl$a.so timeout -k 10 120 python3 tools/timeline_f16x3w.py 3200 | tail -14
done
