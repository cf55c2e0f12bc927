#!/bin/bash
# Round 6, lease 4: tools/soak.py on the final library, torch-free (system HIP 7.2) and with torch imported first (bundled 7.0): random
# sizes / offsets / stream modes through the asynchronous API, the slice pipeline, and host-pointer calls from two threads on windows of one
# pageable input pool and one output pool (every call classifies its buffers through hipPointerGetAttribute: host_range_kind).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/soak; rm -rf $out; mkdir -p $out
timeout -k 10 400 python3 tools/soak.py 60 sine > $out/soak_torch_free.txt 2>&1; rc=$?; echo "torch-free rc $rc"; tail -5 $out/soak_torch_free.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python3 -c "
import sys, runpy, torch
torch.cuda.init()
sys.argv = ['tools/soak.py', '60', 'sine']
runpy.run_path('tools/soak.py', run_name='__main__')
" > $out/soak_torch_first.txt 2>&1; echo "torch-first rc $?"; tail -5 $out/soak_torch_first.txt
