#!/bin/bash
# Round 6, lease 8: tools/soak.py 120, every configuration, torch imported first (bundled HIP 7.0 runtime -- the runtime the round-5 fault was seen on).
# (First attempt: output piped through `grep | tee` -- grep block-buffers into a pipe, nothing reached gpurun_out/ for 7 minutes and the run was taken
#  for hung and killed at 430 s.  The output now goes straight to the file.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/soak2; rm -rf $out; mkdir -p $out
timeout -k 10 1000 python3 -u -c "
import sys, runpy, torch
torch.cuda.init()
sys.argv = ['tools/soak.py', '120']
runpy.run_path('tools/soak.py', run_name='__main__')
" > $out/soak_torch_first_all.txt 2>&1
echo "rc $?"; grep -v amdgpu.ids $out/soak_torch_first_all.txt
