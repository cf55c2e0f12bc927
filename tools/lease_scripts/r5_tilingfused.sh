#!/bin/bash
# Round 5: the slice pipeline's tiling + flags + plan as one launch, the pass counter's reset inside the fold (MSIREN_TILING_FUSED=0: separate)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/tilingfused
rm -rf $out && mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_eval.py -q -x > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 -m pytest tests -m gpu -q -x -k "recon or harness or tiling or black or slice or masked" > $out/pytest2.log 2>&1; rc=$?; tail -3 $out/pytest2.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/host_reconstruct_ab.py 2>&1 | grep -v amdgpu | tee $out/ab.txt
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run rec_fused --pipeline reconstruct
MSIREN_TILING_FUSED=0 run rec_sep --pipeline reconstruct
run rec_mask_fused --pipeline reconstruct --brain-mask
MSIREN_TILING_FUSED=0 run rec_mask_sep --pipeline reconstruct --brain-mask
run rec_s1_fused --pipeline reconstruct --streams 1
MSIREN_TILING_FUSED=0 run rec_s1_sep --pipeline reconstruct --streams 1
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms')
    except Exception as e: print(f, 'ERR', e)
PY
