#!/bin/bash
# Round 5: first run of the split-fp16 encoder tail + Modulator (one launch): accuracy against the fp64 oracle beside the fp32
# launches per layer, then the GPU suite, then one-stream / two-stream lines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/run2
rm -rf $out && mkdir -p $out
timeout -k 10 300 python3 tools/prologue_check.py > $out/check_new.txt 2>&1; echo "check rc=$?"; grep -v amdgpu.ids $out/check_new.txt | tail -25
MSIREN_PROLOGUE_F16X3=0 timeout -k 10 300 python3 tools/prologue_check.py > $out/check_old.txt 2>&1; grep -v amdgpu.ids $out/check_old.txt | tail -8
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }; }
run s1 --streams 1 --steps 600 --warmup 30 --no-cpu-baseline --no-extras
MSIREN_PROLOGUE_F16X3=0 run s1_old --streams 1 --steps 600 --warmup 30 --no-cpu-baseline --no-extras
run s2 --steps 600 --warmup 30 --no-cpu-baseline --no-extras
MSIREN_PROLOGUE_F16X3=0 run s2_old --steps 600 --warmup 30 --no-cpu-baseline --no-extras
run strong64_s1 --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras
MSIREN_PROLOGUE_F16X3=0 run strong64_s1_old --total-slices 64 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-extras
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/run2/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3), [(k['kernel'][12:19], round(k['avg_launch_ms'],3)) for k in r['timed_region_kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
