#!/bin/bash
# Round 6, lease 1: the GPU suite ONCE on the tree without per-call registration; the default bench line and the same command with
# --torch-first (libmsiren on torch's bundled HIP runtime) back to back, twice (same-box A/B); rocprofv3 kernel stats + PMC passes of the
# exact-fp32 trunk (siren_trunk_f32_kernel<256,0,0>), which had no record under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/run1
rm -rf $out && mkdir -p $out
python3 -c "import sys; sys.path.insert(0, '.'); from mri_inr_amd import _lib; print('torch-free:', _lib.runtime_info())" > $out/runtime.txt 2>&1
python3 -c "import sys, torch; torch.cuda.init(); sys.path.insert(0, '.'); from mri_inr_amd import _lib; print('torch-first:', _lib.runtime_info())" >> $out/runtime.txt 2>&1
cat $out/runtime.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 $out/pytest.log
if [ $rc -ge 124 ]; then echo "suite killed: no further GPU step in this call"; exit $rc; fi
b() { name=$1; shift; timeout -k 10 420 python3 bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc $?"; python3 -c "
import json,sys
d=json.loads([l for l in open('$out/bench_$name.json') if l.startswith('{')][0])
print('$name', round(d['value'],1), d['config']['hip_runtime']['hip_runtime_version'], d['config']['hip_runtime']['libamdhip64'], 'h2h', d.get('host_to_host',{}).get('value'), 'fp32', d.get('fp32',{}).get('value'))
" || true; }
b default --steps 1000 --warmup 50 || exit 1
b torch_first --torch-first --steps 1000 --warmup 50 || exit 1
b default_2 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras
b torch_first_2 --torch-first --steps 1000 --warmup 50 --no-cpu-baseline --no-extras
b default_streams1 --streams 1 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras
b torch_first_streams1 --torch-first --streams 1 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras
# exact-fp32 trunk: kernel stats + PMC (separate passes: tools/profile.sh)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_fp32 -- python3 bench.py --precision fp32 --streams 1 --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/bench_under_rocprof_fp32.json 2> $out/prof_fp32.err
f=$(find $out/prof_fp32 -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_fp32_streams1.csv; rm -rf $out/prof_fp32; head -5 $out/kernel_stats_fp32_streams1.csv | cut -c1-200
bash tools/profile.sh $out/pmc_fp32 --precision fp32 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/pmc_fp32.log 2>&1
python3 tools/pmc_summary.py $out/pmc_fp32 > $out/pmc_summary_fp32_streams1.txt; rm -rf $out/pmc_fp32
grep -A30 "siren_trunk_f32_kernel" $out/pmc_summary_fp32_streams1.txt | head -40
