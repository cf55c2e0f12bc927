#!/bin/bash
# Round-4 record, second part: the default and driver-like lines with every other configuration measured in child runs
# (extra.configs), and the GPU test that asserts the line's contract.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/final2
rm -rf $out && mkdir -p $out
run() { name=$1; shift; t0=$SECONDS; timeout -k 10 500 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; echo "$name: $((SECONDS - t0)) s wall"; }
run bench_driver_like --steps 20 --warmup 5
run bench_default
timeout -k 10 900 python3 -m pytest tests/test_gpu_multi.py -x -q -k "single_gpu_line or strong_scaling" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
python3 - <<'PY'
import json
for n in ('bench_driver_like','bench_default'):
    d=json.loads([l for l in open(f'gpurun_out/r4/final2/{n}.json').read().strip().splitlines() if l.startswith('{')][-1])
    r=d['roofline']; a=d['roofline_kernel_alone']
    print(n, round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['frac'],3), 'alone', a['kernel'], round(a['frac'],3), d['extra']['host_to_host_mpixel_s'], d['extra']['reconstruct_mpixel_s'])
    for k,v in d['extra']['configs'].items():
        if isinstance(v,dict): print('   ',k, v.get('error') or (round(v['value'],1), round(v['ms_per_step'],4), v['timed_region_kernel'], round(v['timed_frac'],3), v['kernel'], round(v['kernel_alone_frac'],3)))
PY
