#!/bin/bash
# Round-5 record: full GPU parity suite, bench lines of every BASELINE configuration (run via gpurun; outputs under
# gpurun_out/r5/final, summaries copied to profiles/r5/10_final).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/final
rm -rf $out && mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; sed -i '/^\[Gloo\]/d' $out/$name.json; }
run bench_default
run bench_driver_like --steps 20 --warmup 5
run bench_streams1 --streams 1 --no-cpu-baseline
run bench_morlet --activation morlet --no-cpu-baseline --check
run bench_morlet_streams1 --activation morlet --streams 1 --no-cpu-baseline --no-extras --check
run bench_slices8 --slices 8 --no-cpu-baseline
run bench_slices8_streams1 --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 200 --warmup 20
run bench_strong64_n1 --total-slices 64 --steps 40 --warmup 5 --no-cpu-baseline
run bench_strong64_n1_streams1 --total-slices 64 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras
MSIREN_SPLIT_MIN=0 run bench_strong64_n1_streams1_uncut --total-slices 64 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras
run bench_reconstruct --pipeline reconstruct --no-cpu-baseline --check
run bench_reconstruct_mask --pipeline reconstruct --brain-mask --no-cpu-baseline --check
run bench_fp32 --precision fp32 --no-cpu-baseline --check --steps 300
run bench_config5_bf16 --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
run bench_config5_f16 --model deep_residual --precision f16 --no-cpu-baseline --check --steps 300
g++ -std=c++17 -O1 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/rccl_stub.cpp -o $out/librccl_stub.so -L/opt/rocm/lib -lamdhip64 \
  && MSIREN_BENCH_ALLOW_SHARED=1 MSIREN_RCCL_LIB=$PWD/$out/librccl_stub.so RCCL_STUB_DIR=$PWD/$out run bench_two_ranks_one_card_stub_rccl --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline
rm -f $out/librccl_stub.so $out/stub-*
MSIREN_BENCH_BACKEND=gloo run bench_strong64_gloo4_one_card --gpus 4 --total-slices 64 --steps 10 --warmup 3 --no-cpu-baseline
timeout -k 10 300 python3 bench.py --scaling-selftest --steps 300 --warmup 20 > $out/scaling_selftest.json 2> $out/scaling_selftest.err
python3 tools/latency.py > $out/latency.txt 2>&1; timeout -k 10 200 python3 tools/latency_sweep.py > $out/latency_sweep_host_calls.txt 2>&1; timeout -k 10 300 python3 tools/host_pinned_ab.py > $out/host_calls_pageable_vs_pinned.txt 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/final/bench_*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; a=d.get('roofline_kernel_alone',{})
        print(f.split('/')[-1].ljust(40), d['n_gpus'], d['scaling'], round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['frac'],3),
              '| alone', a.get('kernel'), round(a.get('frac',0),3), d.get('check_nerr_vs_fp64_oracle'), {k:(round(v,1) if isinstance(v,float) else None) for k,v in (d.get('extra') or {}).items() if k.endswith('_s')}, d.get('collective_fallback'), d['config'].get('rccl_ranks'))
    except Exception as e: print(f, 'ERR', e)
d=json.loads([l for l in open('gpurun_out/r5/final/bench_driver_like.json').read().strip().splitlines() if l.startswith('{')][-1])
for k,v in d['extra']['configs'].items():
    if isinstance(v,dict): print(k, round(v['value'],1), round(v['ms_per_step'],4), v['kernel'], round(v['kernel_alone_frac'],3), round(v['timed_frac'],3), v.get('one_stream'))
PY
cat $out/scaling_selftest.json; tail -4 $out/latency.txt
