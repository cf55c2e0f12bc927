#!/bin/bash
# Round-3 record: full GPU parity suite, bench lines of every BASELINE configuration (run via gpurun; outputs under
# gpurun_out/r3/final, summaries copied to profiles/r3/10_final).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/final
rm -rf $out && mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; sed -i '/^\[Gloo\]/d' $out/$name.json; }
run bench_default
run bench_driver_like --steps 20 --warmup 5
run bench_streams1 --streams 1 --no-cpu-baseline
MSIREN_F16_WS=0 run bench_streams1_n_forced --streams 1 --no-cpu-baseline --no-extras
run bench_morlet --activation morlet --no-cpu-baseline --check
run bench_morlet_streams1 --activation morlet --streams 1 --no-cpu-baseline --no-extras --check
run bench_slices8 --slices 8 --no-cpu-baseline
run bench_slices8_streams1 --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 200 --warmup 20
MSIREN_F16_WS=0 run bench_slices8_streams1_n_forced --slices 8 --streams 1 --no-cpu-baseline --no-extras --steps 200 --warmup 20
run bench_strong64_n1 --total-slices 64 --steps 40 --warmup 5 --no-cpu-baseline
run bench_strong64_n1_streams1 --total-slices 64 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras
run bench_reconstruct --pipeline reconstruct --no-cpu-baseline --check
run bench_reconstruct_mask --pipeline reconstruct --brain-mask --no-cpu-baseline --check
run bench_fp32 --precision fp32 --no-cpu-baseline --check --steps 300
run bench_config5_bf16 --model deep_residual --precision bf16 --no-cpu-baseline --check --steps 300
# two ranks on the one card through the RCCL stand-in of the tests (file-carried collectives): the N > 1 path end to end
g++ -std=c++17 -O1 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/rccl_stub.cpp -o $out/librccl_stub.so -L/opt/rocm/lib -lamdhip64 \
  && MSIREN_BENCH_ALLOW_SHARED=1 MSIREN_RCCL_LIB=$PWD/$out/librccl_stub.so RCCL_STUB_DIR=$PWD/$out run bench_two_ranks_one_card_stub_rccl --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline
rm -f $out/librccl_stub.so $out/stub-*
MSIREN_BENCH_BACKEND=gloo run bench_strong64_gloo4_one_card --gpus 4 --total-slices 64 --steps 10 --warmup 3 --no-cpu-baseline 
python3 tools/latency.py > $out/latency.txt 2>&1
# the opt-in one-launch modulator chain, same box: latency and the single-stream line
MSIREN_CHAIN=1 python3 tools/latency.py > $out/latency_chain.txt 2>&1
MSIREN_CHAIN=1 run bench_streams1_chain --streams 1 --no-cpu-baseline --no-extras
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/final/*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']; t=d.get('roofline_timed_mode',{})
        print(f.split('/')[-1], d['n_gpus'], d['scaling'], round(d['value'],1), 'Mpx/s', round(d['ms_per_step'],4),'ms', r['kernel'], round(r['achieved'],1),'TF', round(r['frac'],3), 'sustained', r.get('sustained_fp16_mfma_tflops_measured') and round(r['sustained_fp16_mfma_tflops_measured']), 'timed', round(t.get('frac',0),3), d.get('check_nerr_vs_fp64_oracle'), d.get('extra'), d.get('collective_fallback'), d['config'].get('rccl_ranks'))
    except Exception as e: print(f, 'ERR', e)
PY
cat $out/latency.txt | tail -4; cat $out/latency_chain.txt | tail -3
