#!/bin/bash
# Round 6, lease 7: rehearsals of bench.py --gpus N for N = 3, 4 on the ONE card of the box (ranks share it: the efficiency printed is ~1/N by
# construction -- what is rehearsed is the code path: launcher, rendezvous, uneven shards, both regions, the JSON line).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/rehearsal; rm -rf $out; mkdir -p $out
show() { python3 -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); s=d['extra']['configs']['config3_64_slices_strong']
print('$1: n_gpus', d['n_gpus'], 'backend', d['config']['backend'][:28], '| weak value', round(d['value'],1), '| strong', round(s['value'],1), 'Mpixel/s', 'slices_per_rank', s['slices_per_rank'], 'eff_vs_n1', round(s['efficiency_vs_n1'],3), 'rccl_ranks', s['rccl_ranks'], 'identical weights', d['config']['ranks_hold_identical_weights'])
"; }
MSIREN_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 3 --steps 20 --warmup 5 --no-cpu-baseline > $out/gloo_n3.json 2> $out/gloo_n3.err && show $out/gloo_n3.json
MSIREN_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > $out/gloo_n4.json 2> $out/gloo_n4.err && show $out/gloo_n4.json
# stub RCCL (file-carried collectives, the library's own msiren_comm_* path), four ranks under torch.distributed.run as the driver starts them
g++ -std=c++17 -O1 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/rccl_stub.cpp -o $out/librccl_stub.so -L/opt/rocm/lib -lamdhip64 || exit 1
MSIREN_BENCH_ALLOW_SHARED=1 MSIREN_RCCL_LIB=$PWD/$out/librccl_stub.so RCCL_STUB_DIR=$PWD/$out timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 4 --steps 20 --warmup 5 > $out/torchrun_stub_n4.json 2> $out/torchrun_stub_n4.err && show $out/torchrun_stub_n4.json
tail -3 $out/torchrun_stub_n4.err
