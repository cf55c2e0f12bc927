# single-stream mode: ragged last round as half-units on the idle second stream (MSIREN_F16_TAIL=1, default) vs one launch (0)
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3 or fixtures or forward_tiles or two_stream or half_unit or linearity or slice_recon or config3" > gpurun_out/r2/pytest_tail.log 2>&1 || { tail -30 gpurun_out/r2/pytest_tail.log; exit 1; }
tail -2 gpurun_out/r2/pytest_tail.log
run() { tag=$1; shift; env "$@" python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))"; }
for rep in 1 2 3; do
  run tail1 MSIREN_F16_TAIL=1
  run tail0 MSIREN_F16_TAIL=0
done
for t in 1 0; do
env MSIREN_F16_TAIL=$t python bench.py --steps 1000 --warmup 100 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail=$t default (2 streams):', round(d['value'],1), 'Mpx/s; roofline phase trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), d['extra'])"
done
