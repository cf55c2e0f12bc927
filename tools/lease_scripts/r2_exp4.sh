mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3 or fixtures or forward_tiles or two_stream or half_unit or linearity or slice_recon or variants" > gpurun_out/r2/pytest_exp4.log 2>&1 || { tail -30 gpurun_out/r2/pytest_exp4.log; exit 1; }
tail -2 gpurun_out/r2/pytest_exp4.log
run() { tag=$1; shift; "$@" python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))"; }
for rep in 1 2; do
  run shipped env
  run sgb4 env MSIREN_LIB=$PWD/ab/libmsiren_sgb4.so
done
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped streams 2:', round(d['value'],1), 'Mpx/s')"
python tools/latency.py | head -2
