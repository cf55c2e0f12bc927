# A/B of the split-fp16 trunk's MFMA tile on one box: parity subset on the default (16x16x32) kernel, then bench lines
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3 or fixtures or forward_tiles or config3 or two_stream or default_precision or linearity or slice_recon" > gpurun_out/r2/pytest_n16.log 2>&1 || { tail -40 gpurun_out/r2/pytest_n16.log; exit 1; }
tail -3 gpurun_out/r2/pytest_n16.log
for t in 32 16 32 16; do
  MSIREN_F16_TILE=$t python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check > gpurun_out/r2/ab_tile${t}_s1.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2/ab_tile${t}_s1.json').read().strip().splitlines()[-1])
print('tile', $t, 'streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))
PY
done
for t in 32 16; do
  MSIREN_F16_TILE=$t python bench.py --steps 1000 --warmup 100 --streams 2 --no-cpu-baseline --no-extras > gpurun_out/r2/ab_tile${t}_s2.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2/ab_tile${t}_s2.json').read().strip().splitlines()[-1])
print('tile', $t, 'streams 2:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4))
PY
done
