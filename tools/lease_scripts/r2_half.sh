# half-unit trunk instance: parity (bit-exact split tests included), latency, single-slice A/B through MSIREN_F16_HALF
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "not config5 and not residual" > gpurun_out/r2/pytest_half.log 2>&1 || { tail -40 gpurun_out/r2/pytest_half.log; exit 1; }
tail -2 gpurun_out/r2/pytest_half.log
for hf in 1 0 1 0; do
  echo "MSIREN_F16_HALF=$hf"
  MSIREN_F16_HALF=$hf python tools/latency.py 2>&1 | tail -3
  MSIREN_F16_HALF=$hf python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))"
done
for hf in 1 0; do
  MSIREN_F16_HALF=$hf python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half=$hf streams 2:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4))"
done
