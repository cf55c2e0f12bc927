# same-box A/B of library builds through MSIREN_LIB: for every .so given, parity subset (first one only) + bench lines
# usage: bash tools/r2_ab_lib.sh tag1=path1.so tag2=path2.so ...   ("main" = the in-tree build)
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3 or fixtures or forward_tiles or two_stream or default_precision or linearity or slice_recon" > gpurun_out/r2/pytest_ab.log 2>&1 || { tail -40 gpurun_out/r2/pytest_ab.log; exit 1; }
tail -2 gpurun_out/r2/pytest_ab.log
for rep in 1 2; do
for kv in "$@"; do
  tag=${kv%%=*}; lib=${kv#*=}
  if [ "$lib" = "main" ]; then unset MSIREN_LIB; else export MSIREN_LIB=$PWD/$lib; fi
  python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check > gpurun_out/r2/ab_${tag}_s1.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2/ab_${tag}_s1.json').read().strip().splitlines()[-1])
print('$tag', 'streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))
PY
done
done
for kv in "$@"; do
  tag=${kv%%=*}; lib=${kv#*=}
  if [ "$lib" = "main" ]; then unset MSIREN_LIB; else export MSIREN_LIB=$PWD/$lib; fi
  python bench.py --steps 1000 --warmup 100 --streams 2 --no-cpu-baseline --no-extras > gpurun_out/r2/ab_${tag}_s2.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2/ab_${tag}_s2.json').read().strip().splitlines()[-1])
print('$tag', 'streams 2:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4))
PY
done
