set -e
mkdir -p gpurun_out/r2
for g in 256 240 225 200; do
  MSIREN_GRID=$g python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline > gpurun_out/r2/grid_${g}_s1.json
done
for g in 256 225; do
  MSIREN_GRID=$g python bench.py --steps 600 --warmup 100 --streams 2 --no-cpu-baseline > gpurun_out/r2/grid_${g}_s2.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/grid_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))
PY
