# round-2 GPU pass 1: parity suite, MFMA shape probe, bench lines (N=1, gloo 2 ranks, strong scaling)
set -e
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_gpu_1.log 2>&1 || { tail -40 gpurun_out/r2/pytest_gpu_1.log; exit 1; }
tail -3 gpurun_out/r2/pytest_gpu_1.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_shape_probe.hip -o /tmp/shape
/tmp/shape > gpurun_out/r2/mfma_shape_probe.txt
cat gpurun_out/r2/mfma_shape_probe.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_driver_like.json
python bench.py > gpurun_out/r2/bench_default.json
python bench.py --total-slices 64 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2/bench_strong64_n1.json
MSIREN_BENCH_BACKEND=gloo python bench.py --gpus 4 --total-slices 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_strong64_gloo4_one_card.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['n_gpus'], d['scaling'], round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), d.get('extra'))
PY
