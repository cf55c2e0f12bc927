# experiment batch: sched variants 3/4 vs shipped (1); ring depth 3 vs 4 single-stream; probe with B operands in AGPRs
mkdir -p gpurun_out/r2
run() { tag=$1; shift; "$@" python bench.py --steps 600 --warmup 100 --streams 1 --no-cpu-baseline --no-extras --check | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'streams 1:', round(d['value'],1), 'Mpx/s trunk', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'nerr', d.get('check_nerr_vs_fp64_oracle'))"; }
for rep in 1 2; do
  run sgb1 env
  run sgb3 env MSIREN_LIB=$PWD/ab/libmsiren_sgb3.so
  run sgb4 env MSIREN_LIB=$PWD/ab/libmsiren_sgb4.so
  run ring3 env MSIREN_F16_RING=3
done
for v in main sgb3 sgb4; do
  if [ $v = main ]; then L=""; else L="MSIREN_LIB=$PWD/ab/libmsiren_$v.so"; fi
  env $L python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v streams 2:', round(d['value'],1), 'Mpx/s')"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_shape_probe.hip -o /tmp/shape && /tmp/shape | grep -E "256 CUs" | tail -8
