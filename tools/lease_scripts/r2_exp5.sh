# two-stream mode: does reserving CUs for the other stream's small kernels (trunk grid < 256, ring 4 = no LDS left beside
# a trunk workgroup) beat co-residency (grid 256, ring 3)?
mkdir -p gpurun_out/r2
run() { tag=$1; shift; env "$@" python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'streams 2:', round(d['value'],1), 'Mpx/s  step', round(d['ms_per_step'],4), 'ms')"; }
for rep in 1 2; do
  run "grid256_ring3(shipped)" X=1
  run grid256_ring4 MSIREN_F16_RING=4
  run grid248_ring4 MSIREN_F16_RING=4 MSIREN_GRID=248
  run grid240_ring4 MSIREN_F16_RING=4 MSIREN_GRID=240
  run grid232_ring4 MSIREN_F16_RING=4 MSIREN_GRID=232
  run grid248_ring3 MSIREN_GRID=248
  run grid240_ring3 MSIREN_GRID=240
done
