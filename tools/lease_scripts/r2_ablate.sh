# timeline (in-kernel stamps) + single-stream trunk time for ablation builds of the 16x16x32 trunk
mkdir -p gpurun_out/r2
for v in main abl1 abl4 abl8 abl10 abl15; do
  if [ "$v" = "main" ]; then unset MSIREN_LIB; else export MSIREN_LIB=$PWD/ab/libmsiren_$v.so; fi
  echo "== $v"
  python tools/timeline_f16x3.py 400 2>&1 | grep -E "loads|hidden|final|pass total|clock"
  python bench.py --steps 300 --warmup 50 --streams 1 --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   trunk ms', round(d['roofline']['avg_launch_ms'],4))"
done 2>&1 | tee gpurun_out/r2/ablation_n16.txt
