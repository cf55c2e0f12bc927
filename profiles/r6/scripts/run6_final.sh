#!/bin/bash
# Round 6, lease 6: the record of the final tree.  GPU suite once, smoke, latency of small fp32 calls before / after the fused per-tile
# encoder left the < 48-tile path (A/B library: mri_inr_amd/libmsiren_ab.so = the tree one commit earlier), the driver's command, the default
# and one-stream lines, rocprofv3 kernel stats + PMC passes of the default command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6/final; rm -rf $out; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $out/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -6 $out/pytest.log
if [ $rc -ge 124 ]; then echo "suite killed: no further GPU step in this call"; exit $rc; fi
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $out/smoke.log
for lib in libmsiren_ab.so libmsiren.so; do
  echo "== fp32 latency, $lib" | tee -a $out/latency_fp32.txt
  MSIREN_LIB=$GRAFT_REPO_ROOT/mri_inr_amd/$lib timeout -k 10 200 python3 tools/latency.py fp32 2>&1 | grep -v amdgpu.ids | tee -a $out/latency_fp32.txt
done
timeout -k 10 200 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tee $out/latency_default.txt
b() { name=$1; shift; timeout -k 10 420 python3 bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc $?"; python3 -c "
import json
d=json.loads([l for l in open('$out/bench_$name.json') if l.startswith('{')][0])
print('$name', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'h2h', d.get('host_to_host',{}).get('value'), 'fp32', d.get('fp32',{}).get('value'))
" || true; }
b driver_like --gpus 1 --steps 20 --warmup 5
b default --steps 1000 --warmup 50
b streams1 --streams 1 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_def -- python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-extras > $out/bench_under_rocprof_default.json 2> $out/prof_def.err
f=$(find $out/prof_def -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_default_two_streams.csv; rm -rf $out/prof_def; head -6 $out/kernel_stats_default_two_streams.csv | cut -c1-170
bash tools/profile.sh $out/pmc_def --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/pmc_def.log 2>&1
python3 tools/pmc_summary.py $out/pmc_def > $out/pmc_summary_default_two_streams.txt; rm -rf $out/pmc_def
grep -A26 "siren_trunk_f16x3n_kernel" $out/pmc_summary_default_two_streams.txt | head -30
