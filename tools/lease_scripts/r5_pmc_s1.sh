#!/bin/bash
# Round 5: PMC passes (separate runs, tools/profile.sh) of the one-stream command: what the one-launch prologue's kernels fetch and where from
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/pmc_s1
rm -rf $out && mkdir -p $out
bash tools/profile.sh $out/raw --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $out/profile.log 2>&1
python3 tools/pmc_summary.py $out/raw > $out/pmc_summary_streams1.txt; rm -rf $out/raw
grep -A40 "latent_mods" $out/pmc_summary_streams1.txt | head -60
grep -A40 "encoder_conv_f16x3" $out/pmc_summary_streams1.txt | head -45
