#!/bin/bash
# Round 5: the synchronous host-pointer call (numpy -> numpy, 400 tiles) with the one-launch prologue: one chunk against the two-chunk cut
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r5/host1
rm -rf $out && mkdir -p $out
MSIREN_HOST_CHUNKS=1 timeout -k 10 120 python3 tools/host_trace.py 50 > $out/chunks1.txt 2> $out/chunks1.err; cat $out/chunks1.txt; grep msiren_forward $out/chunks1.err | tail -2
MSIREN_HOST_CHUNKS=2 timeout -k 10 200 python3 tools/host_trace.py 50 35 25 15 > $out/chunks2.txt 2> $out/chunks2.err; cat $out/chunks2.txt; grep msiren_forward $out/chunks2.err | tail -8
