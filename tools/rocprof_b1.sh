#!/bin/bash
# kernel-level timing of a single-tile forward (BASELINE configs[0]); run via gpurun
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/b1_prof
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/latency.py > $out/latency.txt 2> $out/err.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# first 60 dispatches after warm-up belong to B=1
tr=[i for i,r in enumerate(rows) if 'trunk' in r['Kernel_Name']]
i0=tr[20]
prev=int(rows[tr[19]]['End_Timestamp'])
for r in rows[tr[19]+1:i0+1]:
    print(r['Kernel_Name'][:48].ljust(50), 'grid', r['Grid_Size_X'], 'start+%5.1f us'%((int(r['Start_Timestamp'])-prev)/1e3), 'dur %5.1f us'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print('step span %.1f us'%((int(rows[i0]['End_Timestamp'])-prev)/1e3))
PY
