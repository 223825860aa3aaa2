#!/bin/bash
# round 5, run 4: timeline of the six-lane pipeline (kernel trace with timestamps) to see where the GPU idles
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run4; mkdir -p $O
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/trace" -o t -- python "$GRAFT_REPO_ROOT/tools/r5/tp_probe.py" 150 trace > "$GRAFT_REPO_ROOT/$O/probe.json" 2> "$GRAFT_REPO_ROOT/$O/err.log" )
find $O/trace -name "*.csv" | head; cat $O/probe.json
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5_run4/trace/**/*kernel_trace.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(len(rows), rows[0].keys())
# keep it small: name(short), start, end, queue, stream
import json
out = []
for r in rows:
    out.append([r['Kernel_Name'][:24], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id'), r.get('Stream_Id'), r.get('Workgroup_Size_X'), r.get('Grid_Size_X')])
t0 = min(o[1] for o in out)
for o in out: o[1] -= t0; o[2] -= t0
json.dump(out, open('gpurun_out/r5_run4/timeline.json', 'w'))
PY
rm -rf $O/trace
ls -la $O
