#!/bin/bash
# round 5, run 11: kernel timelines of the group pipeline: one group alone (3 lanes) and four groups (12 lanes)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=10
O=gpurun_out/r5_run11; mkdir -p $O
for cfg in "3 3" "12 3"; do set -- $cfg
( cd /tmp && S4P_LANES=$1 S4P_GROUP=$2 timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/trace_$1_$2" -o t -- python "$GRAFT_REPO_ROOT/tools/r5/tp_probe.py" 150 trace > "$GRAFT_REPO_ROOT/$O/probe_$1_$2.json" 2> "$GRAFT_REPO_ROOT/$O/err_$1_$2.log" )
python - <<PY
import csv, glob, json
f = glob.glob('$O/trace_$1_$2/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = [[r['Kernel_Name'][:24], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id'), r.get('Grid_Size_X'), r.get('Grid_Size_Y')] for r in rows]
t0 = min(o[1] for o in out)
for o in out: o[1] -= t0; o[2] -= t0
json.dump(out, open('$O/timeline_$1_$2.json', 'w'))
print(open('$O/probe_$1_$2.json').read()[:300])
PY
rm -rf $O/trace_$1_$2
done
