#!/bin/bash
# round 4, run 14: kernel trace of the selection batch probe (where do the ~340 us of a call go?)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run14; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o r -- python $R/tools/r4/select_probe.py > $R/$O/probe.json 2> $R/$O/probe.err
cd $R
cat $O/probe.json
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r4_run14/prof/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print('  ', r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
