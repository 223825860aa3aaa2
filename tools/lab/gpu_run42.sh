#!/bin/bash
# round 2, pass 42: rocprofv3 kernel stats of the bench command's default-configuration launches only (no exclusive re-run, no
# instrumented / parity launches), to be set against roofline.avg_launch_ms
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2stats42_bench -o r --output-format csv -- python $R/bench.py --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive > $R/gpurun_out/r2stats42_bench.log 2>&1
cd $R
grep -v "^[EW]2026" gpurun_out/r2stats42_bench.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench avg_launch_ms', d['roofline']['avg_launch_ms'], 'value', d['value'])"
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2stats42_bench/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print('  ', r['Name'][:44], r['Calls'], r['AverageNs'], r['Percentage'])
PY
