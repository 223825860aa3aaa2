#!/bin/bash
# round 2, pass 28: LCP cell edge / delta (S4P_CELL_FACTOR) around the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'], round(r['exclusive']['avg_launch_ms'], 4), r['pass_fractions'], round(r['kbar'], 3))"; }
q S4P_CELL_FACTOR=1.002
q S4P_CELL_FACTOR=1.1
q S4P_CELL_FACTOR=1.25
q S4P_CELL_FACTOR=1.5
q S4P_CELL_FACTOR=2.0
