#!/bin/bash
# round 2, pass 26: single-entry exact stage (fewer vector instructions) against the dual-entry one in the final configuration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'], round(r['avg_launch_ms'], 4), r['exclusive'] and round(r['exclusive']['avg_launch_ms'], 4))"; }
q S4P_LIB=$R/scratch/libsingle.so
q S4P_X=1
q S4P_LIB=$R/scratch/libsingle.so
q S4P_X=1
