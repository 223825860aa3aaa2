#!/bin/bash
# round 2, pass 20: bench.py (timed region + parity gate only) with 3 and 4 lanes on the new defaults
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lanes in 3 4 3 4; do
  S4P_LANES=$lanes timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes $lanes', round(d['value'] / 1e6, 2), d['spread'], d['parity']['mismatches'], d['stage_ms_per_step'])"
done
