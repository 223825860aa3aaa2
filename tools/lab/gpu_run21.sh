#!/bin/bash
# round 2, pass 21: bench.py (timed region + parity gate + time-to-register) with 3..6 lanes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lanes in 4 5 6 3 8; do
  S4P_LANES=$lanes timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes $lanes', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'median', 'max')], d['parity']['mismatches'], d['stage_ms_per_step'], d['config']['time_to_register']['seconds'])"
done
