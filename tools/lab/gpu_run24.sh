#!/bin/bash
# round 2, pass 24: bench.py with the exclusive / aggregate roofline rows; k_verify workgroup count and lane count around the defaults
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'], round(r['avg_launch_ms'], 4), r['exclusive'] and {k: round(v, 4) for k, v in r['exclusive'].items() if k != 'note'}, {k: round(v, 3) for k, v in r['aggregate'].items() if k != 'note'})"; }
q S4P_X=0
q S4P_VERIFY_BLOCKS=384
q S4P_VERIFY_BLOCKS=640
q S4P_VERIFY_BLOCKS=768
q S4P_VERIFY_BLOCKS=1024
q S4P_LANES=7
